"""Distribution nodes.  As in the reference (dists/__init__.py:1-18) each class is re-exported under
the name of its submodule, so `import pyvbmp_amd.dists.Wishart as Wishart` yields the class."""
from .Delta import Delta
from .DiagonalWishart import DiagonalWishart
from .Dirichlet import Dirichlet
from .Gamma import Gamma
from .MVN_ard import MVN_ard
from .Mixture import Mixture
from .MultivariateNormal import MultivariateNormal
from .MultivariateNormal_vector_format import MultivariateNormal_vector_format
from .NormalInverseWishart import NormalInverseWishart
from .Wishart import Wishart
