"""Transform nodes; each class re-exported under its submodule name, as in the reference
(transforms/__init__.py:1-13)."""
from .MatrixNormalGamma import MatrixNormalGamma
from .MatrixNormalWishart import MatrixNormalWishart
from .MixtureofLinearTransforms import MixtureofLinearTransforms
from .MultiNomialLogisticRegression import MultiNomialLogisticRegression
from .dMixtureofLinearTransforms import dMixtureofLinearTransforms
