"""Diagonal precision matrix whose entries are independent Gamma variables: the noise model of MatrixNormalGamma
(surface of the reference's dists/DiagonalWishart.py:7-66).  A thin adapter over `Gamma`: the matrix-valued
expectations are the Gamma expectations laid out on a diagonal, the scalar ones sums over the event axis."""
import torch

from .._common import as_param, resolve
from .Gamma import Gamma

# reference method -> (Gamma expectation, how it is presented)
_DIAGONAL = {"ESigma": "meaninv", "EinvSigma": "mean", "mean": "mean"}


class DiagonalWishart():
    def __init__(self, event_shape, batch_shape=(), prior_parms=None, scale=1.0, device=None, dtype=None):
        prior = {'nu': 2.0, 'U': 0.5} if prior_parms is None else prior_parms
        self.device, self.dtype = resolve(device, dtype)
        self.event_shape, self.batch_shape = tuple(event_shape), tuple(batch_shape)
        self.event_dim, self.batch_dim = len(self.event_shape), len(self.batch_shape)
        self.dim = self.event_shape[-1]
        shape_parm = as_param(prior['nu'], self.device, self.dtype)
        rate_parm = scale ** 2 / as_param(prior['U'], self.device, self.dtype)
        self.gamma = Gamma(self.event_shape, self.batch_shape, prior_parms={'alpha': shape_parm, 'beta': rate_parm},
                           device=self.device, dtype=self.dtype)

    def to_event(self, n):
        if n != 0:
            self.event_dim, self.batch_dim = self.event_dim + n, self.batch_dim - n
            self.event_shape = self.batch_shape[-n:] + self.event_shape
            self.batch_shape = self.batch_shape[:-n]
            self.gamma.to_event(n)
        return self

    def ss_update(self, SExx, N, lr=1.0, beta=None):
        """SExx: the DIAGONAL of the second-moment statistic, N: counts, both batch + event shaped"""
        nd = self.batch_dim + self.event_dim
        assert SExx.ndim == nd and N.ndim == nd
        self.gamma.ss_update(0.5 * N, 0.5 * SExx, lr, beta)

    def tensor_diag(self, A):
        return A.unsqueeze(-1) * torch.eye(A.shape[-1], device=A.device, dtype=A.dtype)

    def tensor_extract_diag(self, A):
        return A.diagonal(dim1=-2, dim2=-1)

    def ElogdetinvSigma(self):
        return self.gamma.loggeomean().sum(-1)

    def logdetEinvSigma(self):
        return self.gamma.mean().log().sum(-1)


def _delegate(name):
    def method(self):
        return getattr(self.gamma, name)()
    method.__name__ = name
    return method


def _on_diagonal(name, source):
    def method(self):
        return self.tensor_diag(getattr(self.gamma, source)())
    method.__name__ = name
    return method


for _n in ("KLqprior", "logZ"):
    setattr(DiagonalWishart, _n, _delegate(_n))
for _n, _src in _DIAGONAL.items():
    setattr(DiagonalWishart, _n, _on_diagonal(_n, _src))
