"""Diagonal precision with independent Gamma entries (surface of the reference's dists/DiagonalWishart.py:7-66)."""
import torch

from .._common import as_param, resolve
from .Gamma import Gamma


class DiagonalWishart():
    def __init__(self, event_shape, batch_shape=(), prior_parms=None, scale=1.0, device=None, dtype=None):
        if prior_parms is None:
            prior_parms = {'nu': 2.0, 'U': 0.5}
        self.device, self.dtype = resolve(device, dtype)
        self.dim = event_shape[-1]
        self.event_shape, self.batch_shape = tuple(event_shape), tuple(batch_shape)
        self.event_dim, self.batch_dim = len(self.event_shape), len(self.batch_shape)
        nu = as_param(prior_parms['nu'], self.device, self.dtype)
        U = as_param(prior_parms['U'], self.device, self.dtype)
        self.gamma = Gamma(self.event_shape, self.batch_shape, prior_parms={'alpha': nu, 'beta': scale ** 2 / U},
                           device=self.device, dtype=self.dtype)

    def to_event(self, n):
        if n == 0:
            return self
        self.event_dim = self.event_dim + n
        self.batch_dim = self.batch_dim - n
        self.event_shape = self.batch_shape[-n:] + self.event_shape
        self.batch_shape = self.batch_shape[:-n]
        self.gamma.to_event(n)
        return self

    def ss_update(self, SExx, N, lr=1.0, beta=None):
        """SExx holds the DIAGONAL of the second-moment statistic"""
        assert (SExx.ndim == self.batch_dim + self.event_dim)
        assert (N.ndim == self.batch_dim + self.event_dim)
        self.gamma.ss_update(N / 2.0, SExx / 2.0, lr, beta)

    def KLqprior(self):
        return self.gamma.KLqprior()

    def logZ(self):
        return self.gamma.logZ()

    def tensor_diag(self, A):
        return A.unsqueeze(-1) * torch.eye(A.shape[-1], device=A.device, dtype=A.dtype)

    def tensor_extract_diag(self, A):
        return A.diagonal(dim1=-2, dim2=-1)

    def ESigma(self):
        return self.tensor_diag(self.gamma.meaninv())

    def EinvSigma(self):
        return self.tensor_diag(self.gamma.mean())

    def mean(self):
        return self.tensor_diag(self.gamma.mean())

    def ElogdetinvSigma(self):
        return self.gamma.loggeomean().sum(-1)

    def logdetEinvSigma(self):
        return self.gamma.mean().log().sum(-1)
