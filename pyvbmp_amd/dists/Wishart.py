"""Wishart posterior over a precision matrix; conjugate update on the HIP device.

Mirrors the method surface of the reference node (dists/Wishart.py:7-97): same constructor,
attributes (invU_0, nu_0, logdet_invU_0, invU, U, nu, logdet_invU, SExx, N), ss_update / to_event /
expectations / KLqprior / logZ.  The inverse + logdet of ss_update run as ONE fused HIP kernel
(K2a, csrc/k_niw.hip) instead of LU inverse + slogdet passes.
"""
import math

import torch

from .. import ops
from .._common import as_param, resolve, trailing

_LOG2 = math.log(2.0)


class Wishart():
    def __init__(self, event_shape, batch_shape=(), scale=1.0, device=None, dtype=None):
        assert event_shape[-1] == event_shape[-2]
        self.device, self.dtype = resolve(device, dtype)
        self.dim = event_shape[-1]
        self.event_shape = tuple(event_shape)
        self.event_dim = len(event_shape)
        self.batch_shape = tuple(batch_shape)
        self.batch_dim = len(batch_shape)
        D = self.dim
        scale = as_param(scale, self.device, self.dtype)
        eye = torch.eye(D, device=self.device, dtype=self.dtype)
        # stride-0 expanded priors, as in the reference (Wishart.py:17-18): never materialised
        self.invU_0 = (scale ** 2 * eye).expand(self.batch_shape + self.event_shape)
        self.nu_0 = torch.tensor(D + 2.0, device=self.device, dtype=self.dtype).expand(
            self.batch_shape + self.event_shape[:-2])
        # closed form for the scaled identity (the reference calls logdet/inverse on it, :20-24)
        self.logdet_invU_0 = (2.0 * D * scale.log()).expand(self.nu_0.shape)
        self.invU = self.invU_0
        self.U = (eye / scale ** 2).expand(self.batch_shape + self.event_shape)
        self.nu = self.nu_0
        self.logdet_invU = self.logdet_invU_0
        self.SExx = 0.0
        self.N = 0.0
        self._arange = torch.arange(D, device=self.device, dtype=self.dtype) / 2.0

    def to_event(self, n):
        if n == 0:
            return self
        self.event_dim = self.event_dim + n
        self.batch_dim = self.batch_dim - n
        self.event_shape = self.batch_shape[-n:] + self.event_shape
        self.batch_shape = self.batch_shape[:-n]
        return self

    def log_mvgamma(self, nu):
        return (nu.unsqueeze(-1) - self._arange).lgamma().sum(-1)

    def log_mvdigamma(self, nu):
        return (nu.unsqueeze(-1) - self._arange).digamma().sum(-1)

    def ss_update(self, SExx, N, lr=1.0, beta=None):
        """Posterior from sufficient statistics (ref dists/Wishart.py:43-56)."""
        assert (SExx.ndim == self.batch_dim + self.event_dim)
        assert (N.ndim == self.batch_dim + self.event_dim - 2)
        if beta is not None:
            self.SExx = SExx + beta * self.SExx
            self.N = N + beta * self.N
            SExx = self.SExx
            N = self.N
        full = self.batch_shape + self.event_shape
        if tuple(SExx.shape) != full:
            SExx = SExx.expand(full)
        self.invU, self.nu, self.U, self.logdet_invU = ops.wishart_ss_update(
            SExx, N.expand(full[:-2]), self.invU_0, self.nu_0, self.invU, self.nu, lr)

    def _nu2(self):
        return trailing(self.nu, 2)

    def _half_nu_terms(self, nu, fn):
        """sum_i fn(nu/2 - i/2), i < dim: the multivariate (di)gamma sums of the Wishart normaliser"""
        return fn(0.5 * nu.unsqueeze(-1) - self._arange).sum(-1)

    def ElogdetinvSigma(self):
        return self._half_nu_terms(self.nu, torch.digamma) + self.dim * _LOG2 - self.logdet_invU

    def logdetEinvSigma(self):
        return torch.log(self.nu) - self.logdet_invU

    def KLqprior(self):
        if self.U.is_cuda:  # K15: one launch
            kl = ops.wishart_kl(self.invU_0, self.U, self.nu, self.nu_0, self.logdet_invU, self.logdet_invU_0)
            for _ in range(self.event_dim - 2):
                kl = kl.sum(-1)
            return kl
        return self._KLqprior_composed()

    def _KLqprior_composed(self):
        half, half0 = 0.5 * self.nu, 0.5 * self.nu_0
        trace_term = (self.invU_0 * self.U).sum((-1, -2))  # tr(invU_0 U)
        kl = half0 * (self.logdet_invU - self.logdet_invU_0) + half * (trace_term - self.dim)
        kl = kl + self._half_nu_terms(self.nu_0, torch.lgamma) - self._half_nu_terms(self.nu, torch.lgamma) \
            + (half - half0) * self._half_nu_terms(self.nu, torch.digamma)
        for _ in range(self.event_dim - 2):
            kl = kl.sum(-1)
        return kl

    def logZ(self):
        half = 0.5 * self.nu
        return self._half_nu_terms(self.nu, torch.lgamma) + half * (self.dim * _LOG2 - self.logdet_invU)


# matrix-valued expectations under the reference's names (dists/Wishart.py:67-80): a cached matrix times or over a
# function of the degrees of freedom
for _name, _matrix, _scale in (("mean", "U", lambda nu, d: nu), ("EinvSigma", "U", lambda nu, d: nu),
                               ("meaninv", "invU", lambda nu, d: 1.0 / (nu - d - 1)),
                               ("ESigma", "invU", lambda nu, d: 1.0 / (nu - d - 1)),
                               ("invEinvSigma", "invU", lambda nu, d: 1.0 / nu)):
    def _make(matrix, scale):
        def method(self):
            return getattr(self, matrix) * scale(self._nu2(), self.dim)
        return method
    setattr(Wishart, _name, _make(_matrix, _scale))
