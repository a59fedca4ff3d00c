"""Dirichlet posterior over mixing weights / transition rows (surface of the reference's dists/Dirichlet.py:3-86).

A K-vector of concentrations per batch element: O(K) elementwise arithmetic and `digamma` / `lgamma`, i.e. plain torch
on the device -- except the KL term, whose ~20 parameter-sized launches per call K15 does in one.  The update is the usual
blend `alpha <- lr (alpha_0 + counts) + (1 - lr) alpha`; the counts themselves (responsibilities summed over
samples) come out of the fused E-step kernels.
"""
import torch

from .._common import as_param, blend, derived_key, resolve


def _finite(values, bad):
    """zero out the +/-inf that lgamma / digamma return at a zero concentration (structural zeros of a masked
    transition matrix), as the reference's KL helpers do (:65-71)"""
    return torch.where(values == bad, torch.zeros_like(values), values)


class Dirichlet():
    def __init__(self, event_shape, batch_shape=(), prior_parms=None, device=None, dtype=None):
        self.device, self.dtype = resolve(device, dtype)
        self.event_shape, self.batch_shape = tuple(event_shape), tuple(batch_shape)
        self.event_dim, self.batch_dim = len(self.event_shape), len(self.batch_shape)
        concentration = 0.5 if prior_parms is None else prior_parms['alpha']
        shape = self.batch_shape + self.event_shape
        self.alpha_0 = as_param(concentration, self.device, self.dtype).expand(shape)
        # the reference starts from a jittered copy of the prior (:10)
        self.alpha = self.alpha_0 * (1.0 + torch.rand(shape, device=self.device, dtype=self.dtype))
        self.NA = 0.0

    def _ev(self):
        return tuple(range(-self.event_dim, 0))

    def _total(self, keepdim=True):
        return self.alpha.sum(self._ev(), keepdim=keepdim)

    def to_event(self, n):
        if n != 0:
            self.event_dim, self.batch_dim = self.event_dim + n, self.batch_dim - n
            self.event_shape = self.batch_shape[-n:] + self.event_shape
            self.batch_shape = self.batch_shape[:-n]
        return self

    def ss_update(self, NA, lr=1.0, beta=None):
        assert NA.shape == self.batch_shape + self.event_shape
        self.NA = NA if beta is None else beta * self.NA + NA
        self.alpha = blend(self.alpha_0 + self.NA, self.alpha, lr)

    def raw_update(self, X, p=None, lr=1.0, beta=None):
        sample_axes = tuple(range(X.ndim - self.event_dim - self.batch_dim))
        weighted = X if p is None else X * p.reshape(tuple(p.shape) + (1,) * self.event_dim)
        self.ss_update(weighted.sum(sample_axes), lr, beta)

    update = raw_update

    def Elog_like(self, X):
        ev = self._ev()
        return (X * self.loggeomean()).sum(ev) + torch.lgamma(1 + X.sum(ev)) - torch.lgamma(1 + X).sum(ev)

    def mean(self):
        return self.alpha / self._total()

    def loggeomean(self):
        # kept until alpha is rebound or written: a VB iteration reads it in the E-step and again in the evidence (4 launches each)
        key = derived_key(self.alpha) + (self.event_dim,)
        c = self.__dict__.get("_vbmp_loggeomean")
        if c is None or c[0] != key:
            c = self._vbmp_loggeomean = (key, torch.digamma(self.alpha) - torch.digamma(self._total()), self.alpha)
        return c[1]

    ElogX = loggeomean

    def var(self):
        m = self.mean()
        return m * (1 - m) / (self._total() + 1)

    def KL_lgamma(self, x):
        return _finite(torch.lgamma(x), torch.inf)

    def KL_digamma(self, x):
        return _finite(torch.digamma(x), -torch.inf)

    def KLqprior(self):
        if self.alpha.is_cuda:  # K15: one launch (composed below: ~20)
            from .. import ops
            return ops.dirichlet_kl(self.alpha, self.alpha_0, self.event_dim)
        return self._KLqprior_composed()

    def _KLqprior_composed(self):
        ev = self._ev()
        tot, tot0 = self._total(False), self.alpha_0.sum(ev)
        log_norm = torch.lgamma(tot) - self.KL_lgamma(self.alpha).sum(ev)
        log_norm0 = torch.lgamma(tot0) - self.KL_lgamma(self.alpha_0).sum(ev)
        elog = self.KL_digamma(self.alpha) - torch.digamma(self._total())
        KL = log_norm - log_norm0 + ((self.alpha - self.alpha_0) * elog).sum(ev)
        while KL.ndim > self.batch_dim:
            KL = KL.sum(-1)
        return KL

    def logZ(self):
        return torch.lgamma(self.alpha).sum(self._ev()) - torch.lgamma(self._total(False))
