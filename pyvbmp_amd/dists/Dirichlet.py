"""Dirichlet posterior over mixing weights / transition rows.  O(K) arithmetic: plain torch ops on the
device (out of scope for HIP).  Surface of the reference node, dists/Dirichlet.py:3-86."""
import torch

from .._common import as_param, resolve


class Dirichlet():
    def __init__(self, event_shape, batch_shape=(), prior_parms=None, device=None, dtype=None):
        if prior_parms is None:
            prior_parms = {'alpha': 0.5}
        self.device, self.dtype = resolve(device, dtype)
        self.event_shape = tuple(event_shape)
        self.batch_shape = tuple(batch_shape)
        self.event_dim = len(self.event_shape)
        self.batch_dim = len(self.batch_shape)
        full = self.batch_shape + self.event_shape
        self.alpha_0 = as_param(prior_parms['alpha'], self.device, self.dtype).expand(full)
        self.alpha = self.alpha_0 * (1.0 + torch.rand(full, device=self.device, dtype=self.dtype))
        self.NA = 0.0

    def _ev(self):
        return tuple(range(-self.event_dim, 0))

    def to_event(self, n):
        if n == 0:
            return self
        self.event_dim = self.event_dim + n
        self.batch_dim = self.batch_dim - n
        self.event_shape = self.batch_shape[-n:] + self.event_shape
        self.batch_shape = self.batch_shape[:-n]
        return self

    def ss_update(self, NA, lr=1.0, beta=None):
        assert (NA.shape == self.batch_shape + self.event_shape)
        self.NA = NA if beta is None else beta * self.NA + NA
        self.alpha = lr * (self.NA + self.alpha_0) + (1 - lr) * self.alpha

    def raw_update(self, X, p=None, lr=1.0, beta=None):
        sd = tuple(range(X.ndim - self.event_dim - self.batch_dim))
        if p is None:
            NA = X.sum(sd)
        else:
            NA = (X * p.reshape(tuple(p.shape) + (1,) * self.event_dim)).sum(sd)
        self.ss_update(NA, lr, beta)

    def update(self, X, p=None, lr=1.0, beta=None):
        self.raw_update(X, p, lr, beta)

    def Elog_like(self, X):
        ev = self._ev()
        return (X * self.loggeomean()).sum(ev) + (1 + X.sum(ev)).lgamma() - (1 + X).lgamma().sum(ev)

    def mean(self):
        return self.alpha / self.alpha.sum(self._ev(), keepdim=True)

    def loggeomean(self):
        return self.alpha.digamma() - self.alpha.sum(self._ev(), keepdim=True).digamma()

    def ElogX(self):
        return self.loggeomean()

    def var(self):
        a0 = self.alpha.sum(self._ev(), keepdim=True)
        m = self.mean()
        return m * (1 - m) / (a0 + 1)

    def KL_lgamma(self, x):
        out = x.lgamma()
        return torch.where(out == torch.inf, torch.zeros_like(out), out)

    def KL_digamma(self, x):
        out = x.digamma()
        return torch.where(out == -torch.inf, torch.zeros_like(out), out)

    def KLqprior(self):
        ev = self._ev()
        a_sum, a0_sum = self.alpha.sum(ev), self.alpha_0.sum(ev)
        KL = a_sum.lgamma() - self.KL_lgamma(self.alpha).sum(ev) - a0_sum.lgamma() + self.KL_lgamma(self.alpha_0).sum(ev)
        dg = self.KL_digamma(self.alpha) - a_sum.digamma().reshape(tuple(a_sum.shape) + (1,) * self.event_dim)
        KL = KL + ((self.alpha - self.alpha_0) * dg).sum(ev)
        while KL.ndim > self.batch_dim:
            KL = KL.sum(-1)
        return KL

    def logZ(self):
        ev = self._ev()
        return self.alpha.lgamma().sum(ev) - self.alpha.sum(ev).lgamma()
