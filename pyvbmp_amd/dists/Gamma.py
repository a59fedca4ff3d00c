"""Gamma posterior (conjugate prior of a Poisson rate / of a diagonal precision entry).  O(B.D) elementwise
arithmetic: plain torch ops on the device.  Surface of the reference's dists/Gamma.py:6-105."""
import torch

from .._common import as_param, resolve


class Gamma():
    def __init__(self, event_shape=(), batch_shape=(), prior_parms=None, device=None, dtype=None):
        if prior_parms is None:
            prior_parms = {'alpha': 1.0, 'beta': 1.0}
        self.device, self.dtype = resolve(device, dtype)
        self.event_shape, self.batch_shape = tuple(event_shape), tuple(batch_shape)
        self.event_dim, self.batch_dim = len(self.event_shape), len(self.batch_shape)
        self.nat_parms_0 = prior_parms
        full = self.batch_shape + self.event_shape
        self.alpha_0 = as_param(prior_parms['alpha'], self.device, self.dtype).expand(full)
        self.beta_0 = as_param(prior_parms['beta'], self.device, self.dtype).expand(full)
        self.alpha = self.alpha_0 + torch.rand(full, device=self.device, dtype=self.dtype)
        self.beta = self.beta_0 + torch.rand(full, device=self.device, dtype=self.dtype)
        self.SEx = 0.0
        self.SElogx = 0.0

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

    def ss_update(self, SElogx, SEx, lr=1.0, beta=None):
        assert (SElogx.ndim == self.batch_dim + self.event_dim)
        assert (SEx.ndim == self.batch_dim + self.event_dim)
        if beta is not None:
            self.SEx = beta * self.SEx + SEx
            self.SElogx = beta * self.SElogx + SElogx
            SEx, SElogx = self.SEx, self.SElogx
        self.alpha = (self.alpha_0 + SElogx) * lr + self.alpha * (1 - lr)
        self.beta = (self.beta_0 + SEx) * lr + self.beta * (1 - lr)

    def _count_and_sum(self, X, p):
        sd = tuple(range(X.ndim - self.event_dim - self.batch_dim))
        if p is None:
            n = 1
            for i in sd:
                n *= X.shape[i]
            N = torch.tensor(float(n), device=X.device, dtype=X.dtype).expand(self.batch_shape + self.event_shape)
            return X.sum(sd), N
        pe = p.reshape(tuple(p.shape) + (1,) * self.event_dim)
        return (X * pe).sum(sd), pe.sum(sd)

    def update(self, pX, p=None, lr=1.0, beta=None):
        SEx, N = self._count_and_sum(pX.mean(), p)
        self.ss_update(SEx, N, lr=lr, beta=beta)

    def raw_update(self, X, p=None, lr=1.0, beta=None):
        SEx, N = self._count_and_sum(X, p)
        self.ss_update(SEx, N, lr=lr, beta=beta)

    def Elog_like(self, X):
        return (X * self.loggeomean() - (X + 1).lgamma() - self.mean()).sum(self._ev())

    def mean(self):
        return self.alpha / self.beta

    def var(self):
        return self.alpha / self.beta ** 2

    def meaninv(self):
        return self.beta / (self.alpha - 1)

    def ElogX(self):
        return self.alpha.digamma() - self.beta.log()

    def loggeomean(self):
        return self.alpha.log() - self.beta.log()

    def entropy(self):
        return self.alpha.log() - self.beta.log() + self.alpha.lgamma() + (1 - self.alpha) * self.alpha.digamma()

    def logZ(self):
        return -self.alpha * self.beta.log() + self.alpha.lgamma()

    def logZprior(self):
        return -self.alpha_0 * self.beta_0.log() + self.alpha_0.lgamma()

    def KLqprior(self):
        KL = (self.alpha - self.alpha_0) * self.alpha.digamma() - self.alpha.lgamma() + self.alpha_0.lgamma() \
            + self.alpha_0 * (self.beta.log() - self.beta_0.log()) + self.alpha * (self.beta_0 / self.beta - 1)
        return KL.sum(self._ev())
