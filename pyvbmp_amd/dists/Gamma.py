"""Gamma posterior: conjugate prior of a Poisson rate and of one diagonal precision entry (DiagonalWishart, the ARD
prior of MVN_ard).  Surface of the reference's dists/Gamma.py:6-105.

Only O(batch x event) elementwise arithmetic happens here, so it is plain torch on the device: the conjugate update
is a blend of natural parameters, and every expectation is a closed form of the shape `a` and the rate `b`, kept in
one table (name -> formula) from which the reference's getter methods are generated.
"""
import torch

from .._common import as_param, blend, resolve

# closed forms of Gamma(shape a, rate b)
_FORMULAS = {
    "mean": lambda a, b: a / b,
    "var": lambda a, b: a / b ** 2,
    "meaninv": lambda a, b: b / (a - 1),
    "ElogX": lambda a, b: torch.digamma(a) - torch.log(b),
    "loggeomean": lambda a, b: torch.log(a) - torch.log(b),
    "entropy": lambda a, b: torch.log(a) - torch.log(b) + torch.lgamma(a) + (1 - a) * torch.digamma(a),
    "logZ": lambda a, b: torch.lgamma(a) - a * torch.log(b),
}


class Gamma():
    def __init__(self, event_shape=(), batch_shape=(), prior_parms=None, device=None, dtype=None):
        self.device, self.dtype = resolve(device, dtype)
        self.nat_parms_0 = prior_parms = {'alpha': 1.0, 'beta': 1.0} if prior_parms is None else prior_parms
        self.event_shape, self.batch_shape = tuple(event_shape), tuple(batch_shape)
        self.event_dim, self.batch_dim = len(self.event_shape), len(self.batch_shape)
        shape = self.batch_shape + self.event_shape
        kw = dict(device=self.device, dtype=self.dtype)
        for key in ("alpha", "beta"):
            prior = as_param(prior_parms[key], self.device, self.dtype).expand(shape)
            setattr(self, key + "_0", prior)
            setattr(self, key, prior + torch.rand(shape, **kw))  # the reference starts off the prior (:20-21)
        self.SEx = self.SElogx = 0.0  # forgetting-factor accumulators

    def _ev(self):
        return tuple(range(-self.event_dim, 0))

    def to_event(self, n):
        if n != 0:
            self.event_dim, self.batch_dim = self.event_dim + n, self.batch_dim - n
            self.event_shape = self.batch_shape[-n:] + self.event_shape
            self.batch_shape = self.batch_shape[:-n]
        return self

    def ss_update(self, SElogx, SEx, lr=1.0, beta=None):
        """natural-parameter blend: shape <- prior shape + SElogx, rate <- prior rate + SEx (ref :34-46)"""
        nd = self.batch_dim + self.event_dim
        assert SElogx.ndim == nd and SEx.ndim == nd
        if beta is not None:
            self.SEx, self.SElogx = beta * self.SEx + SEx, beta * self.SElogx + SElogx
            SEx, SElogx = self.SEx, self.SElogx
        self.alpha = blend(self.alpha_0 + SElogx, self.alpha, lr)
        self.beta = blend(self.beta_0 + SEx, self.beta, lr)

    def _count_and_sum(self, X, p):
        """(sum_s p_s x_s, sum_s p_s) over the sample axes; p None = unit weights"""
        sample_axes = tuple(range(X.ndim - self.event_dim - self.batch_dim))
        if p is not None:
            w = p.reshape(tuple(p.shape) + (1,) * self.event_dim)
            return (X * w).sum(sample_axes), w.sum(sample_axes)
        count = 1
        for ax in sample_axes:
            count *= X.shape[ax]
        N = torch.tensor(float(count), device=X.device, dtype=X.dtype).expand(self.batch_shape + self.event_shape)
        return X.sum(sample_axes), N

    def raw_update(self, X, p=None, lr=1.0, beta=None):
        self.ss_update(*self._count_and_sum(X, p), lr=lr, beta=beta)

    def update(self, pX, p=None, lr=1.0, beta=None):
        self.ss_update(*self._count_and_sum(pX.mean(), p), lr=lr, beta=beta)

    def Elog_like(self, X):
        """expected Poisson log-likelihood of counts X under the rate posterior (ref :76-77)"""
        return (X * self.loggeomean() - torch.lgamma(X + 1) - self.mean()).sum(self._ev())

    def logZprior(self):
        return _FORMULAS["logZ"](self.alpha_0, self.beta_0)

    def KLqprior(self):
        if self.alpha.is_cuda:  # K15: one launch
            from .. import ops
            return ops.gamma_kl(self.alpha, self.beta, self.alpha_0, self.beta_0, self.event_dim)
        return self._KLqprior_composed()

    def _KLqprior_composed(self):
        a, b, a0, b0 = self.alpha, self.beta, self.alpha_0, self.beta_0
        kl = (a - a0) * torch.digamma(a) - torch.lgamma(a) + torch.lgamma(a0) + a0 * (torch.log(b) - torch.log(b0)) \
            + a * (b0 / b - 1)
        return kl.sum(self._ev())


for _name, _formula in _FORMULAS.items():
    setattr(Gamma, _name, (lambda f: lambda self: f(self.alpha, self.beta))(_formula))
