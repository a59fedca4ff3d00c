"""Multivariate normal with lazily converted moment / natural parameters, event = last axis
(surface of the reference's dists/MultivariateNormal.py:3-115).

Every `.inverse()` / `.logdet()` of the reference is one K1 launch here (batched SPD inverse +
logdet in a single pass, the logdet cached next to the inverse); the quadratic form of Elog_like is
K3a and the weighted moments of raw_update are K4.
"""
import math

import torch

from .. import ops

_LOG2PI = math.log(2.0 * math.pi)


class MultivariateNormal():
    def __init__(self, mu=None, Sigma=None, invSigmamu=None, invSigma=None):
        self.mu = mu
        self.Sigma = Sigma
        self.invSigmamu = invSigmamu
        self.invSigma = invSigma
        ref = mu if mu is not None else invSigmamu
        if ref is None:
            print('mu and invSigmamu are both None: cannont initialize MultivariateNormal')
            return None
        self.dim = ref.shape[-1]
        self.event_shape = tuple(ref.shape[-1:])
        self.batch_shape = tuple(ref.shape[:-1])
        self.batch_dim = len(self.batch_shape)
        self.event_dim = len(self.event_shape)
        self.device, self.dtype = ref.device, ref.dtype

    def to_event(self, n):
        if n == 0:
            return self
        self.event_dim = self.event_dim + n
        self.batch_dim = self.batch_dim - n
        self.event_shape = self.batch_shape[-n:] + self.event_shape
        self.batch_shape = self.batch_shape[:-n]

    def _matvec(self, M, v):
        return (M @ v.unsqueeze(-1)).squeeze(-1)

    def mean(self):
        if self.mu is None:
            self.mu = self._matvec(ops.spd_inverse(self.invSigma), self.invSigmamu)
        return self.mu

    def ESigma(self):
        if self.Sigma is None:
            self.Sigma = ops.spd_inverse(self.invSigma)
        return self.Sigma

    def EinvSigma(self):
        if self.invSigma is None:
            self.invSigma = ops.spd_inverse(self.Sigma)
        return self.invSigma

    def EinvSigmamu(self):
        if self.invSigmamu is None:
            # NB: the reference multiplies the mean by EinvSigma().inverse() here (MultivariateNormal.py:50-53),
            # i.e. by the covariance.  Kept for parity; the vector-format class has the textbook product.
            self.invSigmamu = self._matvec(ops.spd_inverse(self.EinvSigma()), self.mean())
        return self.invSigmamu

    def ElogdetinvSigma(self):
        if self.Sigma is None:
            return ops.spd_inv_logdet(self.invSigma)[1]
        return -ops.spd_inv_logdet(self.Sigma)[1]

    def EX(self):
        return self.mean()

    def EXXT(self):
        m = self.mean()
        return self.ESigma() + m.unsqueeze(-1) * m.unsqueeze(-2)

    def EXTX(self):
        return self.EXXT().sum((-1, -2))

    def ss_update(self, SExx, SEx, n, lr=1.0):
        self.mu = SEx / n.unsqueeze(-1)
        self.Sigma = SExx / n.unsqueeze(-1).unsqueeze(-1) - self.mu.unsqueeze(-1) * self.mu.unsqueeze(-2)
        self.invSigma = None
        self.invSigmamu = None

    def raw_update(self, X, p=None, lr=1.0):
        nsd = X.ndim - self.event_dim - self.batch_dim
        if p is None:
            n, SEx, SExx = ops.weighted_moments(X, None, nsd, self.batch_shape)
        else:
            n, SEx, SExx = ops.weighted_moments(X, p, nsd, self.batch_shape)
        self.ss_update(SExx, SEx, n, lr)

    def Elog_like(self, X):
        P = self.EinvSigma()
        d = X - self.mu
        zero = torch.zeros(self.batch_shape + (self.dim,), device=X.device, dtype=X.dtype)
        cst = 0.5 * self.ElogdetinvSigma() - 0.5 * self.dim * _LOG2PI
        out = ops.quadform_loglike(d, P, zero, cst.expand(self.batch_shape))
        for i in range(self.event_dim - 2):
            out = out.sum(-1)
        return out

    def KLqprior(self):
        return torch.tensor(0.0, device=self.device, dtype=self.dtype)
