"""Multivariate normal in "vector format" (event = (dim, 1)): the message type exchanged by the
linear-Gaussian transforms (surface of the reference's dists/MultivariateNormal_vector_format.py:3-168).

Attributes mu / Sigma / invSigmamu / invSigma / logdetinvSigma are plain tensors that callers read,
index-assign and reset to None, exactly as with the reference.  Conversions run K1 (one launch gives
the inverse AND the logdet, which is cached in `logdetinvSigma` so that Res() does not factor again).
"""
import math

import torch

from .. import ops

_LOG2PI = math.log(2.0 * math.pi)


class MultivariateNormal_vector_format():
    def __init__(self, mu=None, Sigma=None, invSigmamu=None, invSigma=None, logdetinvSigma=None):
        self.mu = mu
        self.Sigma = Sigma
        self.invSigmamu = invSigmamu
        self.invSigma = invSigma
        self.logdetinvSigma = logdetinvSigma
        ref = mu if mu is not None else invSigmamu
        if ref is None:
            print('mu and invSigmamu are both None: cannont initialize MultivariateNormal')
            return None
        self.dim = ref.shape[-2]
        self.event_shape = tuple(ref.shape[-2:])
        self.batch_shape = tuple(ref.shape[:-2])
        self.batch_dim = len(self.batch_shape)
        self.event_dim = len(self.event_shape)
        self.device, self.dtype = ref.device, ref.dtype

    @property
    def shape(self):
        return self.batch_shape + self.event_shape

    def to_event(self, n):
        if n == 0:
            return self
        self.event_dim = self.event_dim + n
        self.batch_dim = self.batch_dim - n
        self.event_shape = self.batch_shape[-n:] + self.event_shape
        self.batch_shape = self.batch_shape[:-n]
        return self

    def unsqueeze(self, dim):
        assert (dim + self.event_dim < 0)
        parts = [None if t is None else t.unsqueeze(dim) for t in (self.mu, self.Sigma, self.invSigmamu, self.invSigma)]
        return MultivariateNormal_vector_format(*parts).to_event(self.event_dim - 2)

    def _reset_moments(self):
        self.Sigma = None
        self.mu = None
        self.logdetinvSigma = None

    def combiner(self, other):
        self.invSigma = self.EinvSigma() + other.EinvSigma()
        self.invSigmamu = self.EinvSigmamu() + other.EinvSigmamu()
        self._reset_moments()

    def nat_combiner(self, invSigma, invSigmamu):
        self.invSigma = self.EinvSigma() + invSigma
        self.invSigmamu = self.EinvSigmamu() + invSigmamu
        self._reset_moments()

    def _factor_precision(self):
        """Sigma and log det invSigma from ONE factorisation of invSigma."""
        self.Sigma, ld = ops.spd_inv_logdet(self.invSigma)
        if self.logdetinvSigma is None:
            self.logdetinvSigma = ld

    def mean(self):
        if self.mu is None:
            self.mu = self.ESigma() @ self.invSigmamu
        return self.mu

    def ESigma(self):
        if self.Sigma is None:
            self._factor_precision()
        return self.Sigma

    def EinvSigma(self):
        if self.invSigma is None:
            self.invSigma, ld = ops.spd_inv_logdet(self.Sigma)
            if self.logdetinvSigma is None:
                self.logdetinvSigma = -ld
        return self.invSigma

    def EinvSigmamu(self):
        if self.invSigmamu is None:
            self.invSigmamu = self.EinvSigma() @ self.mean()
        return self.invSigmamu

    def ElogdetinvSigma(self):
        if self.logdetinvSigma is None:
            if self.invSigma is None:
                self.EinvSigma()
            else:
                self._factor_precision()
        return self.logdetinvSigma

    def EX(self):
        return self.mean()

    def EXXT(self):
        m = self.mean()
        return self.ESigma() + m @ m.transpose(-2, -1)

    def EXTX(self):
        m = self.mean()
        return self.ESigma().sum((-1, -2)) + (m.transpose(-2, -1) @ m).squeeze(-1).squeeze(-1)

    def Res(self):
        return -0.5 * (self.mean() * self.EinvSigmamu()).sum((-1, -2)) + 0.5 * self.ElogdetinvSigma() \
            - 0.5 * self.dim * _LOG2PI

    def ss_update(self, SExx, SEx, n, lr=1.0):
        # the moment form (the reference defines ss_update twice; this later definition, :121-126, wins)
        n = n.unsqueeze(-1).unsqueeze(-1)
        self.mu = SEx / n
        self.Sigma = SExx / n - self.mu @ self.mu.transpose(-2, -1)
        self.invSigma = None
        self.invSigmamu = None

    def raw_update(self, X, p=None, lr=1.0):
        nsd = X.ndim - self.event_dim - self.batch_dim
        n, SEx, SExx = ops.weighted_moments(X.squeeze(-1), p, nsd, self.batch_shape)
        self.ss_update(SExx, SEx.unsqueeze(-1), n, lr)

    def Elog_like(self, X):
        P = self.EinvSigma()
        d = (X - self.mu).squeeze(-1)
        zero = torch.zeros(self.batch_shape + (self.dim,), device=X.device, dtype=X.dtype)
        cst = 0.5 * self.ElogdetinvSigma() - 0.5 * self.dim * _LOG2PI
        out = ops.quadform_loglike(d, P, zero, cst.expand(self.batch_shape))
        for i in range(self.event_dim - 2):
            out = out.sum(-1)
        return out

    def KLqprior(self):
        return torch.tensor(0.0, device=self.device, dtype=self.dtype)
