"""Multivariate normal with an automatic-relevance-determination (Gamma) prior on its precision diagonal
(surface of the reference's dists/MVN_ard.py:23-113): the coefficient posterior of the logistic-regression gate
(transforms/MultiNomialLogisticRegression).  Every `.inverse()` / `.logdet()` is one K1 launch over the whole
(batch, class) stack of p x p precisions."""
import torch

from .. import ops
from .._common import resolve
from .Gamma import Gamma


class MVN_ard():
    def __init__(self, event_shape, batch_shape=(), scale=1.0, pad_X=False, device=None, dtype=None):
        event_shape, batch_shape = tuple(event_shape), tuple(batch_shape)
        assert event_shape[-1] == 1
        self.device, self.dtype = resolve(device, dtype)
        kw = dict(device=self.device, dtype=self.dtype)
        self.dim = event_shape[-2]
        self.event_dim = len(event_shape)
        self.event_shape = event_shape
        self.batch_shape = batch_shape
        self.batch_dim = len(batch_shape)
        self.mu = torch.randn(batch_shape + event_shape, **kw) * scale
        self.invSigma = torch.zeros(batch_shape + event_shape[:-1] + (self.dim,), **kw) + torch.eye(self.dim, **kw) / scale ** 2
        self.Sigma = self.invSigma  # sic (ref :35): the initial covariance aliases the precision
        self.logdetinvSigma = ops.spd_inv_logdet(self.invSigma)[1]
        self.invSigmamu = self.invSigma @ self.mu
        self.alpha = Gamma(event_shape, batch_shape, prior_parms={'alpha': 0.5, 'beta': 0.5 * scale ** 2}, **kw)
        self.SEx = 0.0
        self.SExx = 0.0

    def to_event(self, n):
        if n == 0:
            return self
        self.event_dim = self.event_dim + n
        self.batch_dim = self.batch_dim - n
        self.event_shape = self.batch_shape[-n:] + self.event_shape
        self.batch_shape = self.batch_shape[:-n]
        return self

    def ss_update(self, SExx, SEx, iters=2, lr=1.0, beta=None):
        """ref :48-72 (including its first mean, which is formed with the PREVIOUS natural mean, :58)"""
        if beta is not None:
            self.SExx = self.SExx * beta + SExx
            self.SEx = self.SEx * beta + SEx
            SExx, SEx = self.SExx, self.SEx
        eye = torch.eye(self.dim, device=self.device, dtype=self.dtype)
        invSigmamu = SEx
        invSigma = SExx + self.alpha.mean() * eye + 1e-6 * eye
        Sigma = ops.spd_inverse(invSigma)
        mu = Sigma @ self.invSigmamu
        half = torch.full((), 0.5, device=self.device, dtype=self.dtype).expand(self.alpha.batch_shape + self.alpha.event_shape)
        for i in range(iters):
            EXXT = Sigma.diagonal(dim1=-1, dim2=-2).unsqueeze(-1) + mu ** 2
            self.alpha.ss_update(half, 0.5 * EXXT, lr=lr, beta=beta)
            invSigma = SExx + self.alpha.mean() * eye
            Sigma = ops.spd_inverse(invSigma)
            mu = Sigma @ invSigmamu
        self.invSigma = (1 - lr) * self.invSigma + lr * invSigma
        self.invSigmamu = (1 - lr) * self.invSigmamu + lr * invSigmamu
        self.Sigma, self.logdetinvSigma = ops.spd_inv_logdet(self.invSigma)
        self.mu = self.Sigma @ self.invSigmamu

    def KLqprior(self):
        ev = list(range(-self.event_dim, 0))
        KL = 0.5 * (self.mu.pow(2) * self.alpha.mean()).sum(ev)
        KL = KL - 0.5 * self.alpha.loggeomean().sum(ev) + 0.5 * self.ElogdetinvSigma().sum(list(range(2 - self.event_dim, 0)))
        KL = KL + (self.Sigma.diagonal(dim1=-1, dim2=-2) * self.alpha.mean().squeeze(-1)).sum(list(range(1 - self.event_dim, 0)))
        return KL + self.alpha.KLqprior()

    def mean(self):
        return self.mu

    def ESigma(self):
        return self.Sigma

    def EinvSigma(self):
        return self.invSigma

    def EinvSigmamu(self):
        return self.invSigmamu

    def ElogdetinvSigma(self):
        return self.logdetinvSigma

    def EX(self):
        return self.mean()

    def EXXT(self):
        return self.ESigma() + self.mean() @ self.mean().transpose(-2, -1)

    def EXTX(self):
        return self.ESigma().sum(-1).sum(-1) + self.mean().pow(2).sum(-2).squeeze(-1)

    def EXTinvUX(self):
        return (self.mean().transpose(-2, -1) @ self.EinvSigma() @ self.mean()).squeeze(-1).squeeze(-1)

    def Res(self):
        return - 0.5 * (self.mean() * self.EinvSigmamu()).sum(-1).sum(-1) + 0.5 * self.ElogdetinvSigma() \
            - 0.5 * self.dim * torch.log(2 * torch.tensor(torch.pi, device=self.device, dtype=self.dtype))
