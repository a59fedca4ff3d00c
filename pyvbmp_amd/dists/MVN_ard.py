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

    def _precision(self, SExx, jitter=0.0):
        """data precision + expected ARD precisions on the diagonal (+ the reference's start-up jitter)"""
        eye = torch.eye(self.dim, device=self.device, dtype=self.dtype)
        return SExx + (self.alpha.mean() + jitter) * eye

    def ss_update(self, SExx, SEx, iters=2, lr=1.0, beta=None):
        """Coordinate ascent between the Gaussian factor and the Gamma factor of the ARD prior (ref :48-72).
        SExx: batch + (n, p, p) data precision, SEx: batch + (n, p, 1) natural mean.  Each sweep: the Gamma factor
        sees E[x_i^2] = Sigma_ii + mu_i^2, the Gaussian factor the new expected precisions.  The reference forms its
        very first mean with the PREVIOUS natural mean (:58); kept, since the first Gamma sweep depends on it."""
        if beta is not None:
            self.SExx, self.SEx = self.SExx * beta + SExx, self.SEx * beta + SEx
            SExx, SEx = self.SExx, self.SEx
        half = torch.full((), 0.5, device=self.device, dtype=self.dtype).expand(self.alpha.batch_shape + self.alpha.event_shape)
        precision = self._precision(SExx, jitter=1e-6)
        cov = ops.spd_inverse(precision)
        mean = cov @ self.invSigmamu
        for sweep in range(iters):
            second_moment = cov.diagonal(dim1=-1, dim2=-2).unsqueeze(-1) + mean ** 2
            self.alpha.ss_update(half, 0.5 * second_moment, lr=lr, beta=beta)
            precision = self._precision(SExx)
            cov = ops.spd_inverse(precision)
            mean = cov @ SEx
        self.invSigma = lr * precision + (1 - lr) * self.invSigma
        self.invSigmamu = lr * SEx + (1 - lr) * self.invSigmamu
        self.Sigma, self.logdetinvSigma = ops.spd_inv_logdet(self.invSigma)
        self.mu = self.Sigma @ self.invSigmamu

    def KLqprior(self):
        ev = list(range(-self.event_dim, 0))
        KL = 0.5 * (self.mu.pow(2) * self.alpha.mean()).sum(ev)
        KL = KL - 0.5 * self.alpha.loggeomean().sum(ev) + 0.5 * self.logdetinvSigma.sum(list(range(2 - self.event_dim, 0)))
        KL = KL + (self.Sigma.diagonal(dim1=-1, dim2=-2) * self.alpha.mean().squeeze(-1)).sum(list(range(1 - self.event_dim, 0)))
        return KL + self.alpha.KLqprior()

    def EXXT(self):
        return self.Sigma + self.mu @ self.mu.transpose(-2, -1)

    def EXTX(self):
        return self.Sigma.sum(-1).sum(-1) + self.mu.pow(2).sum(-2).squeeze(-1)

    def EXTinvUX(self):
        return (self.mu.transpose(-2, -1) @ self.invSigma @ self.mu).squeeze(-1).squeeze(-1)

    def Res(self):
        return - 0.5 * (self.mu * self.invSigmamu).sum(-1).sum(-1) + 0.5 * self.logdetinvSigma \
            - 0.5 * self.dim * torch.log(2 * torch.tensor(torch.pi, device=self.device, dtype=self.dtype))


# the cached moments / natural parameters under the reference's getter names (dists/MVN_ard.py:81-97)
for _getter, _attr in {"mean": "mu", "EX": "mu", "ESigma": "Sigma", "EinvSigma": "invSigma", "EinvSigmamu": "invSigmamu",
                       "ElogdetinvSigma": "logdetinvSigma"}.items():
    setattr(MVN_ard, _getter, (lambda a: lambda self: getattr(self, a))(_attr))
