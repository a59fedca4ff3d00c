"""Mixture of linear transforms whose gate depends on the input: `mixture_dim` MatrixNormalWishart / MatrixNormalGamma
experts and a Polya-Gamma multinomial-logistic gate (surface of the reference's
transforms/dMixtureofLinearTransforms.py:8-176; the model of its examples/two_moons.py).

E-step: expert log-likelihoods of every (sample, expert) -- the joint quadratic form of z = [x; y], K3a -- plus the
gate's class log-probabilities; M-step: gate update (K3a + K4 + K1 inside MultiNomialLogisticRegression) and the
responsibility-weighted expert update (K4 moments of z, K1/K2a)."""
import math

import torch

from ..dists.MultivariateNormal_vector_format import MultivariateNormal_vector_format
from .MatrixNormalGamma import MatrixNormalGamma
from .MatrixNormalWishart import MatrixNormalWishart
from .MultiNomialLogisticRegression import MultiNomialLogisticRegression


class dMixtureofLinearTransforms():
    def __init__(self, n, p, mixture_dim, batch_shape=(), pad_X=True, type='Wishart', fixed_precision=False, device=None,
                 dtype=None):
        batch_shape = tuple(batch_shape)
        self.event_shape = (mixture_dim, n, p)
        self.batch_shape = batch_shape
        self.batch_dim = len(batch_shape)
        self.event_dim = 3
        self.n, self.p = n, p
        self.mix_dim = mixture_dim
        scale = 1.0 / mixture_dim ** (1.0 / n)
        if type == 'Wishart':
            cls = MatrixNormalWishart
        elif type == 'Gamma':
            cls = MatrixNormalGamma
        elif type == 'MVN_ard':
            raise NotImplementedError
        else:
            raise ValueError('type must be either Wishart (default) or Gamma')
        self.A = cls(event_shape=(n, p), batch_shape=batch_shape + (mixture_dim,), scale=scale, pad_X=pad_X,
                     fixed_precision=fixed_precision, device=device, dtype=dtype)
        self.device, self.dtype = self.A.device, self.A.dtype
        self.pi = MultiNomialLogisticRegression(mixture_dim, p, batch_shape=batch_shape, pad_X=True, device=self.device,
                                                dtype=self.dtype)
        self.ELBO_last = torch.full((), -torch.inf, device=self.device, dtype=self.dtype)
        self.ELBO_save = []

    @staticmethod
    def _responsibilities(log_p):
        logZ = torch.logsumexp(log_p, -1)
        return torch.exp(log_p - logZ.unsqueeze(-1)), logZ

    def raw_update(self, X, Y, p=None, iters=1, lr=1.0, verbose=False):
        AX = X.unsqueeze(-1).unsqueeze(-3)  # vector format, expert axis
        AY = Y.unsqueeze(-1).unsqueeze(-3)
        for i in range(iters):
            log_p = self.A.Elog_like(AX, AY) + self.pi.log_predict(X)
            p_ass, logZ = self._responsibilities(log_p)
            if verbose:
                ELBO = logZ.sum(0) - self.KLqprior()
                print("dMixture Percent Change in ELBO = ", ((ELBO - self.ELBO_last) / self.ELBO_last.abs()).data * 100)
                self.ELBO_last = ELBO
            self.pi.raw_update(X, p_ass, p=p, lr=lr, verbose=False)
            self.A.raw_update(AX, AY, p=p_ass if p is None else p_ass * p.unsqueeze(-1), lr=lr)

    def update(self, pX, pY, p=None, iters=1, lr=1.0, verbose=False):
        pAX = pX.unsqueeze(-3)
        pAY = pY.unsqueeze(-3)
        for i in range(iters):
            log_p = self.A.Elog_like_given_pX_pY(pAX, pAY) + self.pi.log_forward(pX)
            p_ass, self.logZ = self._responsibilities(log_p)
            self.NA = p_ass.sum(0)
            self.pi.update(pX, p_ass, p=p, lr=lr, verbose=False)
            self.A.update(pAX, pAY, p=p_ass if p is None else p_ass * p.unsqueeze(-1), lr=lr)
            ELBO = self.logZ.sum() - self.KLqprior().sum()
            if verbose:
                print('dMixLT Percent Change in ELBO: ', (ELBO - self.ELBO_last) / self.ELBO_last.abs())
            self.ELBO_last = ELBO

    def postdict(self, Y):
        """message to the input given an observed output: per-expert Gaussian messages combined with the gate's
        message, then mixed by their evidences (ref :52-73)"""
        invSigma, invSigmamu, Res = self.A.Elog_like_X(Y.unsqueeze(-2).unsqueeze(-1))
        like_X = MultivariateNormal_vector_format(invSigma=invSigma.unsqueeze(0).movedim(-3, -3 - self.batch_dim),
                                                  invSigmamu=invSigmamu.movedim(-3, -3 - self.batch_dim))
        Res = Res.movedim(-1, -1 - self.batch_dim)
        Z = torch.eye(self.mix_dim, device=self.device, dtype=self.dtype)
        for i in range(self.batch_dim):
            Z = Z.unsqueeze(-2)
        invSigma, invSigmamu, Sigma, mu, Res_z = self.pi.Elog_like_X(like_X, Z, iters=4)
        Res = Res + Res_z + 0.5 * (mu * invSigmamu).sum(-2).squeeze(-1) - 0.5 * torch.logdet(invSigma) \
            + like_X.dim / 2.0 * math.log(2 * math.pi)
        logZ = Res.logsumexp(-1 - self.batch_dim, True)
        p = (Res - logZ).exp()
        logZ = logZ.squeeze(-1)
        pv = p.reshape(tuple(p.shape) + (1, 1))
        invSigma = (invSigma * pv).sum(-3 - self.batch_dim)
        invSigmamu = (invSigmamu * pv).sum(-3 - self.batch_dim)
        return MultivariateNormal_vector_format(invSigma=invSigma, invSigmamu=invSigmamu), logZ.squeeze(-1 - self.batch_dim), p

    def predict(self, X):
        p = self.pi.predict(X)
        pv = p.reshape(tuple(p.shape) + (1, 1))
        Xv = X.reshape(tuple(X.shape[:-1]) + (1,) + tuple(X.shape[-1:]) + (1,))
        pY = self.A.predict(Xv)[0]
        mu = (pY.mean() * pv).sum(-3)
        Sigma = (pY.EXXT() * pv).sum(-3) - mu @ mu.transpose(-2, -1)
        return MultivariateNormal_vector_format(mu=mu, Sigma=Sigma), p

    def forward(self, pX):
        p = self.pi.forward(pX)
        pv = p.reshape(tuple(p.shape) + (1, 1))
        pY = self.A.forward(pX.unsqueeze(-3))[0]
        mu = (pY.mean() * pv).sum(-3)
        Sigma = (pY.EXXT() * pv).sum(-3) - mu @ mu.transpose(-2, -1)
        return MultivariateNormal_vector_format(Sigma=Sigma, mu=mu)

    def forward_mix(self, pX):
        return self.A.forward(pX.unsqueeze(-3)), self.pi.forward(pX)

    def backward(self, pY):
        pX, ResA = self.A.backward(pY.unsqueeze(-3))
        pX, Res = self.pi.backward(torch.eye(self.mix_dim, device=self.device, dtype=self.dtype), pX)
        log_p = Res + ResA
        p = torch.softmax(log_p, -1).unsqueeze(-1).unsqueeze(-1)
        invSigma = (pX.EinvSigma() * p).sum(-3)
        invSigmamu = (pX.EinvSigmamu() * p).sum(-3)
        return MultivariateNormal_vector_format(invSigma=invSigma, invSigmamu=invSigmamu), log_p - log_p.logsumexp(-1, True)

    def Elog_like_given_pX_pY(self, pX, pY):
        log_p = self.A.Elog_like_given_pX_pY(pX.unsqueeze(-3), pY.unsqueeze(-3)) + self.pi.log_forward(pX)
        return log_p.logsumexp(-1)

    def Elog_like(self, X, Y):
        log_p = self.A.Elog_like(X.unsqueeze(-1).unsqueeze(-3), Y.unsqueeze(-1).unsqueeze(-3)) + self.pi.log_predict(X)
        return log_p.logsumexp(-1)

    def KLqprior(self):
        return self.A.KLqprior().sum(-1) + self.pi.KLqprior()
