"""Matrix-normal-Gamma node: the MatrixNormalWishart regression with DIAGONAL noise precision
(independent Gamma entries).  Surface of the reference's transforms/MatrixNormalGamma.py:10-441; it is the
default transition of LinearDynamicalSystems and of DynamicMarkovBlanketDiscovery.

Everything that does not touch the noise model is inherited from MatrixNormalWishart (same K4 moments,
K1 eliminations, K3a likelihoods); the differences of the reference are kept:
  * the initial mean is drawn WITHOUT the prior mean (:47), the noise update uses the diagonal of the
    statistic (:126), KLqprior uses the Gamma term (:216-231);
  * forward() returns only the message (natural parameters, no Res, :315-335); backward(pY) takes no Res
    argument (:337-357); predict() returns the un-normalised Res (:363-371).
"""
import math

import torch

from .. import ops
from ..dists.DiagonalWishart import DiagonalWishart
from ..dists.MultivariateNormal_vector_format import MultivariateNormal_vector_format
from ..utils.matrix_utils import matrix_utils
from .MatrixNormalWishart import MatrixNormalWishart, _LOG2PI, _T


class MatrixNormalGamma(MatrixNormalWishart):
    def __init__(self, event_shape, batch_shape=(), prior_parms=None, scale=1.0, uniform_precision=False, mask=None,
                 X_mask=None, pad_X=False, fixed_precision=False, device=None, dtype=None):
        self.uniform_precision = uniform_precision
        super().__init__(event_shape, batch_shape, prior_parms=prior_parms, scale=scale, mask=mask, X_mask=X_mask,
                         pad_X=pad_X, fixed_precision=fixed_precision, device=device, dtype=dtype)

    def _initial_mean(self, mu_0):
        return torch.randn(mu_0.shape, device=self.device, dtype=self.dtype) / math.sqrt(self.p)

    def _make_noise(self, event_shape, batch_shape, scale):
        return DiagonalWishart(event_shape=event_shape[:-1], batch_shape=batch_shape, scale=scale, device=self.device,
                               dtype=self.dtype)

    def _update_noise(self, W_arg, N, lr):
        self.invU.ss_update(W_arg.diagonal(dim1=-2, dim2=-1), N.unsqueeze(-1), lr=lr)
        if self.uniform_precision is True:
            self.invU.gamma.alpha = self.invU.gamma.alpha.sum(-1, keepdim=True)  # the reference's "HACK" (:128)

    def _constrain_mean(self, mu, invV, V_new):
        """Constrained posterior mean (see MatrixNormalWishart._constrain_mean).  With a DIAGONAL noise precision the
        objective tr[(M - mu)' E[R] (M - mu) invV] separates over the rows of M, and E[R]_ii cancels from each row's
        normal equations:  invV[F_i, F_i] m_i = (mu invV)[i, F_i]  with F_i the free columns of row i.  The n systems
        are padded to p x p (constrained rows / columns replaced by the identity, right-hand side zero there) and go
        through ONE batched K1 launch -- instead of the reference's single dense system over all zeroed entries
        (transforms/MatrixNormalGamma.py:111-123): 2112 x 2112 for the transition matrix of the flocking DMBD."""
        free = self.mask.to(mu.dtype)                                            # (n, p)
        rhs = (mu @ invV) * free
        Kp = invV.unsqueeze(-3) * free.unsqueeze(-1) * free.unsqueeze(-2) + torch.diag_embed(1.0 - free)
        return (ops.spd_inverse(Kp) @ rhs.unsqueeze(-1)).squeeze(-1)

    def KLqprior(self):
        KL = self._mn_kl()  # (K15; R = the diagonal E[invSigma] of this class)
        if KL is None:
            return self._KLqprior_composed()
        KL = KL + (self.invU.KLqprior() / self.n if self.uniform_precision is True else self.invU.KLqprior())
        for i in range(self.event_dim - 2):
            KL = KL.sum(-1)
        return KL

    def _KLqprior_composed(self):
        KL = self.n / 2.0 * self.logdetinvV - self.n / 2.0 * self.logdetinvV_0 - self.n * self.p / 2.0
        if self.X_mask is not None:
            KL = KL + self.n / 2.0 * self.logdetinvV_0 * (self.X_mask).sum((-1, -2))
        KL = KL + 0.5 * self.n * (self.invV_0 * self.V).sum((-1, -2))
        d = self.mu - self.mu_0
        KL = KL + 0.5 * (self.invV_0 * (_T(d) @ (self.invU.gamma.mean().unsqueeze(-1) * d))).sum((-1, -2))
        for i in range(self.event_dim - 2):
            KL = KL.sum(-1)
        KL = KL + (self.invU.KLqprior() / self.n if self.uniform_precision is True else self.invU.KLqprior())
        for i in range(self.event_dim - 2):
            KL = KL.sum(-1)
        return KL

    # ------------------------------------------------------------------ messages (reference signatures)
    def forward(self, pX):
        """natural-parameter message x -> y (ref :315-335): returns MVN_vf(invSigma, invSigmamu) only"""
        R, G, H = self.EinvSigma(), self.EinvUX(), self.EXTinvUX()
        if self.pad_X:
            Jyx, Jxx = -G[..., :, :-1], H[..., :-1, :-1] + pX.EinvSigma()
            jy, jx = G[..., :, -1:], pX.EinvSigmamu() - H[..., :-1, -1:]
        else:
            Jyx, Jxx = -G, H + pX.EinvSigma()
            jy, jx = torch.zeros(tuple(R.shape[:-1]) + (1,), device=self.device, dtype=self.dtype), pX.EinvSigmamu()
        Pyy, nBiD = matrix_utils.block_precision_marginalizer(R, Jyx, _T(Jyx), Jxx)[0:2]
        return MultivariateNormal_vector_format(invSigma=Pyy, invSigmamu=jy + nBiD @ jx)

    def backward(self, pY):
        """message y -> x (ref :337-357)"""
        return super().backward(pY, Res=0.0)

    def predict(self, X):
        G, H = self.EinvUX(), self.EXTinvUX()
        c = 0.5 * self.ElogdetinvSigma() - 0.5 * self.n * _LOG2PI
        if self.pad_X:
            eta = G[..., :, :-1] @ X + G[..., :, -1:]
            P, b, c = H[..., :-1, :-1], -H[..., :-1, -1], c - 0.5 * H[..., -1, -1]
        else:
            eta = G @ X
            P, b = H, torch.zeros(tuple(c.shape) + (self.p,), device=self.device, dtype=self.dtype)
        Res = ops.quadform_loglike(X.squeeze(-1).expand(tuple(eta.shape[:-2]) + (X.shape[-2],)), P, b, c)
        return MultivariateNormal_vector_format(invSigma=self.EinvSigma(), invSigmamu=eta), Res

    def postdict(self, Y):
        raise NotImplementedError("the reference's MatrixNormalGamma has no postdict")

    # ------------------------------------------------------------------ expectations that differ
    def EinvUX(self):
        return self.invU.gamma.mean().unsqueeze(-1) * self.mu

    def EXTAX(self, A):
        return self.V * (self.invU.gamma.meaninv() * A.diagonal(dim1=-2, dim2=-1)).sum(-1) + _T(self.mu) @ A @ self.mu

    def EXmMUTAXmMU(self, A):
        return self.V * (self.invU.gamma.meaninv() * A.diagonal(dim1=-2, dim2=-1)).sum(-1).sum(-1)

    def EXTinvUX(self):
        return self.n * self.V + _T(self.mu) @ (self.invU.gamma.mean().unsqueeze(-1) * self.mu)

    def EXTX(self):
        return self.V * self.invU.gamma.meaninv().sum() + _T(self.mu) @ self.mu

    def ElogdetinvU(self):
        return self.invU.gamma.loggeomean().sum(-1)

    def ElogdetinvSigma(self):
        return self.invU.gamma.loggeomean().sum(-1)

    def EinvSigma(self):
        return self.invU.mean()

    def logdetEinvSigma(self):
        return self.invU.logdetEinvSigma()

    def invEinvSigma(self):
        return self.invU.tensor_diag(1.0 / self.invU.gamma.mean())
