"""Normal-inverse-Wishart posterior over (mean, covariance); conjugate updates on the HIP device.

Same surface as the reference node (dists/NormalInverseWishart.py:4-132): constructor signature,
attributes (lambda_mu_0, mu_0, lambda_mu, mu, invU = Wishart, SExx/SEx/N accumulators), ss_update,
raw_update, Elog_like, KLqprior, to_event and the expectation getters.

Hot operations and the kernels behind them:
  ss_update   -> ONE fused kernel (K2): rank-1 corrections + lr blend + inverse + logdet
  raw_update  -> K4 weighted-moment reduction (no (N,K,D,D) temporary) then K2
  Elog_like   -> K3 quadratic log-likelihood (no (N,K,D,D) temporary)
"""
import math

import torch

from .. import ops
from .._common import as_param, collapse_to, resolve, trailing
from .Wishart import Wishart

_LOG2PI = math.log(2.0 * math.pi)


class NormalInverseWishart():
    def __init__(self, event_shape, batch_shape=(), scale=1.0, fixed_precision=False,
                 prior_parms=None, device=None, dtype=None):
        if prior_parms is None:
            prior_parms = {'lambda_mu': 1.0, 'mu': 0.0, 'nu': None, 'invU': None}
        self.device, self.dtype = resolve(device, dtype)
        self.dim = event_shape[-1]
        self.event_shape = tuple(event_shape)
        self.event_dim = len(event_shape)
        self.batch_shape = tuple(batch_shape)
        self.batch_dim = len(batch_shape)
        self.fixed_precision = fixed_precision

        lam0 = as_param(prior_parms['lambda_mu'], self.device, self.dtype)
        self.lambda_mu_0 = lam0.expand(self.batch_shape + (self.event_dim - 1) * (1,))
        self.lambda_mu = self.lambda_mu_0
        self.mu_0 = as_param(prior_parms['mu'], self.device, self.dtype).expand(self.batch_shape + self.event_shape)
        # random initial mean around the prior mean (ref :22)
        self.mu = self.mu_0 + torch.randn(self.mu_0.shape, device=self.device, dtype=self.dtype)

        self.invU = Wishart(event_shape=self.event_shape + (self.dim,), batch_shape=self.batch_shape, scale=scale,
                            device=self.device, dtype=self.dtype)
        p_invU, p_nu = prior_parms.get('invU'), prior_parms.get('nu')
        if p_invU is not None and p_nu is not None:
            if self.invU.invU_0.shape == p_invU.shape:
                self.invU.invU_0 = p_invU.to(device=self.device, dtype=self.dtype)
            else:
                print('Warning: NormalInverseWishart prior invU shape does not match Wishart invU_0 shape.  Using default.')
            if self.invU.nu_0.shape == p_nu.shape:
                self.invU.nu_0 = p_nu.to(device=self.device, dtype=self.dtype)
            else:
                print('Warning: NormalInverseWishart prior nu shape does not match Wishart nu_0 shape.  Using default.')

        self.SExx = torch.tensor(0.0, device=self.device, dtype=self.dtype)
        self.SEx = torch.tensor(0.0, device=self.device, dtype=self.dtype)
        self.N = torch.tensor(0.0, device=self.device, dtype=self.dtype)

    def to_event(self, n):
        if n == 0:
            return self
        self.event_dim = self.event_dim + n
        self.batch_dim = self.batch_dim - n
        self.event_shape = self.batch_shape[-n:] + self.event_shape
        self.batch_shape = self.batch_shape[:-n]
        self.invU.to_event(n)
        return self

    # ------------------------------------------------------------------ updates
    def ss_update(self, SExx, SEx, N, lr=1.0, beta=0.0):
        """Sufficient statistics -> posterior (ref dists/NormalInverseWishart.py:49-68)."""
        assert (SExx.ndim == self.batch_dim + self.event_dim + 1)
        assert (SEx.ndim == self.batch_dim + self.event_dim)
        assert (N.ndim == self.batch_dim + self.event_dim - 1)

        if beta is not None:
            if beta == 0.0:
                # 0*old + new: keep the new statistics without a full-size copy
                self.SExx, self.SEx, self.N = SExx, SEx, N
            else:
                self.SExx = beta * self.SExx + SExx
                self.SEx = beta * self.SEx + SEx
                self.N = beta * self.N + N
            SExx, SEx, N = self.SExx, self.SEx, self.N

        W = self.invU
        mat_batch = self.batch_shape + self.event_shape[:-1]  # one D x D problem per entry
        lam_shape = torch.broadcast_shapes(self.lambda_mu_0.shape, N.shape)
        lam, mu, invU, nu, U, logdet = ops.niw_ss_update(
            SExx.expand(mat_batch + (self.dim, self.dim)), SEx.expand(mat_batch + (self.dim,)), N.expand(mat_batch),
            self.lambda_mu_0.expand(mat_batch), self.mu_0, W.invU_0, W.nu_0,
            self.lambda_mu.expand(mat_batch), self.mu, W.invU, W.nu, lr, fixed_precision=self.fixed_precision)
        self.lambda_mu = collapse_to(lam, lam_shape)
        self.mu = mu
        if self.fixed_precision is False:
            W.invU, W.nu, W.U, W.logdet_invU = invU, nu, U, logdet

    def raw_update(self, X, p=None, lr=1.0, beta=None):
        """Data (+ responsibilities) -> sufficient statistics -> ss_update (ref :70-86)."""
        SExx, SEx, N = self.raw_moments(X, p)
        self.ss_update(SExx, SEx, N, lr, beta)

    def raw_moments(self, X, p=None):
        """(SExx, SEx, N) of raw_update without applying them (sample-sharded runs all-reduce these)."""
        nd = self.event_dim + self.batch_dim
        sample_shape = tuple(X.shape[:X.ndim - nd])
        mat_batch = self.batch_shape + self.event_shape[:-1]
        if p is None:
            N, SEx, SExx = ops.weighted_moments(X, None, len(sample_shape), mat_batch)  # N = prod(sample_shape)
        else:
            pv = p.reshape(tuple(p.shape) + (1,) * (self.event_dim - 1))
            N, SEx, SExx = ops.weighted_moments(X, pv, len(sample_shape), mat_batch)
            N = collapse_to(N, self.batch_shape + (1,) * (self.event_dim - 1))
        return SExx, SEx, N

    def update(self, pX, p=None, lr=1.0, beta=None):
        pass

    # ------------------------------------------------------------------ likelihood / ELBO
    def Elog_like(self, X):
        """E_q[log N(X | mu, Sigma)] per (sample, batch); extra event dims are summed (ref :91-97)."""
        P, b, c = self.mixture_estep_params()
        out = ops.quadform_loglike(X, P, b, c)
        for i in range(self.event_dim - 1):
            out = out.sum(-1)
        return out

    def mixture_estep_params(self, alpha=None):
        """(P, b, c) of the quadratic form  -1/2 x^T P x + x^T b + c  = Elog_like(x), per component
        (fed to the fused mixture E-step kernel K3); with the Dirichlet counts `alpha` of the mixing weights c also carries
        E log pi_k.  One launch (K13) for a plain vector-valued family on the device; the getters otherwise."""
        W = self.invU
        lead = tuple(self.mu.shape[:-1])  # batch + leading event axes: one "component" each
        if self.mu.is_cuda and self.dim <= 64 and (alpha is None or (len(lead) == 1 and alpha.shape == lead)):
            from .. import ops
            K, D = 1, self.dim
            for n in lead:
                K *= n
            P, b, c = ops.niw_estep_params(W.U.expand(lead + (D, D)).reshape(K, D, D), W.nu.expand(lead).reshape(K),
                                           self.mu.reshape(K, D), self.lambda_mu.expand(lead).reshape(K),
                                           W.logdet_invU.expand(lead).reshape(K), alpha)
            return P.reshape(lead + (D, D)), b.reshape(lead + (D,)), c.reshape(lead)
        c = -0.5 * self.EXTinvUX() + 0.5 * W.ElogdetinvSigma() - 0.5 * self.dim * _LOG2PI
        if alpha is not None:
            c = c + torch.digamma(alpha) - torch.digamma(alpha.sum(-1, True))
        return W.EinvSigma(), self.EinvSigmamu(), c

    def KLqprior(self):
        W = self.invU
        if self.mu.is_cuda and type(W) is Wishart:  # K15: Wishart and Normal part in one launch
            lead = tuple(self.mu.shape[:-1])
            kl = ops.wishart_kl(W.invU_0, W.U, W.nu, W.nu_0, W.logdet_invU, W.logdet_invU_0, mu=self.mu, mu0=self.mu_0,
                                lam=self.lambda_mu.expand(lead), lam0=self.lambda_mu_0.expand(lead))
            for i in range(self.event_dim - 1):
                kl = kl.sum(-1)
            return kl
        return self._KLqprior_composed()

    def _KLqprior_composed(self):
        d = self.mu - self.mu_0
        Pd = (self.invU.mean() * d.unsqueeze(-2)).sum(-1)
        KL = 0.5 * (self.lambda_mu_0 / self.lambda_mu - 1 + (self.lambda_mu / self.lambda_mu_0).log()) * self.dim
        KL = KL + 0.5 * self.lambda_mu_0 * (d * Pd).sum(-1)
        for i in range(self.event_dim - 1):
            KL = KL.sum(-1)
        return KL + self.invU.KLqprior()

    # ------------------------------------------------------------------ expectations
    def mean(self):
        return self.mu

    def EX(self):
        return self.mu

    def EXXT(self):
        return self.mu.unsqueeze(-1) * self.mu.unsqueeze(-2) + self.invU.ESigma() / trailing(self.lambda_mu, 2)

    def ESigma(self):
        return self.invU.ESigma()

    def ElogdetinvSigma(self):
        return self.invU.ElogdetinvSigma()

    def EinvSigma(self):
        return self.invU.EinvSigma()

    def EinvSigmamu(self):
        return (self.invU.EinvSigma() @ self.mu.unsqueeze(-1)).squeeze(-1)

    def EinvUX(self):
        return self.EinvSigmamu()

    def EXTinvUX(self):
        return (self.mu * self.EinvSigmamu()).sum(-1) + self.dim / self.lambda_mu
