"""Multinomial logistic regression by Polya-Gamma augmentation of a stick-breaking softmax
(surface of the reference's transforms/MultiNomialLogisticRegression.py:5-300): the gate of
dMixtureofLinearTransforms.

With n+1 classes there are n stick-breaking logits psi_k = beta_k . [x; 1]; the coefficient posterior
beta ~ MVN_ard((n, p, 1)) is Gaussian given the expected Polya-Gamma weights, so one update is again a
"statistics -> natural parameters" step on the path this package accelerates:

    c_{sk}  = sqrt(x_s' E[beta_k beta_k'] x_s)          quadratic forms of every (sample, logit): K3a
    Ew_{sk} = b_{sk} tanh(c_{sk}/2) / (2 c_{sk})          elementwise
    SExx_k  = sum_s Ew_{sk} x_s x_s'                      weighted second moments with n weight columns: K4
    SEyx_k  = sum_s (y_{sk} - b_{sk}/2) x_s               one small GEMM
    beta.ss_update(SExx, SEyx)                            n inverses of p x p: K1

Nothing of size (samples, n, p, p) is formed (the reference broadcasts exactly that, :62,:70-74).
"""
import math

import torch

from .. import ops
from .._common import resolve, rows_matmul
from ..dists.MVN_ard import MVN_ard
from ..dists.MultivariateNormal_vector_format import MultivariateNormal_vector_format


def _stick_counts(Y):
    """b_k = number of trials of logit k (everything not taken by the classes before k), y_k - b_k/2 (ref :49-50)"""
    # N_k = sum_{j >= k} y_j, written as a product with a triangular 0/1 matrix: torch's scan along a short innermost
    # axis costs 1.3 ms per 1e6 rows on this device, the (rows x 8) @ (8 x 8) product a few microseconds
    c = Y.shape[-1]
    tri = torch.tril(torch.ones(c, c, device=Y.device, dtype=Y.dtype))
    N = rows_matmul(Y, tri)
    return N[..., :-1], (Y - N / 2.0)[..., :-1]


def _ew(pgb, pgc):
    return pgb / 2.0 / pgc * (pgc / 2.0).tanh()


class MultiNomialLogisticRegression():
    def __init__(self, n, p, batch_shape=(), pad_X=True, device=None, dtype=None):
        if pad_X is True:
            p = p + 1
        n = n - 1
        self.n, self.p = n, p
        self.device, self.dtype = resolve(device, dtype)
        batch_shape = tuple(batch_shape)
        self.beta = MVN_ard(event_shape=(n, p, 1), batch_shape=batch_shape, device=self.device, dtype=self.dtype)
        self.beta.mu = torch.randn_like(self.beta.mu) / math.sqrt(self.p)
        self.pad_X = pad_X
        self.batch_shape = batch_shape
        self.batch_dim = len(batch_shape)
        self.event_shape = (n, p)
        self.event_dim = 2
        self.ELBO_last = torch.full((), -torch.inf, device=self.device, dtype=self.dtype)

    def to_event(self, n):
        if n < 1:
            return self
        self.event_dim = self.event_dim + n
        self.batch_dim = self.batch_dim - n
        self.event_shape = self.batch_shape[-n:] + self.event_shape
        self.batch_shape = self.batch_shape[:-n]
        self.beta.to_event(n)
        return self

    # ------------------------------------------------------------------ helpers
    def _pad(self, X):
        if self.pad_X is True:
            return torch.cat((X, torch.ones(tuple(X.shape[:-1]) + (1,), device=X.device, dtype=X.dtype)), dim=-1)
        return X

    def _mat_batch(self):
        return tuple(self.beta.mu.shape[:-2])  # batch (+ event batch) + (n,)

    def _pgc(self, EX):
        """sqrt(x' E[beta beta'] x) for every (sample, logit); EX: sample + batch + (p,)  ->  sample + batch + (n,)"""
        B = self.beta.EXXT()
        mb = self._mat_batch()
        zero_b = torch.zeros(mb + (self.p,), device=self.device, dtype=self.dtype)
        zero_c = torch.zeros(mb, device=self.device, dtype=self.dtype)
        return ops.quadform_loglike(EX.unsqueeze(-2), -2.0 * B, zero_b, zero_c).sqrt()

    def _pgc_moments(self, EXXT):
        """sqrt(sum_ij E[beta beta']_ij E[x x']_ij); EXXT: sample + batch + (p,p) -> sample + batch + (n,)"""
        B = self.beta.EXXT()
        if self.batch_dim == 0 and B.ndim == 3:
            lead = tuple(EXXT.shape[:-2])
            return (EXXT.reshape(-1, self.p * self.p) @ B.reshape(self.n, -1).transpose(0, 1)).reshape(lead + (self.n,)).sqrt()
        return (B * EXXT.unsqueeze(-3)).sum(-1).sum(-1).sqrt()

    def _padded_moments(self, pX):
        EX, EXXT = pX.mean(), pX.EXXT()
        if self.pad_X is True:
            EXXT = torch.cat((EXXT, EX), dim=-1)
            EX = torch.cat((EX, torch.ones(tuple(EX.shape[:-2]) + (1, 1), device=EX.device, dtype=EX.dtype)), dim=-2)
            EXXT = torch.cat((EXXT, EX.transpose(-2, -1)), dim=-2)
        return EX, EXXT

    # ------------------------------------------------------------------ updates
    def raw_update(self, X, Y, iters=2, p=None, lr=1.0, beta=None, verbose=False):
        """X: sample + batch + (p,), Y: sample + batch + (n+1,) counts / probabilities (ref :43-80)"""
        nsd = X.ndim - (self.event_dim + self.batch_dim - 1)
        sample_dims = tuple(range(nsd))
        pgb, YmN = _stick_counts(Y)
        EX = self._pad(X)
        w = None if p is None else p.reshape(tuple(p.shape) + (1,))
        YmNw = YmN if w is None else YmN * w
        mb = self._mat_batch()
        # SEyx_k = sum_s (y_sk - b_sk/2) x_s: the first-moment output of K4 with the weights y - b/2 (a library GEMM
        # with the sample axis as its inner dimension runs on one workgroup: 30 ms at 1e6 samples in fp64)
        SEyx = ops.weighted_moments(EX.unsqueeze(-2), YmNw, nsd, mb)[1].unsqueeze(-1)
        for i in range(iters):
            pgc = self._pgc(EX)
            Ew = _ew(pgb, pgc)
            _, _, SExx = ops.weighted_moments(EX.unsqueeze(-2), Ew if w is None else Ew * w, nsd, mb)
            if verbose is True:
                ELBO = (SEyx * self.beta.mean()).sum((-3, -2, -1)) - (pgb * (0.5 * pgc).cosh().log()).sum(sample_dims).sum(-1) \
                    - pgb.sum(sample_dims).sum(-1) * math.log(2.0) - self.KLqprior()
                print("MNLR Percent Change in ELBO: ", ((ELBO - self.ELBO_last) / self.ELBO_last.abs() * 100))
                self.ELBO_last = ELBO
            self.beta.ss_update(SExx, SEyx, lr=lr, beta=beta)

    def update(self, pX, pY, iters=2, p=None, lr=1, beta=None, verbose=False):
        """pX: distribution with mean()/EXXT() in vector format, pY: sample + batch + (n+1,) (ref :82-118)"""
        nsd = pX.mean().ndim - 2 - self.batch_dim
        sample_dims = tuple(range(nsd))
        pgb, YmN = _stick_counts(pY)
        EX, EXXT = self._padded_moments(pX)
        w = None if p is None else p.reshape(tuple(p.shape) + (1,))
        YmNw = YmN if w is None else YmN * w
        flat = self.batch_dim == 0
        if flat:
            # one weight column per logit, the sample axis reduced by K5b (as a library GEMM it is the inner dimension:
            # tens of milliseconds in fp64 at 1e5..1e6 samples)
            SEyx = ops.weighted_matsum_cols(EX.reshape(-1, self.p), YmNw.reshape(-1, self.n)).unsqueeze(-1)
        else:
            SEyx = (YmNw.reshape(tuple(YmNw.shape) + (1, 1)) * EX.unsqueeze(-3)).sum(sample_dims)
        for i in range(iters):
            pgc = self._pgc_moments(EXXT)
            Ew = _ew(pgb, pgc)
            Eww = Ew if w is None else Ew * w
            if flat:
                SExx = ops.weighted_matsum_cols(EXXT.reshape(-1, self.p, self.p), Eww.reshape(-1, self.n))
            else:
                SExx = (Eww.reshape(tuple(Eww.shape) + (1, 1)) * EXXT.unsqueeze(-3)).sum(sample_dims)
            if verbose is True:
                ELBO = (SEyx * self.beta.mean()).sum((-3, -2, -1)) - (pgb * (0.5 * pgc).cosh().log()).sum(sample_dims).sum(-1) \
                    - pgb.sum(sample_dims).sum(-1) * math.log(2.0) - self.KLqprior()
                print("MNLR Percent Change in ELBO: ", ((ELBO - self.ELBO_last) / self.ELBO_last.abs() * 100))
                self.ELBO_last = ELBO
            self.beta.ss_update(SExx, SEyx, lr=lr, beta=beta)

    # ------------------------------------------------------------------ likelihoods
    def Elog_like(self, X, Y):
        """lower bound on log p(Y | X) per sample (ref :176-192)"""
        pgb, YmN = _stick_counts(Y)
        EX = self._pad(X)
        psi = (EX.unsqueeze(-2) * self.beta.mean().squeeze(-1)).sum(-1)  # sample + batch + (n,)
        pgc = self._pgc(EX)
        return (YmN * psi).sum(-1) - (pgb * (0.5 * pgc).cosh().log()).sum(-1) - pgb.sum(-1) * math.log(2.0)

    def Elog_like_given_pX_pY(self, pX, Y):
        """same with a Gaussian input (vector format) (ref :157-174)"""
        EX, EXXT = self._padded_moments(pX)
        pgb, YmN = _stick_counts(Y)
        psi = (EX.unsqueeze(-3).squeeze(-1) * self.beta.mean().squeeze(-1)).sum(-1)
        pgc = self._pgc_moments(EXXT)
        return (YmN * psi).sum(-1) - (pgb * (0.5 * pgc).cosh().log()).sum(-1) - pgb.sum(-1) * math.log(2.0)

    def _class_logits(self, psi, pgc):
        """bound on the log-probability of every class from the n logits' means psi and root second moments pgc
        (both lead + (n,)).  The reference evaluates Elog_like on the n+1 one-hot targets (:229-239), i.e. it broadcasts
        every sample against every class; with one-hot targets the stick-breaking counts are constants, so the same
        numbers are two small GEMMs against (n+1) x n constant matrices."""
        pgb_c, YmN_c = _stick_counts(torch.eye(self.n + 1, device=self.device, dtype=self.dtype))
        lc = (0.5 * pgc).cosh().log()
        return rows_matmul(psi, YmN_c.transpose(0, 1)) - rows_matmul(lc, pgb_c.transpose(0, 1)) - pgb_c.sum(-1) * math.log(2.0)

    def log_predict(self, X):
        """log-probability bound of every class: sample + batch + (n+1,) (ref :229-235)"""
        EX = self._pad(X)
        psi = (EX.unsqueeze(-2) * self.beta.mean().squeeze(-1)).sum(-1) if self.batch_dim > 0 else \
            rows_matmul(EX, self.beta.mean().squeeze(-1).transpose(-2, -1))
        return self._class_logits(psi, self._pgc(EX))

    def log_forward(self, pX):
        EX, EXXT = self._padded_moments(pX)
        psi = (EX.unsqueeze(-3).squeeze(-1) * self.beta.mean().squeeze(-1)).sum(-1)
        return self._class_logits(psi, self._pgc_moments(EXXT))

    def loggeomean(self, X):
        return self.log_predict(X)

    def log_predict_1(self, X):
        """ref :268-280"""
        X = self._pad(X)
        lnpsb = rows_matmul(X, self.beta.mean().squeeze(-1).transpose(-2, -1))
        pgc = self._pgc(X)
        lnpsb_N = - (0.5 * pgc).cosh().log() - math.log(2.0)
        lnpsb_0 = -0.5 * lnpsb.sum(-1, True) + lnpsb_N.sum(-1, True)
        lnpsb = lnpsb - 0.5 * lnpsb.cumsum(-1) + lnpsb_N.cumsum(-1)
        return torch.cat((lnpsb, lnpsb_0), dim=-1)

    def log_predict_2(self, X):
        """betas marginalised exactly under the expected Polya-Gamma weight (ref :241-266)"""
        X = self._pad(X)
        psi_bar = (X.unsqueeze(-2) * self.beta.mean().squeeze(-1)).sum(-1)
        pgc = self._pgc(X)
        Ew = 0.5 / pgc * (0.5 * pgc).tanh()
        Xc = X.unsqueeze(-2).unsqueeze(-1)
        psi_var = (Xc * (self.beta.ESigma() @ Xc)).sum(-1).sum(-1)
        nat1_plus = 0.5 + psi_bar / psi_var
        nat1_minus = nat1_plus - 1.0
        nat2 = Ew + 1.0 / psi_var
        Res = (0.5 * pgc).cosh().log()
        lnpsb = 0.5 * nat1_plus.pow(2) / nat2 - 0.5 * nat2.log() - 0.5 * psi_bar.pow(2) / psi_var - 0.5 * psi_var.log() \
            - math.log(2.0) + Res
        lnpsb_minus = lnpsb + 0.5 * (nat1_minus.pow(2) - nat1_plus.pow(2)) / nat2
        lnp = torch.zeros(tuple(lnpsb.shape[:-1]) + (lnpsb.shape[-1] + 1,), device=self.device, dtype=self.dtype)
        lnp[..., 1:] = lnpsb_minus.cumsum(-1)
        lnp[..., :-1] = lnp[..., :-1] + lnpsb
        return lnp

    def predict(self, X):
        return torch.softmax(self.log_predict(X), -1)

    def forward(self, pX):
        return torch.softmax(self.log_forward(pX), -1)

    def predict_2(self, X):
        return torch.softmax(self.log_predict_2(X), -1)

    def ELBO(self, X=None, Y=None):
        if X is not None:
            return self.Elog_like(X, Y).sum() - self.KLqprior()
        return self.ELBO_last

    def KLqprior(self):
        KL = self.beta.KLqprior()
        for i in range(self.event_dim - 2):
            KL = KL.sum(-1)
        return KL

    def weights(self):
        if self.pad_X is True:
            mu = self.beta.mean()[..., :-1, 0]
        else:
            mu = self.beta.mean()[..., 0]
        return 2 * mu - mu.cumsum(-2)

    # ------------------------------------------------------------------ message to the input
    def backward(self, pY, like_X=None):
        if like_X is None:
            p = self.p - self.pad_X
            lead = (pY.ndim - 1) * (1,)
            like_X = MultivariateNormal_vector_format(
                invSigmamu=torch.zeros(lead + (p, 1), device=self.device, dtype=self.dtype),
                invSigma=torch.eye(p, device=self.device, dtype=self.dtype).expand(lead + (p, p)))
        invSigma, invSigmamu, Sigma, mu, Res = self.Elog_like_X(like_X, pY)
        return MultivariateNormal_vector_format(invSigma=invSigma, invSigmamu=invSigmamu, Sigma=Sigma, mu=mu), Res

    def Elog_like_X(self, like_X, pY, iters=2):
        """Gaussian message to x given class probabilities pY and a Gaussian likelihood term like_X (ref :201-227);
        the p x p inverses of every sample are one K1 launch per sweep"""
        pgb, YmN = _stick_counts(pY)
        BBT = self.beta.EXXT()
        bm = self.beta.mean()
        pgc = BBT.sum(-1).sum(-1).sqrt()
        Ew = _ew(pgb, pgc)
        v = lambda t: t.reshape(tuple(t.shape) + (1, 1))  # noqa: E731
        for i in range(iters):
            if self.pad_X is True:
                invSigmamu = (v(YmN) * bm[..., :-1, -1:] - v(Ew) * BBT[..., :-1, -1:]).sum(-3)
                invSigmamu = like_X.EinvSigmamu() + invSigmamu
                invSigma = like_X.EinvSigma() + (v(Ew) * BBT[..., :-1, :-1]).sum(-3)
                Sigma = ops.spd_inverse(invSigma)
                mu = Sigma @ invSigmamu
                pgc = ((BBT[..., :-1, :-1] * (Sigma + mu @ mu.transpose(-1, -2)).unsqueeze(-3)).sum(-1).sum(-1)
                       + 2 * (BBT[..., -1:, :-1] @ mu.unsqueeze(-3)).squeeze(-1).squeeze(-1) + BBT[..., -1, -1]).sqrt()
            else:
                invSigmamu = like_X.EinvSigmamu() + (v(YmN) * bm).sum(-3)
                invSigma = like_X.EinvSigma() + (v(Ew) * BBT).sum(-3)
                Sigma = ops.spd_inverse(invSigma)
                mu = Sigma @ invSigmamu
                pgc = ((BBT * (Sigma + mu @ mu.transpose(-1, -2)).unsqueeze(-3)).sum(-1).sum(-1)).sqrt()
            Ew = _ew(pgb, pgc)
        if self.pad_X is True:
            Res = - pgb.sum(-1) * math.log(2.0) + (YmN * ((bm[..., -1:, :-1] * mu.unsqueeze(-3)).sum(-1).sum(-1)
                                                           + bm[..., -1, -1])).sum(-1)
        else:
            Res = - pgb.sum(-1) * math.log(2.0) + (YmN * ((bm * mu.unsqueeze(-3)).sum(-1).sum(-1))).sum(-1)
        Res = Res - (pgb * (0.5 * pgc).cosh().log()).sum(-1) + like_X.Res()
        return invSigma, invSigmamu, Sigma, mu, Res
