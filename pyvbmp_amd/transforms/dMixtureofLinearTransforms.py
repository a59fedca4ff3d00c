"""Input-gated mixture of linear transforms: `mixture_dim` MatrixNormalWishart / MatrixNormalGamma experts under a
Polya-Gamma multinomial-logistic gate (public surface of the reference's transforms/dMixtureofLinearTransforms.py:8-176,
the model of its examples/two_moons.py).

Organisation (ours): the experts live on one extra trailing batch axis of a single transform node, so every quantity
"per (sample, expert)" is ONE launch -- the joint quadratic form of z = [x; y] (K3a) for the expert evidences, the gate's
class log-probabilities beside it -- and the two learning entry points (`raw_update` on data, `update` on Gaussian
messages) are the same EM sweep over a small table of four callables (`_Sweep`).  The mixing of the experts' outputs is
factored into two reductions used by all four inference methods: moment matching of a Gaussian mixture
(`_collapse_moments`, for predict / forward) and responsibility-weighted natural parameters (`_blend_natural`, for
postdict / backward).
"""
import math
from collections import namedtuple

import torch

from .. import ops
from ..dists.MultivariateNormal_vector_format import MultivariateNormal_vector_format
from .MatrixNormalGamma import MatrixNormalGamma
from .MatrixNormalWishart import MatrixNormalWishart
from .MultiNomialLogisticRegression import MultiNomialLogisticRegression

_EXPERT_NODES = {'Wishart': MatrixNormalWishart, 'Gamma': MatrixNormalGamma}
_LOG_2PI = math.log(2.0 * math.pi)

# one EM sweep = evidence of every (sample, expert) + gate log-probabilities -> responsibilities -> gate and expert updates
_Sweep = namedtuple("_Sweep", "expert_evidence gate_logits fit_gate fit_experts")


def _softmax_with_evidence(scores, axis=-1, keepdim=False):
    """responsibilities and their normaliser over the expert axis"""
    evidence = torch.logsumexp(scores, axis, keepdim=True)
    return torch.exp(scores - evidence), (evidence if keepdim else evidence.squeeze(axis))


def _as_weights(r, trailing):
    return r.reshape(tuple(r.shape) + (1,) * trailing)


def _collapse_moments(pY, r):
    """the Gaussian with the mean and covariance of the mixture sum_k r_k N(mu_k, Sigma_k) (expert axis -3)"""
    w = _as_weights(r, 2)
    mean = (pY.mean() * w).sum(-3)
    second = (pY.EXXT() * w).sum(-3)
    return mean, second - mean @ mean.transpose(-2, -1)


def _blend_natural(P, eta, r, axis):
    """responsibility-weighted natural parameters sum_k r_k (P_k, eta_k) along the expert axis"""
    w = _as_weights(r, 2)
    return MultivariateNormal_vector_format(invSigma=(P * w).sum(axis), invSigmamu=(eta * w).sum(axis))


class dMixtureofLinearTransforms():
    def __init__(self, n, p, mixture_dim, batch_shape=(), pad_X=True, type='Wishart', fixed_precision=False, device=None,
                 dtype=None):
        if type == 'MVN_ard':
            raise NotImplementedError("MVN_ard experts are declared but not implemented by the reference either")
        if type not in _EXPERT_NODES:
            raise ValueError('type must be either Wishart (default) or Gamma')
        batch_shape = tuple(batch_shape)
        self.n, self.p, self.mix_dim = n, p, mixture_dim
        self.batch_shape, self.batch_dim = batch_shape, len(batch_shape)
        self.event_shape, self.event_dim = (mixture_dim, n, p), 3
        # experts: one node, the mixture on its last batch axis; prior scale shrinks with the number of experts
        self.A = _EXPERT_NODES[type](event_shape=(n, p), batch_shape=batch_shape + (mixture_dim,),
                                     scale=mixture_dim ** (-1.0 / n), pad_X=pad_X, fixed_precision=fixed_precision,
                                     device=device, dtype=dtype)
        self.device, self.dtype = self.A.device, self.A.dtype
        self.pi = MultiNomialLogisticRegression(mixture_dim, p, batch_shape=batch_shape, pad_X=True, device=self.device,
                                                dtype=self.dtype)
        self.ELBO_last = torch.full((), -torch.inf, device=self.device, dtype=self.dtype)
        self.ELBO_save = []

    # ------------------------------------------------------------------ learning
    def _em(self, sweep, sample_weight, iters, lr, report):
        for _ in range(iters):
            r, evidence = _softmax_with_evidence(sweep.expert_evidence() + sweep.gate_logits())
            report(r, evidence, before_fit=True)
            w = r if sample_weight is None else r * sample_weight.unsqueeze(-1)
            sweep.fit_gate(r)
            sweep.fit_experts(w)
            report(r, evidence, before_fit=False)

    def _say(self, bound):
        print("dMixtureofLinearTransforms: ELBO changed by %s %%" % (((bound - self.ELBO_last) / self.ELBO_last.abs()) * 100).tolist())

    def raw_update(self, X, Y, p=None, iters=1, lr=1.0, verbose=False):
        """EM on data pairs (ref :102-124): the bound is only evaluated (and printed) when verbose, before the fit"""
        xe, ye = X.unsqueeze(-1).unsqueeze(-3), Y.unsqueeze(-1).unsqueeze(-3)  # vector format + expert axis

        def report(r, evidence, before_fit):
            if verbose and before_fit:
                bound = evidence.sum(0) - self.KLqprior()
                self._say(bound)
                self.ELBO_last = bound
        self._em(_Sweep(expert_evidence=lambda: self.A.Elog_like(xe, ye),
                        gate_logits=lambda: self.pi.log_predict(X),
                        fit_gate=lambda r: self.pi.raw_update(X, r, p=p, lr=lr, verbose=False),
                        fit_experts=lambda w: self.A.raw_update(xe, ye, p=w, lr=lr)),
                 p, iters, lr, report)

    def update(self, pX, pY, p=None, iters=1, lr=1.0, verbose=False):
        """EM on Gaussian messages (ref :126-149): keeps logZ / NA of the last E-step and the bound after the fit"""
        qx, qy = pX.unsqueeze(-3), pY.unsqueeze(-3)

        def report(r, evidence, before_fit):
            if before_fit:
                self.logZ, self.NA = evidence, r.sum(0)
                return
            bound = self.logZ.sum() - self.KLqprior().sum()
            if verbose:
                self._say(bound)
            self.ELBO_last = bound
        self._em(_Sweep(expert_evidence=lambda: self.A.Elog_like_given_pX_pY(qx, qy),
                        gate_logits=lambda: self.pi.log_forward(pX),
                        fit_gate=lambda r: self.pi.update(pX, r, p=p, lr=lr, verbose=False),
                        fit_experts=lambda w: self.A.update(qx, qy, p=w, lr=lr)),
                 p, iters, lr, report)

    # ------------------------------------------------------------------ inference: input -> output
    def predict(self, X):
        """output distribution for observed inputs: the experts' predictions collapsed under the gate (ref :75-84)"""
        r = self.pi.predict(X)
        per_expert = self.A.predict(X.reshape(tuple(X.shape[:-1]) + (1, X.shape[-1], 1)))[0]
        mean, cov = _collapse_moments(per_expert, r)
        return MultivariateNormal_vector_format(mu=mean, Sigma=cov), r

    def forward(self, pX):
        """the same for a Gaussian input message (ref :86-93)"""
        mean, cov = _collapse_moments(self.A.forward(pX.unsqueeze(-3))[0], self.pi.forward(pX))
        return MultivariateNormal_vector_format(mu=mean, Sigma=cov)

    def forward_mix(self, pX):
        return self.A.forward(pX.unsqueeze(-3)), self.pi.forward(pX)

    # ------------------------------------------------------------------ inference: output -> input
    def postdict(self, Y):
        """message to the input for an observed output (ref :52-73): every expert's likelihood of x meets the gate's
        likelihood of x for "this expert was chosen"; the combined messages are weighted by their evidences"""
        k = self.batch_dim
        P, eta, res = self.A.Elog_like_X(Y.unsqueeze(-2).unsqueeze(-1))
        # expert axis in front of the batch axes, as the gate expects its class axis
        from_experts = MultivariateNormal_vector_format(invSigma=P.unsqueeze(0).movedim(-3, -3 - k),
                                                        invSigmamu=eta.movedim(-3, -3 - k))
        one_hot = torch.eye(self.mix_dim, device=self.device, dtype=self.dtype).reshape(
            (self.mix_dim,) + (1,) * k + (self.mix_dim,))
        P_c, eta_c, _, mean_c, res_gate = self.pi.Elog_like_X(from_experts, one_hot, iters=4)
        score = res.movedim(-1, -1 - k) + res_gate + 0.5 * (mean_c * eta_c).sum(-2).squeeze(-1) \
            - 0.5 * ops.spd_inv_logdet(P_c)[1] + 0.5 * from_experts.dim * _LOG_2PI  # K1 (batched p x p)
        r, evidence = _softmax_with_evidence(score, -1 - k, keepdim=True)
        # (the reference squeezes the trailing axis first, then the expert axis: kept, it matters for size-1 axes)
        return _blend_natural(P_c, eta_c, r, -3 - k), evidence.squeeze(-1).squeeze(-1 - k), r

    def backward(self, pY):
        """message to the input for a Gaussian output message (ref :37-50); also returns the log responsibilities"""
        from_experts, res = self.A.backward(pY.unsqueeze(-3))
        combined, res_gate = self.pi.backward(torch.eye(self.mix_dim, device=self.device, dtype=self.dtype), from_experts)
        score = res + res_gate
        log_r = score - score.logsumexp(-1, True)
        return _blend_natural(combined.EinvSigma(), combined.EinvSigmamu(), log_r.exp(), -3), log_r

    # ------------------------------------------------------------------ evidences
    def Elog_like_given_pX_pY(self, pX, pY):
        return (self.A.Elog_like_given_pX_pY(pX.unsqueeze(-3), pY.unsqueeze(-3)) + self.pi.log_forward(pX)).logsumexp(-1)

    def Elog_like(self, X, Y):
        return (self.A.Elog_like(X.unsqueeze(-1).unsqueeze(-3), Y.unsqueeze(-1).unsqueeze(-3))
                + self.pi.log_predict(X)).logsumexp(-1)

    def KLqprior(self):
        return self.A.KLqprior().sum(-1) + self.pi.KLqprior()
