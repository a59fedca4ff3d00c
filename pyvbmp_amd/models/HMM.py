"""Discrete hidden Markov model over the trailing batch axis of an observation node (surface of the
reference's models/HMM.py:5-178).  SURVEY.md 8(f) row 2: the step on either side of MatrixNormalWishart.update /
Elog_like_X inside DynamicMarkovBlanketDiscovery.

The log-space forward-backward recursion (:72-105) is sequential in time and tiny per step (K x K
log-sum-exp per chain): it is ONE persistent HIP launch (K11, csrc/k_hmm.hip) with the chains across lane
groups; the observation node's likelihoods and updates are the HIP kernels of that node.
"""
import torch

from .. import ops
from ..dists.Dirichlet import Dirichlet


class HMM():
    def __init__(self, obs_dist, transition_mask=None, ptemp=1.0):
        self.obs_dist = obs_dist
        self.device, self.dtype = obs_dist.device, obs_dist.dtype
        self.event_dim = 1
        self.dim = obs_dist.batch_shape[-1]
        self.event_shape = tuple(obs_dist.batch_shape[-1:])
        self.batch_shape = tuple(obs_dist.batch_shape[:-1])
        self.batch_dim = len(self.batch_shape)
        self.transition_mask = transition_mask
        kw = {"device": self.device, "dtype": self.dtype}
        alpha = torch.eye(self.dim, **kw) + 0.5
        if transition_mask is not None:
            alpha = alpha * transition_mask.to(self.device)
        self.transition = Dirichlet(self.event_shape, self.batch_shape + self.event_shape, prior_parms={'alpha': alpha}, **kw)
        self.initial = Dirichlet(self.event_shape, self.batch_shape, **kw)
        self.sumlogZ = -torch.inf
        self.p = None
        self.ptemp = ptemp
        self.logZ = torch.tensor(-torch.inf, **kw)
        self.ELBO_last = torch.tensor(-torch.inf, **kw)

    def forward_backward_logits(self, fw_logits):
        """Smoothed state posteriors and the time-integrated pair statistics from observation logits
        (time first).  ref models/HMM.py:72-105.  Returns p, SEzz, SEz0, logZ."""
        trans = self.transition.loggeomean()
        init = self.initial.loggeomean()
        if self.dim > ops.L.HMM_MAX_K:
            # every model of the reference stays far below (25 roles in the flocking DMBD); a host loop over time in torch ops
            # would be the silent slow path this package does not have
            raise ops.L.VbmpHipError(f"HMM with {self.dim} states: the forward-backward kernel (K11) holds a chain's transition "
                                     f"column in registers and serves up to VBMP_HMM_MAX_K = {ops.L.HMM_MAX_K} states")
        # ONE persistent launch (K11) instead of two host loops over time
        return ops.hmm_forward_backward(fw_logits, trans, init, self.batch_shape, self.ptemp)

    def assignment_pr(self):
        return self.p

    def assignment(self):
        return self.p.argmax(-1)

    def obs_logits(self, X, t=None):
        k = -1 - self.obs_dist.event_dim
        return self.obs_dist.Elog_like((X if t is None else X[t]).unsqueeze(k))

    def update_states(self, X, T=None):
        self.p, SEzz, SEz0, logZ = self.forward_backward_logits(self.obs_logits(X))
        NA = self.p.sum(0)
        sd = tuple(range(NA.ndim - self.batch_dim - self.event_dim))
        if sd:
            NA, SEzz, SEz0, logZ = NA.sum(sd), SEzz.sum(sd), SEz0.sum(sd), logZ.sum(sd)
        return SEzz, SEz0, NA, logZ

    def update_markov_parms(self, SEzz, SEz0, lr=1.0, beta=None):
        self.transition.ss_update(SEzz, lr=lr, beta=beta)
        self.initial.ss_update(SEz0, lr=lr, beta=beta)

    def update_obs_parms(self, X, lr=1.0, beta=None):
        self.obs_dist.raw_update(X.unsqueeze(-1 - self.obs_dist.event_dim), p=self.p, lr=lr, beta=beta)

    def update(self, X, iters=1, T=None, lr=1.0, beta=None, verbose=False):
        for i in range(iters):
            SEzz, SEz0, self.NA, self.logZ = self.update_states(X, T)
            self.KLqprior_last = self.KLqprior()
            self.update_markov_parms(SEzz, SEz0, lr=lr, beta=beta)
            self.update_obs_parms(X, lr=lr, beta=beta)
            ELBO = self.ELBO()
            if verbose:
                print('Percent Change in ELBO = ', ((ELBO - self.ELBO_last) / torch.abs(self.ELBO_last) * 100))
            self.ELBO_last = ELBO

    def KLqprior(self):
        return self.obs_dist.KLqprior().sum(-1) + self.transition.KLqprior().sum(-1) + self.initial.KLqprior()

    def ELBO(self):
        return self.logZ - self.KLqprior()

    def average(self, A, keepdim=False):
        return (A * self.p).sum(-1, keepdim)

    def event_average(self, A, keepdim=False):
        ded = self.obs_dist.event_dim
        out = (A * self.p.reshape(tuple(self.p.shape) + (1,) * ded)).sum(-ded - 1, keepdim)
        for i in range(self.event_dim - 1):
            out = out.sum(-ded - 1, keepdim)
        return out

    def average_f(self, function_string, keepdim=False):
        return self.average(getattr(self.obs_dist, function_string)(), keepdim)

    def event_average_f(self, function_string, keepdim=False):
        return self.event_average(getattr(self.obs_dist, function_string)(), keepdim)
