"""Dynamic Markov blanket discovery: a masked linear dynamical system (environment s / boundary b / internal z
latents) whose observation model is a role-HMM over MatrixNormalWishart emissions.  Surface of the reference's
models/DynamicMarkovBlanketDiscovery.py:13-342 (plotting helpers excluded): constructor, update, update_assignments,
update_obs_parms, update_latents, ELBO, KLqprior, assignment_pr / assignment / particular_assignment*, the mask
builders.

Everything heavy runs on the kernels of the nodes it is built from: the LDS smoother (K9, per-(t, series)
likelihood precision because it depends on the role assignments), MatrixNormalWishart.update (K4 moments),
Elog_like_given_pX_pY (K3a), MatrixNormalGamma / NormalInverseWishart M-steps (K1/K2).
"""
import time

import torch

from ..dists.MultivariateNormal_vector_format import MultivariateNormal_vector_format
from ..dists.NormalInverseWishart import NormalInverseWishart
from ..transforms.MatrixNormalGamma import MatrixNormalGamma
from .._common import resolve
from .ARHMM import ARHMM_prXRY
from .LinearDynamicalSystems import LinearDynamicalSystems, _LOG2PI


class DynamicMarkovBlanketDiscovery(LinearDynamicalSystems):
    def __init__(self, obs_shape, role_dims, hidden_dims, control_dim=0, regression_dim=0, batch_shape=(),
                 number_of_objects=1, unique_obs=False, device=None, dtype=None):
        # NB: like the reference (:14-96) this does not run the parent constructor
        self.device, self.dtype = resolve(device, dtype)
        kw = {"device": self.device, "dtype": self.dtype}
        control_dim = control_dim + 1
        regression_dim = regression_dim + 1
        obs_shape, batch_shape = tuple(obs_shape), tuple(batch_shape)
        obs_dim, n_obs = obs_shape[-1], obs_shape[0]
        if number_of_objects > 1:
            hidden_dim = hidden_dims[0] + number_of_objects * (hidden_dims[1] + hidden_dims[2])
            role_dim = role_dims[0] + number_of_objects * (role_dims[1] + role_dims[2])
            A_mask, B_mask, role_mask = self.n_object_mask(number_of_objects, hidden_dims, role_dims, control_dim, obs_dim,
                                                           regression_dim)
        else:
            hidden_dim, role_dim = sum(hidden_dims), sum(role_dims)
            A_mask, B_mask, role_mask = self.one_object_mask(hidden_dims, role_dims, control_dim, obs_dim, regression_dim)
        A_mask, B_mask, role_mask = A_mask.to(self.device), B_mask.to(self.device), role_mask.to(self.device)
        self.A_mask, self.B_mask, self.role_mask = A_mask, B_mask, role_mask
        self.number_of_objects = number_of_objects
        self.unique_obs = unique_obs
        self.obs_shape, self.obs_dim, self.event_dim, self.n_obs = obs_shape, obs_dim, len(obs_shape), n_obs
        self.role_dims, self.role_dim = role_dims, role_dim
        self.hidden_dims, self.hidden_dim = hidden_dims, hidden_dim
        self.control_dim, self.regression_dim = control_dim, regression_dim
        self.batch_shape, self.batch_dim = batch_shape, len(batch_shape)
        self.latent_noise = 'independent'
        self.expand_to_batch = True
        self.offset = (1,) * (len(obs_shape) - 1)
        self.logZ = -torch.tensor(torch.inf, **kw)
        self.ELBO_save = -torch.inf * torch.ones(1, **kw)
        self.iters = 0
        self.px = None
        self.ELBO_last = -torch.tensor(torch.inf, **kw)

        self.x0 = NormalInverseWishart(self.offset + (hidden_dim,), batch_shape, **kw)
        self.x0.mu = torch.zeros_like(self.x0.mu)
        self.A = MatrixNormalGamma(self.offset + (hidden_dim, hidden_dim + control_dim), batch_shape, mask=A_mask,
                                   pad_X=False, uniform_precision=False, **kw)
        if unique_obs is True:
            raise NotImplementedError("unique_obs=True needs HMM.to_event, which the reference's HMM does not define either")
        self.obs_model = ARHMM_prXRY(role_dim, obs_dim, hidden_dim, regression_dim, batch_shape=batch_shape,
                                     X_mask=B_mask.sum(-2, True) > 0, transition_mask=role_mask, pad_X=False, **kw)
        self.B = self.obs_model.obs_dist
        W = self.B.invU  # flatter prior on the emission noise (ref :81-84)
        W.invU_0 = W.invU_0 / float(role_dim ** 2)
        W.invU = W.invU_0.clone()  # NB: like the reference, the cached inverse W.U is left as constructed
        shift = -self.B.n * torch.log(torch.tensor(float(role_dim ** 2), **kw))
        W.logdet_invU_0 = W.logdet_invU_0 + shift
        W.logdet_invU = W.logdet_invU_0.clone()
        self.B.ptemp = 20.0
        self.set_latent_parms()
        self.log_like = -torch.tensor(torch.inf, **kw)
        self.log2pi = torch.tensor(_LOG2PI, **kw)
        self.reducer = None  # SuffStatReducer when the series are sharded over ranks (BASELINE config 5)
        self._role_entropy = None

    # ------------------------------------------------------------------ likelihood of the latents
    def log_likelihood_function(self, Y, R):
        """natural parameters of p(y_t | x_t) averaged over the role posterior, summed over observables (ref :98-104)"""
        k = self.obs_model.event_dim + 2
        # (the sum over the observables rides along: for the shared role precisions it is taken on the weights, before the GEMM)
        if self.obs_model.p is not None and self.obs_model.p.ndim >= k - 1:
            return self.obs_model.Elog_like_X((Y.unsqueeze(-k), R.unsqueeze(-k)), sum_axis=-(k - 1))
        P, eta, Res = self.obs_model.Elog_like_X((Y.unsqueeze(-k), R.unsqueeze(-k)))
        return P.sum(-k, True), eta.sum(-k, True), Res.sum(-k + 2, True)

    def KLqprior(self):
        KL = self.x0.KLqprior() + self.A.KLqprior()
        for i in range(len(self.offset)):
            KL = KL.squeeze(-1)
        return KL + self.obs_model.KLqprior()

    # ------------------------------------------------------------------ role assignments
    def _px_for_roles(self, r):
        h = self.hidden_dim
        tgt = tuple(r.shape[:-2])
        px = self.px
        return MultivariateNormal_vector_format(mu=px.mu.expand(tgt + (h, 1)), Sigma=px.Sigma.expand(tgt + (h, h)),
                                                invSigmamu=px.invSigmamu.expand(tgt + (h, 1)),
                                                invSigma=px.invSigma.expand(tgt + (h, h)))

    def update_assignments(self, y, r):
        """role posteriors + the Markov statistics of the role chain (ref :113-132)"""
        if self.px is None:
            h = self.hidden_dim
            lead = tuple(r.shape[:-3]) + (1,)
            eye = torch.eye(h, device=r.device, dtype=r.dtype)
            zeros = torch.zeros(lead + (h, 1), device=r.device, dtype=r.dtype)
            self.px = MultivariateNormal_vector_format(mu=zeros, Sigma=eye.expand(lead + (h, h)), invSigmamu=zeros.clone(),
                                                       invSigma=eye.expand(lead + (h, h)))
        k = self.obs_model.event_dim + 2
        px4r = self._px_for_roles(r).unsqueeze(-k)
        self.SEzz, self.SEz0, self.NA, logZ = self.obs_model.update_states((px4r, r.unsqueeze(-k), y.unsqueeze(-k)))

    def update_obs_parms(self, y, r, lr=1.0):
        self.obs_model.update_markov_parms(self.SEzz, self.SEz0, lr)
        k = self.obs_model.event_dim + 2
        self.obs_model.update_obs_parms((self._px_for_roles(r).unsqueeze(-k), r.unsqueeze(-k), y.unsqueeze(-k)), lr)

    def assignment_pr(self):
        pr = self.obs_model.assignment_pr()
        r0, r1, r2 = self.role_dims[0], self.role_dims[1], self.role_dims[2]
        parts = [pr[..., :r0].sum(-1, True)]
        for n in range(self.number_of_objects):
            s = r0 + n * (r1 + r2)
            parts += [pr[..., s:s + r1].sum(-1, True), pr[..., s + r1:s + r1 + r2].sum(-1, True)]
        return torch.cat(parts, dim=-1)

    def particular_assignment_pr(self):
        p_sbz = self.assignment_pr()
        parts = [p_sbz[..., :1]]
        for n in range(self.number_of_objects):
            parts.append(p_sbz[..., n + 1:n + 3].sum(-1, True))
        return torch.cat(parts, dim=-1)

    def particular_assignment(self):
        return self.particular_assignment_pr().argmax(-1)

    def assignment(self):
        return self.assignment_pr().argmax(-1)

    # ------------------------------------------------------------------ VB loop
    def update_latent_parms(self, p=None, lr=1.0):
        self.ss_update(p=None, lr=lr)

    def update_latents(self, y, u, r, p=None, lr=1.0):
        if self.obs_model.p is None:
            pr = torch.ones(tuple(y.shape[:-2]) + (self.role_dim,), device=y.device, dtype=y.dtype)
            self.obs_model.p = pr / pr.sum(-1, True)
        super().update_latents(y, u, r, p=None, lr=lr)

    def Elog_like(self, y, u, r, latent_iters=1, lr=1.0):
        y, u, r = self.reshape_inputs(y, u, r)
        self.px = None
        self.obs_model.p = None
        for i in range(latent_iters):
            self.update_assignments(y, r)
            self.update_latents(y, u, r)
        return self.logZ - (self.obs_model.p * (self.obs_model.p + 1e-8).log()).sum(0).sum((-1, -2))

    def update(self, y, u, r, iters=1, latent_iters=1, lr=1.0, verbose=False, graphed=False):
        """VB iterations (ref models/DynamicMarkovBlanketDiscovery.py:185-211).  graphed=True replays the iteration as ONE HIP
        graph (pyvbmp_amd.graph): at the reference's flocking sizes an iteration is ~470 launches, a third of its time their
        launch cost.  (Round 1 found the replay slower than the eager loop -- 86.9 vs 63.4 ms -- when an iteration still was
        ~10^4 launches.)  Not with verbose (printing synchronises), a multi-rank reducer (the collective is not captured) or
        latent_iters > 1 (the posterior is re-created inside the iteration)."""
        if graphed and not verbose and latent_iters == 1 and (self.reducer is None or self.reducer.world_size == 1):
            from .. import graph
            key = tuple((t.data_ptr(), tuple(t.shape), t.dtype) if t is not None else None for t in (y, u, r)) + (float(lr),)
            graph.run_iterations(self, lambda m: m._vb_iteration(*m.reshape_inputs(y, u, r), 1, lr), iters, key,
                                 post=lambda m: m._after_iteration())
            return
        y, u, r = self.reshape_inputs(y, u, r)
        for i in range(iters):
            t = time.time()
            ELBO_last = self.ELBO_last
            self._vb_iteration(y, u, r, latent_iters, lr)
            if verbose is True:
                print('Percent Change in ELBO = ', ((self.ELBO_last - ELBO_last) / ELBO_last.abs()) * 100,
                      '   Iteration Time = ', (time.time() - t))
            self._after_iteration()

    def _after_iteration(self):
        self.iters = self.iters + 1
        self.ELBO_save = torch.cat((self.ELBO_save, self.ELBO_last * torch.ones(1, device=self.device, dtype=self.dtype)),
                                   dim=-1)

    def _vb_iteration(self, y, u, r, latent_iters, lr):
        for j in range(latent_iters - 1):
            self.px = None
            self.update_assignments(y, r)
            self.update_latents(y, u, r)
        self.update_assignments(y, r)
        if self.reducer is not None:
            # series sharded over ranks: two exchange steps per iteration, each ONE flat all-reduce
            self._update_obs_parms_sharded(y, r, lr)
            self.update_latents(y, u, r)
            om = self.obs_model
            idx = om.p > 1e-8
            ent = -(om.p[idx].log() * om.p[idx]).sum()
            self._role_entropy, = self.reduce_statistics(extra=[ent])
        else:
            self._role_entropy = None
            self.update_obs_parms(y, r, lr=lr)
            self.update_latents(y, u, r)
        ELBO = self.ELBO()
        self.update_latent_parms(p=None, lr=lr)
        self.ELBO_last = ELBO

    def _update_obs_parms_sharded(self, y, r, lr):
        """update_obs_parms when each rank holds a slice of the series: the Markov statistics of the role chain and
        the emission moments cross the ranks in one packed all-reduce, then every rank applies the same update"""
        om, k = self.obs_model, self.obs_model.event_dim + 2
        XRY = (self._px_for_roles(r).unsqueeze(-k), r.unsqueeze(-k), y.unsqueeze(-k))
        B = om.obs_dist
        pXR = om._joint_input(XRY)
        SExx, SEyx, SEyy, N = B._moments(pXR.EX(), XRY[2], pXR.ESigma(), None, om.p)
        self.SEzz, self.SEz0, self.NA, SExx, SEyx, SEyy, N = self.reducer.all_reduce(
            [self.SEzz, self.SEz0, self.NA, SExx, SEyx, SEyy, N])
        om.update_markov_parms(self.SEzz, self.SEz0, lr)
        B.ss_update(SExx, SEyx, SEyy, N, lr=lr, beta=None)

    def ELBO(self):
        om = self.obs_model
        tl = om.transition.loggeomean()
        # masked sums written with where(): boolean-mask indexing would synchronise with the host
        zero = torch.zeros((), device=tl.device, dtype=tl.dtype)
        contrib = (torch.where(tl > -torch.inf, tl, zero) * self.SEzz).sum() + (om.initial.loggeomean() * self.SEz0).sum()
        if getattr(self, "_role_entropy", None) is not None:
            contrib = contrib + self._role_entropy
        else:
            contrib = contrib - torch.where(om.p > 1e-8, om.p.log() * om.p, zero).sum()
        return super().ELBO() + contrib

    # ------------------------------------------------------------------ masks
    @staticmethod
    def _object_blocks(n, d0, d1, d2):
        """connectivity over [s | (b z) x n]: s<->s, s<->b, and (b,z) blocks of the same object (ref :223-275)"""
        tot = d0 + n * (d1 + d2)
        m = torch.zeros(tot, tot)
        m[:d0, :d0] = 1
        for k in range(n):
            s = d0 + k * (d1 + d2)
            m[:d0, s:s + d1] = 1
            m[s:s + d1, :d0] = 1
            m[s:s + d1 + d2, s:s + d1 + d2] = 1
        return m

    def n_object_mask(self, n, hidden_dims, role_dims, control_dim, obs_dim, regression_dim):
        h0, h1, h2 = hidden_dims[0], hidden_dims[1], hidden_dims[2]
        r0, r1, r2 = role_dims[0], role_dims[1], role_dims[2]
        A = self._object_blocks(n, h0, h1, h2)
        A_mask = torch.cat((A, torch.ones(A.shape[0], control_dim)), dim=-1)
        hid, rol = h0 + n * (h1 + h2), r0 + n * (r1 + r2)
        B = torch.zeros(rol, hid)
        B[:r0, :h0] = 1
        for k in range(n):
            rs, hs = r0 + k * (r1 + r2), h0 + k * (h1 + h2)
            B[rs:rs + r1, hs:hs + h1] = 1
            B[rs + r1:rs + r1 + r2, hs + h1:hs + h1 + h2] = 1
        B = B.unsqueeze(-2).expand(rol, obs_dim, hid)
        B_mask = torch.cat((B, torch.ones(rol, obs_dim, regression_dim)), dim=-1)
        role_mask = self._object_blocks(n, r0, r1, r2)
        return A_mask > 0, B_mask > 0, role_mask > 0

    def one_object_mask(self, hidden_dims, role_dims, control_dim, obs_dim, regression_dim):
        """single object; an optional 4th hidden block is a global latent seen by every role (ref :277-342)"""
        h = list(hidden_dims)
        r0, r1, r2 = role_dims[0], role_dims[1], role_dims[2]
        hid, rol = sum(h), sum(role_dims)
        s, b, z = slice(0, h[0]), slice(h[0], h[0] + h[1]), slice(h[0] + h[1], h[0] + h[1] + h[2])
        A = torch.zeros(hid, hid)
        A[s, s] = 1
        A[s, b] = 1
        A[b, s] = 1
        A[b, b] = 1
        A[b, z] = 1
        A[z, b] = 1
        A[z, z] = 1
        if len(h) == 4:
            gl = slice(h[0] + h[1] + h[2], hid)
            A[gl, gl] = 1
        A_mask = torch.cat((A, torch.ones(hid, control_dim)), dim=-1) > 0
        B = torch.zeros(rol, obs_dim, hid)
        B[:r0, :, s] = 1
        B[r0:r0 + r1, :, b] = 1
        B[r0 + r1:r0 + r1 + r2, :, z] = 1
        if len(h) == 4:
            B[:, :, h[0] + h[1] + h[2]:] = 1
        B_mask = torch.cat((B, torch.ones(rol, obs_dim, regression_dim)), dim=-1) > 0
        R = torch.zeros(rol, rol)
        R[:r0, :r0 + r1] = 1
        R[r0:r0 + r1, :] = 1
        R[r0 + r1:, r0:] = 1
        return A_mask, B_mask, R > 0
