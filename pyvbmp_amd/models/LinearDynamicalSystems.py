"""Variational-Bayes linear dynamical system
       x_t = A [x_{t-1}; u_t] + eta_t ,   y_t = B [x_t; r_t] + eps_t
with the same surface as the reference model (models/LinearDynamicalSystems.py:14-383): constructor,
reshape_inputs, update, ss_update, update_latents, forward_backward_loop, ELBO, KLqprior, attributes
px (MultivariateNormal_vector_format), SE_*, T, N, logZ.

E-step: the whole information filter + smoother (the reference's Python loops over T, :358-381) is ONE
persistent HIP launch (K9): series-parallel, time-sequential, h x h state in registers.
M-step: x0.ss_update (K2), A.ss_update / obs_model.ss_update (MatrixNormalWishart: K1 + K2a).

latent_noise='shared' gives a MatrixNormalWishart transition, anything else the reference's default
MatrixNormalGamma (diagonal noise); both run on the same kernels.
"""
import math

import torch

from .. import ops
from .._common import derived_key, resolve, shared_matvec
from ..dists.MultivariateNormal_vector_format import MultivariateNormal_vector_format
from ..dists.NormalInverseWishart import NormalInverseWishart
from ..transforms.MatrixNormalGamma import MatrixNormalGamma
from ..transforms.MatrixNormalWishart import MatrixNormalWishart

_LOG2PI = math.log(2.0 * math.pi)


def _T(a):
    return a.transpose(-2, -1)


def _time_constant(x):
    return x.shape[0] == 1 or x.stride(0) == 0


class _TimeSums:
    """sum_t a[t] b[t]' over the time axis for the statistics of update_latents.  An operand that does not depend on
    time (the appended ones of the control / regression inputs, or their stand-ins when there are none: stride 0)
    factors out of the sum, so those moments need the time sum of the OTHER operand only -- computed once per operand
    and shared (the default model needs sum_t mu_t for three of its eight moments) -- instead of a K10 pass over both;
    everything else is one K10 launch (ops.tsum_outer)."""

    def __init__(self):
        self._sums = {}

    def _window(self, x, frm, steps):
        """sum_{t < steps} x[t + frm]"""
        if _time_constant(x):
            return steps * x[0]
        hit = self._sums.get(id(x))
        if hit is None:
            hit = self._sums[id(x)] = (x, x.sum(0))  # the operand is kept alive with its sum
        s = hit[1]
        for t in list(range(frm)) + list(range(frm + steps, x.shape[0])):
            s = s - x[t]
        return s

    def __call__(self, a, b, M=None, a_from=0, b_from=0, steps=None):
        T = max(a.shape[0], b.shape[0])
        if steps is None:
            steps = T - max(a_from, b_from)
        ca, cb = _time_constant(a), _time_constant(b)
        if M is not None or not (ca or cb) or T - steps > 4:
            return ops.tsum_outer(a, b, M=M, a_from=a_from, b_from=b_from, steps=steps)
        A = a[0] if ca else self._window(a, a_from, steps)
        B = b[0] if cb else self._window(b, b_from, steps)
        out = A.unsqueeze(-1) * B.unsqueeze(-2)
        return steps * out if (ca and cb) else out


class LinearDynamicalSystems():
    # K9's fixed-point shortcut (include/vbmp_hip.h): "auto" (outputs equal the literal recursion's to its
    # last-bit wander), "exact" (bit for bit the literal recursion), "off" (every step runs the full recursion, as the
    # reference does).  Per model (`m.fixed_point = "exact"`) or per call (forward_backward_loop(..., fixed_point=...)).
    fixed_point = "auto"

    def __init__(self, obs_shape, hidden_dim, control_dim=0, regression_dim=0, obs_model=None,
                 latent_noise='independent', batch_shape=(), A_mask=None, B_mask=None, device=None, dtype=None):
        self.device, self.dtype = resolve(device, dtype)
        control_dim = control_dim + 1
        regression_dim = regression_dim + 1
        obs_shape, batch_shape = tuple(obs_shape), tuple(batch_shape)
        self.obs_shape = obs_shape
        self.obs_dim = obs_shape[-1]
        self.hidden_dim = hidden_dim
        self.latent_noise = latent_noise
        self.batch_shape = batch_shape
        self.batch_dim = len(batch_shape)
        self.control_dim = control_dim
        self.regression_dim = regression_dim
        self.event_dim = len(obs_shape)
        self.logZ = torch.tensor(0.0, device=self.device, dtype=self.dtype)
        kw = {"device": self.device, "dtype": self.dtype}

        def pad_mask(m):
            if m is None:
                return None
            m = m.to(self.device)
            return torch.cat((m, torch.ones(tuple(m.shape[:-1]) + (1,), dtype=torch.bool, device=self.device)), dim=-1)
        A_mask, B_mask = pad_mask(A_mask), pad_mask(B_mask)
        self.offset = (1,) * (len(obs_shape) - 1)
        self.expand_to_batch = False
        self.x0 = NormalInverseWishart(self.offset + (hidden_dim,), batch_shape, **kw)
        if latent_noise == 'shared':
            self.A = MatrixNormalWishart(self.offset + (hidden_dim, hidden_dim + control_dim), batch_shape, pad_X=False,
                                         mask=A_mask, **kw)
        else:  # the reference's default: diagonal transition noise
            self.A = MatrixNormalGamma(self.offset + (hidden_dim, hidden_dim + control_dim), batch_shape, pad_X=False,
                                       mask=A_mask, **kw)
        self.obs_model = obs_model
        if obs_model is None:
            self.obs_model = MatrixNormalWishart(obs_shape + (hidden_dim + regression_dim,), batch_shape, mask=B_mask,
                                                 pad_X=False, **kw)
        self.set_latent_parms()
        self.px = None
        self.log2pi = torch.tensor(_LOG2PI, **kw)
        # set to a pyvbmp_amd.parallel.SuffStatReducer when the SERIES (sample axis) are sharded over ranks
        self.reducer = None

    # ------------------------------------------------------------------ inputs
    def reshape_inputs(self, y, u=None, r=None):
        """vector format, appended ones, optional expansion to the batch (ref :56-83)"""
        sample_shape = tuple(y.shape[:y.ndim - len(self.obs_shape)])
        y = y.unsqueeze(-1)
        one = getattr(self, "_one", None)  # one tensor for all calls: the stand-in inputs keep their address, which is
        if one is None or one.device != y.device or one.dtype != y.dtype:  # what the data-moment cache keys on
            one = self._one = torch.ones((), device=y.device, dtype=y.dtype)
        if u is None:
            u = one.expand(sample_shape + (self.control_dim, 1))
        else:
            u = torch.cat((u, u.new_ones(tuple(u.shape[:-1]) + (1,))), -1).unsqueeze(-1)
        if r is None:
            r = one.expand(sample_shape + self.obs_shape[:-1] + (self.regression_dim, 1))
        else:
            r = torch.cat((r, r.new_ones(tuple(r.shape[:-1]) + (1,))), -1).unsqueeze(-1)
        if self.expand_to_batch is True:
            k = len(sample_shape)
            for i in range(len(self.batch_shape)):
                y, u, r = y.unsqueeze(k), u.unsqueeze(k), r.unsqueeze(k)
            y = y.expand(sample_shape + self.batch_shape + self.obs_shape + (1,))
            u = u.expand(sample_shape + self.batch_shape + (self.control_dim, 1))
            r = r.expand(sample_shape + self.batch_shape + self.obs_shape[:-1] + (self.regression_dim, 1))
        for i in range(len(self.offset)):
            u = u.unsqueeze(-3)
        return y, u, r

    # ------------------------------------------------------------------ VB loop
    def update(self, y, u=None, r=None, p=None, iters=1, lr=1.0, verbose=False, graphed=False):
        """VB iterations (ref models/LinearDynamicalSystems.py:85-102).  graphed=True replays the iteration as one
        HIP graph (pyvbmp_amd.graph) -- for short / few series, where the iteration is launch-bound; not with
        verbose (printing synchronises) nor with a multi-rank reducer."""
        if graphed and not verbose and (self.reducer is None or self.reducer.world_size == 1):
            from .. import graph
            key = tuple((t.data_ptr(), tuple(t.shape), t.dtype) if t is not None else None for t in (y, u, r, p)) + (float(lr),)
            graph.run_iterations(self, lambda m: m.update(y, u, r, p, iters=1, lr=lr), iters, key)
            return
        L = torch.full((), -torch.inf, device=self.device, dtype=self.dtype)
        L_last = L
        y, u, r = self.reshape_inputs(y, u, r)
        for i in range(iters):
            L_last = L
            self.update_latents(y, u, r)
            if self.reducer is not None:
                self.reduce_statistics()
            L = self.ELBO().sum()
            self.ss_update(p=p, lr=lr)
            self.obs_model.ss_update(self.SE_xr_xr, self.SE_y_xr, self.SE_y_y, self.T, lr)
            if verbose:
                print("Percent Change in ELBO %f" % ((L - L_last) / L.abs() * 100))
        self.ELBO_last = L

    _STATS = ("SE_x0_x0", "SE_x0", "SE_xpu_xpu", "SE_x_xpu", "SE_x_x", "SE_xr_xr", "SE_y_xr", "SE_y_y")

    def ss_update(self, p=None, lr=1.0):
        """statistics (time already integrated) -> sum over samples -> M-step of x0 and A (ref :104-154)"""
        if p is not None:
            self.weight_statistics(p)
        while self.SE_x_x.ndim > self.batch_dim + len(self.offset) + 2:
            for k in self._STATS + ("T", "N"):
                setattr(self, k, getattr(self, k).sum(0))
        for k in ("SE_x0_x0", "SE_xpu_xpu", "SE_x_x", "SE_xr_xr"):
            v = getattr(self, k)
            setattr(self, k, 0.5 * (v + _T(v)))
        self.x0.ss_update(self.SE_x0_x0, self.SE_x0.squeeze(-1), self.N, lr)
        self.A.ss_update(self.SE_xpu_xpu, self.SE_x_xpu, self.SE_x_x, self.T, lr)
        self.set_latent_parms()

    def weight_statistics(self, p):
        """per-series weights on the statistics of the last E-step (the `p` branch of the reference's ss_update,
        ref :106-121); separate so that sample-sharded callers can weight, reduce over ranks, then update"""
        for i in range(len(self.offset)):
            p = p.unsqueeze(-1)
        self.T = self.T * p
        self.N = self.N * p
        p = p.unsqueeze(-1).unsqueeze(-1)
        for k in self._STATS:
            setattr(self, k, getattr(self, k) * p)

    def reduce_statistics(self, extra=()):
        """Sample-sharded runs (case 2): sum the local series' statistics, then ONE all-reduce of the
        flat packed buffer [SE_*, T, N, logZ (+ extra)] over the ranks.  Afterwards ss_update / ELBO see fully reduced
        tensors (their own sums over the sample axes become no-ops).  Returns the reduced `extra` tensors."""
        nkeep = self.batch_dim + len(self.offset)
        names = self._STATS + ("T", "N")
        local = []
        for k in names:
            v = getattr(self, k)
            extra_dims = v.ndim - nkeep - (0 if k in ("T", "N") else 2)
            local.append(v.sum(tuple(range(extra_dims))) if extra_dims > 0 else v)
        lz = self.logZ
        while lz.ndim > self.batch_dim:
            lz = lz.sum(0)
        red = self.reducer.all_reduce(local + [lz] + list(extra))
        for k, v in zip(names, red):
            setattr(self, k, v)
        self.logZ = red[len(names)]
        return red[len(names) + 1:]

    def _data_moment(self, ts, name, a, b):
        """sum_t a[t] b[t]' of two DATA tensors, remembered for as long as update_latents is called on the same,
        unmodified storage (the key holds address, version counter and layout; the operands are kept alive with the
        result so that the address cannot be handed to another tensor meanwhile)"""
        key = tuple((t.data_ptr(), t._version, tuple(t.shape), tuple(t.stride())) for t in (a, b))
        store = self.__dict__.setdefault("_data_moments", {})
        hit = store.get(name)
        if hit is None or hit[0] != key:
            hit = store[name] = (key, (a, b), ts(a, b))
        return hit[2]

    def update_latents(self, y, u, r, p=None, lr=1.0):
        """E-step: smoothed posteriors px, logZ and the time-integrated statistics (ref :156-216)."""
        if self.px is None:
            self.px = MultivariateNormal_vector_format(
                mu=torch.zeros(tuple(y.shape[:-2]) + (self.hidden_dim, 1), device=y.device, dtype=y.dtype))
        Sigma_t_tp1, Sigma_x0_x0, SE_x0, logZ, logZ_b = self.forward_backward_loop(y, u, r, sums_only=True)
        # time-integrated statistics: K10 cross-moment reductions (no (T, series, h, h) temporaries)
        Tn = y.shape[0]
        mu, Sig = self.px.mu, self.px.Sigma
        mv, yv, uv, rv = mu.squeeze(-1), y.squeeze(-1), u.squeeze(-1), r.squeeze(-1)
        ts = _TimeSums()
        SE_x0_x0 = Sigma_x0_x0 + SE_x0 @ _T(SE_x0)
        sum_xx, sum_xpx = getattr(self, "_time_sums", (None, None))
        sum_mu, sum_xy = getattr(self, "_obs_sums", (None, None))
        if sum_mu is not None:
            ts._sums[id(mv)] = (mv, sum_mu)  # sum_t mu_t came out of the backward sweep: no pass over mu for it
        SE_x_x = sum_xx if sum_xx is not None else ts(mv, mv, M=Sig)
        SE_xp_xp = SE_x_x - (mu[-1] @ _T(mu[-1]) + Sig[-1]) + SE_x0_x0
        SE_x_u = ts(mv, uv)
        SE_xp_u = ts(mv, uv, b_from=1, steps=Tn - 1) + SE_x0 @ _T(u[0])
        SE_xp_x = (sum_xpx if sum_xpx is not None else ts(mv, mv, M=Sigma_t_tp1, b_from=1, steps=Tn - 1)) \
            + SE_x0 @ _T(mu[0]) + Sigma_t_tp1[-1]
        SE_x_r = ts(mv, rv)
        SE_x_y = sum_xy if sum_xy is not None else ts(mv, yv)
        # moments of the data alone do not change between VB iterations on the same (unmodified) tensors
        SE_u_u = self._data_moment(ts, "uu", uv, uv)
        SE_r_r = self._data_moment(ts, "rr", rv, rv)
        SE_y_y = self._data_moment(ts, "yy", yv, yv)
        SE_y_r = self._data_moment(ts, "yr", yv, rv)

        sample_shape = tuple(y.shape[1:y.ndim - self.event_dim - self.batch_dim - 1])
        lead = sample_shape + self.batch_shape
        SE_y_r = SE_y_r.expand(lead + self.obs_shape + (self.regression_dim,))
        SE_u_u = SE_u_u.expand(lead + self.offset + (self.control_dim, self.control_dim))
        SE_r_r = SE_r_r.expand(lead + self.obs_shape[:-1] + (self.regression_dim, self.regression_dim))

        ones = torch.ones(lead + self.offset, device=y.device, dtype=y.dtype)
        self.T = y.shape[0] * ones
        self.N = ones
        self.SE_x_x = SE_x_x
        self.SE_x0_x0 = SE_x0_x0
        self.SE_x0 = SE_x0
        self.SE_y_xr = torch.cat((_T(SE_x_y), SE_y_r), dim=-1)
        self.SE_y_y = SE_y_y
        self.SE_xpu_xpu = torch.cat((torch.cat((SE_xp_xp, SE_xp_u), dim=-1),
                                     torch.cat((_T(SE_xp_u), SE_u_u), dim=-1)), dim=-2)
        self.SE_x_xpu = torch.cat((_T(SE_xp_x), SE_x_u), dim=-1)
        xx = SE_x_x.expand(tuple(SE_x_r.shape[:-2]) + tuple(SE_x_x.shape[-2:]))
        self.SE_xr_xr = torch.cat((torch.cat((xx, SE_x_r), dim=-1), torch.cat((_T(SE_x_r), SE_r_r), dim=-1)), dim=-2)
        for i in range(len(self.offset)):
            logZ = logZ.squeeze(-1)
        self.logZ = logZ.sum(0)

    def KLqprior(self):
        KL = self.x0.KLqprior() + self.A.KLqprior()
        for i in range(len(self.offset)):
            KL = KL.squeeze(-1)
        return KL + self.obs_model.KLqprior()

    def ELBO(self):
        logZ = self.logZ
        while logZ.ndim > self.batch_dim:
            logZ = logZ.sum(0)
        return logZ - self.KLqprior()

    # ------------------------------------------------------------------ model pieces
    def set_latent_parms(self):
        """expectations of the transition that the filter consumes (ref :230-242)"""
        h = self.hidden_dim
        self.invQ = self.A.EinvSigma()
        ATQA = self.A.EXTinvUX()
        self.ATQA_x_x = ATQA[..., :h, :h].contiguous()  # (dense once here: the smoother launch of every E-step wants it dense)
        self._vbmp_invATQA = None  # invATQA_x_x / logdetATQA_x_x (reference attributes; the device smoother never reads them): on demand
        self.ATQA_x_u = ATQA[..., :h, h:]
        self.ATQA_u_u = ATQA[..., h:, h:]
        QA = self.A.EinvUX()
        self.QA_xp_x = QA[..., :, :h].contiguous()
        self.QA_xp_u = QA[..., :, h:]
        # E log|invQ|: a function of the transition's noise parameters like the blocks above (six launches per E-step otherwise)
        self.A_Elogdet = self.A.ElogdetinvSigma()

    def _inv_ATQA(self):
        key = derived_key(self.ATQA_x_x)  # (identity, version, graph-replay epoch)
        c = self.__dict__.get("_vbmp_invATQA")
        if c is None or c[0] != key:
            c = self._vbmp_invATQA = (key,) + tuple(ops.spd_inv_logdet(self.ATQA_x_x)) + (self.ATQA_x_x,)
        return c

    @property
    def invATQA_x_x(self):
        return self._inv_ATQA()[1]

    @property
    def logdetATQA_x_x(self):
        return self._inv_ATQA()[2]

    def log_likelihood_function(self, Y, R):
        """natural parameters of the likelihood of x_t (ref :244-266)"""
        h = self.hidden_dim
        self.invR = self.obs_model.EinvSigma()
        BTRB = self.obs_model.EXTinvUX()
        self.BTRB_xp_xp = BTRB[..., :h, :h]
        self.BTRB_xp_r = BTRB[..., :h, h:]
        self.BTRB_r_r = BTRB[..., h:, h:]
        BTR = self.obs_model.EXTinvU()
        self.BTR_xp_y = BTR[..., :h, :]
        self.BTR_r_y = BTR[..., h:, :]

        invSigma_t_t = self.BTRB_xp_xp
        Rc = self._compact(R, 2)  # the regressor is usually the constant bias column: keep it unexpanded
        # a regressor that is the same for every (t, series) -- the usual bias column -- contributes constants: they ride along
        # in the two streaming kernels below instead of costing three more passes over (T, series, obs)
        const_r = Rc.numel() > 0 and all(s == 1 for s in Rc.shape[:-2]) and self.BTR_xp_y.ndim == 2
        fused = const_r and Y.is_cuda and Y.dim() > 2 and Y.shape[-1] == 1 and self.obs_dim <= ops.ROWS_QUAD_MAX_K and \
            self.invR.ndim == 2 and Y.numel() > 0
        if fused:
            # both halves of the observation message from ONE read of the observations (K12 with the scalar riding along)
            cst = 0.5 * self.obs_model.ElogdetinvSigma() - 0.5 * self.obs_dim * _LOG2PI
            lin = (_T(self.BTR_r_y) @ Rc).reshape(self.obs_dim)
            quad_r = -0.5 * (_T(Rc) @ self.BTRB_r_r @ Rc).reshape(())
            eta2, res2 = ops.rows_affine_quad(Y.reshape(-1, self.obs_dim), self.BTR_xp_y, -(self.BTRB_xp_r @ Rc).reshape(h),
                                              self.invR, lin, cst + quad_r)
            invSigmamu_t = eta2.reshape(tuple(Y.shape[:-2]) + (h, 1))
            Residual = res2.reshape(tuple(Y.shape[:-2]))
            invSigma_t_t = invSigma_t_t.expand(tuple(invSigmamu_t.shape[:-2]) + (h, h))
            return invSigma_t_t, invSigmamu_t, Residual
        if const_r:
            invSigmamu_t = shared_matvec(self.BTR_xp_y, Y, bias=-(self.BTRB_xp_r @ Rc).reshape(h))
        elif self.BTR_xp_y.ndim == 2:
            # one shared (h x obs) map applied to T*S observations: K12 streams the rows once when there are many (a
            # broadcast `@` is a batched (h x obs)@(obs x 1) product per (t, series), and even the tall-skinny row-major
            # GEMM Y2 @ M^T reaches a tenth of the memory bandwidth at 4e6 x 6 by 6 x 6)
            invSigmamu_t = shared_matvec(self.BTR_xp_y, Y) - self.BTRB_xp_r @ Rc
        else:
            invSigmamu_t = self.BTR_xp_y @ Y - self.BTRB_xp_r @ Rc
        # -1/2 y' invR y + y' (BTR_r_y' r) + const as ONE quadratic-form launch (K3a) instead of per-(t, series) bmm
        cst = 0.5 * self.obs_model.ElogdetinvSigma() - 0.5 * self.obs_dim * _LOG2PI
        lin = (_T(self.BTR_r_y) @ Rc).squeeze(-1)
        quad_r = -0.5 * (_T(Rc) @ self.BTRB_r_r @ Rc).squeeze(-1).squeeze(-1)
        if const_r and cst.ndim == 0:
            Residual = ops.quadform_loglike(Y.squeeze(-1), self.invR, lin.reshape(self.obs_dim), cst + quad_r.reshape(()))
        elif lin.ndim == cst.ndim + 1 and tuple(lin.shape[:-1]) == tuple(cst.shape):
            Residual = ops.quadform_loglike(Y.squeeze(-1), self.invR, lin, cst) + quad_r
        else:
            zero = torch.zeros(tuple(cst.shape) + (self.obs_dim,), device=Y.device, dtype=Y.dtype)
            Residual = ops.quadform_loglike(Y.squeeze(-1), self.invR, zero, cst) + quad_r + (Y.squeeze(-1) * lin).sum(-1)
        for i in range(len(self.obs_shape) - 1):
            invSigma_t_t = invSigma_t_t.sum(-3 - i, True)
            invSigmamu_t = invSigmamu_t.sum(-3 - i, True)
            Residual = Residual.sum(-1 - i, True)
        invSigma_t_t = invSigma_t_t.expand(tuple(invSigmamu_t.shape[:-2]) + (h, h))
        return invSigma_t_t, invSigmamu_t, Residual

    @staticmethod
    def _compact(U, keep_last):
        """drop broadcast (stride-0) leading axes of an expanded input so that products are not materialised"""
        idx = tuple(slice(0, 1) if (U.stride(i) == 0 and U.shape[i] > 1) else slice(None)
                    for i in range(U.ndim - keep_last))
        return U[idx]

    def forward_backward_loop(self, y, u, r, sums_only=False, fixed_point=None):
        """Filter + smoother for every series in one persistent kernel launch (K9).
        Returns Sigma_t_tp1, Sigma_x0_x0, mu_x0, logZ, None like the reference (:332-383) and fills self.px.
        sums_only=True (update_latents: it reads the cross terms only through their time sum and slot T-1, and logZ only through
        its time sum): the other slots of the returned Sigma_t_tp1 are unspecified and logZ has one time step, the sum.
        fixed_point: None = self.fixed_point ("auto" | "exact" | "off", see __init__)."""
        h = self.hidden_dim
        T_max = y.shape[0]
        sample_shape = tuple(y.shape[1:y.ndim - self.event_dim - self.batch_dim - 1])
        bo_shape = self.batch_shape + self.offset
        invSigma_like, invSigmamu_like, Residual_like = self.log_likelihood_function(y, r)
        Uc = self._compact(u, 2)
        cu1 = (self.QA_xp_u @ Uc).squeeze(-1)
        cu2 = (self.ATQA_x_u @ Uc).squeeze(-1)
        cu3 = (_T(Uc) @ self.ATQA_u_u @ Uc).squeeze(-1).squeeze(-1)
        x0 = self.x0
        # E[invSigma], E[invSigma mu] and -1/2 EXTinvUX + 1/2 ElogdetinvSigma - h/2 log 2 pi of the initial-state prior: exactly the
        # (P, b, c) of its expected log density, one launch (K13) instead of the four getters' sixteen
        x0_P, x0_eta, x0_res = x0.mixture_estep_params()
        if h <= ops.L.LDS_MAX_H or (h <= ops.L.LDS_MAX_H_BLOCK and ops.L.lds_block_fits(h, y.element_size())):
            out = ops.lds_smoother(T_max, sample_shape, bo_shape, h, self.invQ, self.ATQA_x_x, self.QA_xp_x,
                                   self.A_Elogdet, x0_P, x0_eta, x0_res,
                                   invSigma_like, invSigmamu_like.squeeze(-1), Residual_like, cu1, cu2, cu3,
                                   sums_only=sums_only, y=y.squeeze(-1) if (sums_only and len(self.offset) == 0) else None,
                                   fixed_point=self.fixed_point if fixed_point is None else fixed_point)
        else:
            out = self._smoother_composed(T_max, sample_shape + bo_shape, invSigma_like, invSigmamu_like.squeeze(-1),
                                          Residual_like, cu1, cu2, cu3, x0_res)
        self.px.invSigma = out["invSigma"]
        self.px.invSigmamu = out["invSigmamu"].unsqueeze(-1)
        self.px.Sigma = out["Sigma"]
        self.px.mu = out["mu"].unsqueeze(-1)
        self.px.logdetinvSigma = None
        # K9 accumulates the two matrix-valued time sums of update_latents in its backward sweep (None when composed)
        self._time_sums = (out.get("sum_xx"), out.get("sum_xpx"))
        self._obs_sums = (out.get("sum_mu"), out.get("sum_xy"))  # first-moment sums, where the device form accumulates them
        return out["Sigma_t_tp1"], out["Sigma_x0_x0"], out["mu_x0"].unsqueeze(-1), out["logZ"], None

    def _smoother_composed(self, T, lead, P_like, eta_like, res_like, cu1, cu2, cu3, x0_res):
        """Same recursion as K9 for hidden dimensions beyond its kernels (h > 64, or fp64 with h > 61 whose matrices
        do not fit LDS): a host loop over time whose every step is batched over the series -- the inverses / logdets
        are K1 launches, the products rocBLAS GEMMs.  Launch-bound (about 40 launches per time step)."""
        h = self.hidden_dim
        kw = {"device": eta_like.device, "dtype": eta_like.dtype}
        invQ, ATQA, QA = self.invQ, self.ATQA_x_x, self.QA_xp_x
        QAT = _T(QA)
        AEl = self.A_Elogdet
        x0P, x0e = self.x0.EinvSigma(), self.x0.EinvSigmamu()

        def at(X, t, inner):  # time slice of an operand that may not depend on time
            X = X if X.ndim >= len(lead) + 1 + inner else X.reshape((1,) * (len(lead) + 1 + inner - X.ndim) + tuple(X.shape))
            return X[t if X.shape[0] > 1 else 0]

        def mv(M, v):
            return (M @ v.unsqueeze(-1)).squeeze(-1)
        inv = ops.spd_inv_logdet
        oP = torch.empty((T,) + lead + (h, h), **kw)
        oe = torch.empty((T,) + lead + (h,), **kw)
        oS, om, oC = torch.empty_like(oP), torch.empty_like(oe), torch.empty_like(oP)
        logZ = torch.empty((T,) + lead, **kw)
        P = x0P.expand(lead + (h, h))
        eta = x0e.expand(lead + (h,))
        res = x0_res.expand(lead)
        Pi = mu = None
        for t in range(T):
            Pl, el, rl = at(P_like, t, 2), at(eta_like, t, 1), at(res_like, t, 0)
            c1, c2, c3 = at(cu1, t, 1), at(cu2, t, 1), at(cu3, t, 0)
            S1, ld1 = inv(P + ATQA)
            em = eta - c2
            W = QA @ S1
            P = Pl + invQ - W @ QAT
            eta = (el + c1) + mv(W, em)
            res = res + rl - 0.5 * c3 + 0.5 * AEl + 0.5 * (em * mv(S1, em)).sum(-1) - 0.5 * ld1
            Pi, ld2 = inv(P)
            mu = mv(Pi, eta)
            post = -0.5 * (mu * eta).sum(-1) + 0.5 * ld2 - 0.5 * h * _LOG2PI
            logZ[t] = res - post
            res = post
            oP[t], oe[t] = P, eta
            oC[t - 1] = S1  # t = 0 parks the x0 cross term in the last slot, like the reference
        oS[T - 1], om[T - 1] = Pi, mu
        G = torch.zeros(lead + (h, h), **kw)
        g = torch.zeros(lead + (h,), **kw)
        for t in range(T - 2, -2, -1):
            tl, tc = t + 1, (T - 1 if t < 0 else t)
            Pl, el = at(P_like, tl, 2), at(eta_like, tl, 1)
            c1, c2 = at(cu1, tl, 1), at(cu2, tl, 1)
            C0 = oC[tc]
            Mx = G + Pl + invQ - (QA @ C0) * QAT  # elementwise product as in the reference (:372)
            oC[tc] = C0 @ QAT @ ops.spd_inverse(Mx)
            S2 = ops.spd_inverse(invQ + Pl + G)
            V = QAT @ S2
            g = mv(V, (c1 + el) + g) - c2
            G = ATQA - V @ QA
            if t >= 0:
                oP[t] = oP[t] + G
                oe[t] = oe[t] + g
                oS[t] = ops.spd_inverse(oP[t])
                om[t] = mv(oS[t], oe[t])
        S00 = ops.spd_inverse(G + x0P)
        m0 = mv(S00, g + x0e)
        return {"invSigma": oP, "invSigmamu": oe, "Sigma": oS, "mu": om, "Sigma_t_tp1": oC, "logZ": logZ,
                "Sigma_x0_x0": S00, "mu_x0": m0}

