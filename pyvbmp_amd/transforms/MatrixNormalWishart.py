"""Matrix-normal-Wishart node: conjugate Bayesian linear map  y = A x + noise  with A (n x p) matrix
normal and the noise precision Wishart.  Same surface as the reference node
(transforms/MatrixNormalWishart.py:8-471): constructor, ss_update / update / raw_update, the
Elog_like* family, forward / backward messages, predict / postdict, KLqprior and the expectations.

Where the arithmetic runs:
  * sufficient statistics of update / raw_update: ONE K4 weighted-moment launch on the stacked vector
    z = [x; y] gives SExx, SEyx, SEyy, SEx, SEy and N together (no broadcast (T,S,...,p,p) temporaries);
  * every inverse / logdet (ss_update, forward, backward, Res): K1; the Wishart part: K2a;
  * Elog_like / Elog_like_given_pX_pY: one K3a quadratic-form launch on z with the block precision
    [[E[A'RA], -E[A'R]], [-E[RA], E[R]]];
  * the remaining products are plain batched GEMMs (rocBLAS through torch.matmul).
"""
import math

import torch

from .. import ops
from .._common import as_param, blend, derived_key, resolve, shared_matvec
from ..dists.MultivariateNormal_vector_format import MultivariateNormal_vector_format
from ..dists.Wishart import Wishart
from ..utils.matrix_utils import matrix_utils

_LOG2PI = math.log(2.0 * math.pi)


def _T(a):
    return a.transpose(-2, -1)


def _sq(a):
    return a.squeeze(-1).squeeze(-1)


def _cov_of(d):
    """covariance of an input 'distribution' (None for observed data wrapped in Delta)"""
    if hasattr(d, "ESigma"):
        return d.ESigma()
    return None


class MatrixNormalWishart():
    def __init__(self, event_shape, batch_shape=(), prior_parms=None, scale=1.0, mask=None, X_mask=None,
                 pad_X=False, fixed_precision=False, device=None, dtype=None):
        if prior_parms is None:
            prior_parms = {'mu': 0.0}
        self.device, self.dtype = resolve(device, dtype)
        event_shape, batch_shape = tuple(event_shape), tuple(batch_shape)
        self.n = event_shape[-2]
        self.p = event_shape[-1]
        self.pad_X = pad_X
        self.fixed_precision = fixed_precision
        mu_0 = as_param(prior_parms['mu'], self.device, self.dtype)
        if pad_X:
            self.p = self.p + 1
            event_shape = event_shape[:-1] + (self.p,)
            if mu_0.ndim != 0:
                mu_0 = torch.cat((mu_0, mu_0.new_zeros(tuple(mu_0.shape[:-1]) + (1,))), dim=-1)
        mu_0 = mu_0.expand(batch_shape + event_shape)
        self.event_dim = len(event_shape)
        self.event_shape = event_shape
        self.batch_dim = len(batch_shape)
        self.batch_shape = batch_shape

        self.mask = None if mask is None else mask.to(self.device)
        self.X_mask = None if X_mask is None else X_mask.to(self.device)
        self.mu_0 = mu_0
        self.mu = self._initial_mean(mu_0)

        eye = torch.eye(self.p, device=self.device, dtype=self.dtype)
        self.invV_0 = eye.expand(batch_shape + event_shape[:-2] + (self.p, self.p))
        self.invV = self.invV_0
        self.V = self.invV_0  # inverse of the identity
        zeros = torch.zeros(batch_shape + event_shape[:-2], device=self.device, dtype=self.dtype)
        self.logdetinvV = zeros
        self.logdetinvV_0 = zeros

        self.invU = self._make_noise(event_shape, batch_shape, scale)
        self.SEyy = 0.0
        self.SExx = 0.0
        self.SEyx = 0.0
        self.N = 0.0

        if self.X_mask is not None:
            if pad_X:
                ones = torch.ones(tuple(self.X_mask.shape[:-1]) + (1,), dtype=torch.bool, device=self.device)
                self.X_mask = torch.cat((self.X_mask, ones), dim=-1)
            self.mu_0 = self.mu_0 * self.X_mask
            self.mu = self.mu * self.X_mask
            self.V = self.V * self.X_mask * _T(self.X_mask)
            self.invV = self.invV * self.X_mask * _T(self.X_mask)
        if self.mask is not None:
            if pad_X:
                ones = torch.ones(tuple(self.mask.shape[:-1]) + (1,), dtype=torch.bool, device=self.device)
                self.mask = torch.cat((self.mask, ones), dim=-1)
            self.mu_0 = self.mu_0 * self.mask
            self.mu = self.mu * self.mask
        self.log2pi = torch.tensor(_LOG2PI, device=self.device, dtype=self.dtype)

    # hooks that the diagonal-noise sibling (MatrixNormalGamma) overrides
    def _initial_mean(self, mu_0):
        return torch.randn(mu_0.shape, device=self.device, dtype=self.dtype) / math.sqrt(self.p) + mu_0

    def _make_noise(self, event_shape, batch_shape, scale):
        return Wishart(event_shape=event_shape[:-2] + (self.n, self.n), batch_shape=batch_shape, scale=scale,
                       device=self.device, dtype=self.dtype)

    def _update_noise(self, W_arg, N, lr):
        self.invU.ss_update(W_arg, N, lr=lr, beta=None)

    def to_event(self, n):
        if n == 0:
            return self
        self.event_dim = self.event_dim + n
        self.batch_dim = self.batch_dim - n
        self.event_shape = self.batch_shape[-n:] + self.event_shape
        self.batch_shape = self.batch_shape[:-n]
        self.invU.to_event(n)
        return self

    # ------------------------------------------------------------------ conjugate update
    def ss_update(self, SExx, SEyx, SEyy, N, lr=1.0, beta=None):
        """Sufficient statistics -> posterior (ref transforms/MatrixNormalWishart.py:82-141)."""
        assert (SExx.ndim == self.batch_dim + self.event_dim)
        assert (SEyx.ndim == self.batch_dim + self.event_dim)
        assert (SEyy.ndim == self.batch_dim + self.event_dim)
        assert (N.ndim == self.batch_dim + self.event_dim - 2)
        if beta is not None:
            self.SExx = beta * self.SExx + SExx
            self.SEyx = beta * self.SEyx + SEyx
            self.SEyy = beta * self.SEyy + SEyy
            self.N = beta * self.N + N
            SExx, SEyx, SEyy, N = self.SExx, self.SEyx, self.SEyy, self.N

        if self.X_mask is not None:
            SExx = SExx * self.X_mask * _T(self.X_mask)
            SEyx = SEyx * self.X_mask
        invV = self.invV_0 + SExx
        V_new = ops.spd_inverse(invV)
        mu = (self.mu_0 @ self.invV_0 + SEyx) @ V_new
        if self.X_mask is not None:
            mu = mu * self.X_mask

        if self.mask is not None:  # linear constraint on the posterior mean; same mask for the whole batch
            mu = self._constrain_mean(mu, invV, V_new)

        if self.fixed_precision is False:
            W_arg = SEyy - mu @ invV @ _T(mu) + self.mu_0 @ self.invV_0 @ _T(self.mu_0)
            self._update_noise(W_arg, N, lr)
        invV = blend(invV, self.invV, lr)
        self.invV = 0.5 * (invV + _T(invV))
        self.mu = blend(mu, self.mu, lr)
        if self.mask is not None:
            self.mu = self.mu * self.mask
        self.V, self.logdetinvV = ops.spd_inv_logdet(self.invV)
        if self.X_mask is not None:
            self.mu = self.mu * self.X_mask

    def _mask_entries(self):
        """integer (row, col) indices of the entries the mask forces to zero and of those it leaves free; cached per
        mask tensor.  Boolean-mask indexing and the error check of linalg.solve would each force a device->host
        synchronisation in every update (the CPU could not run ahead of the GPU, and the iteration could not be
        captured into a HIP graph)."""
        cache = getattr(self, "_mask_cache", None)
        if cache is None or cache[0] is not self.mask:
            zi, zj = torch.nonzero(~self.mask, as_tuple=True)
            fi, fj = torch.nonzero(self.mask, as_tuple=True)
            self._mask_cache = cache = (self.mask, (zi, zj), (fi, fj))
        return cache[1], cache[2]

    @staticmethod
    def _kron_block(Rows, Cols, ri, ci):
        """(Cols (x) Rows) restricted to the entries (ri, ci): element (a, b) = Rows[ri[a], ri[b]] * Cols[ci[a], ci[b]],
        gathered directly (the full (n, p, n, p) Kronecker product is never formed)"""
        return Rows[..., ri.unsqueeze(-1), ri.unsqueeze(-2)] * Cols[..., ci.unsqueeze(-1), ci.unsqueeze(-2)]

    @staticmethod
    def _spd_solve(K, rhs):
        """K symmetric positive definite: Cholesky solve.  The reference calls the LU solver (:129); for one
        ~10^3-sized system rocSOLVER's LU is a pivot search, a scale and an update launch PER COLUMN (13 ms of launch
        latency per DMBD iteration at hidden 52), its blocked Cholesky a few dozen launches, and the solutions agree
        to rounding."""
        L, info = torch.linalg.cholesky_ex(K, check_errors=False)
        sol = torch.cholesky_solve(rhs.unsqueeze(-1), L).squeeze(-1)
        # a system that is not numerically positive definite (the reference's LU would still return something) must not
        # pass silently: its solution becomes NaN, on the device (no host synchronisation), and surfaces in the ELBO
        return torch.where((info == 0).unsqueeze(-1), sol, torch.full_like(sol, float("nan")))

    def _constrain_mean(self, mu, invV, V_new):
        """The posterior mean under the constraint mu[~mask] = 0, i.e. the minimiser of tr[(M - mu)' E[R] (M - mu) invV]
        over the matrices M that vanish off the mask.  The reference (:121-131) solves the DUAL system for the Lagrange
        multipliers of the zeroed entries, (V (x) U) restricted to those entries, and projects: mu - U gamma V.  The
        PRIMAL system for the free entries, (invV (x) E[R]) restricted to them with right-hand side (E[R] mu invV),
        has the same solution; whichever set of entries is smaller is solved (the transition matrix of the flocking
        DMBD has 2112 zeroed and 644 free entries: a 35x cheaper factorisation)."""
        (zi, zj), (fi, fj) = self._mask_entries()
        R = self.invU.EinvSigma()
        if fi.numel() <= zi.numel():
            sol = self._spd_solve(self._kron_block(R, invV, fi, fj), (R @ mu @ invV)[..., fi, fj])
            out = torch.zeros_like(mu)
            out[..., fi, fj] = sol
            return out
        U = ops.spd_inverse(R)
        gamma = torch.zeros_like(mu)
        gamma[..., zi, zj] = self._spd_solve(self._kron_block(U, V_new, zi, zj), mu[..., zi, zj])
        return (mu - U @ gamma @ V_new) * self.mask

    def _moments(self, EX, EY, covX, covY, p):
        """SExx, SEyx, SEyy, N (+ bias augmentation) from means / covariances / responsibilities.
        Inputs that are shared by all experts (component axes of size 1, e.g. the latent message every role of a
        DMBD observation sees) are NOT expanded over the experts: the K4 kernel reads each sample once and loops
        over the experts, and the covariance terms one streaming pass with a weight column per expert (K5b)."""
        nd = self.event_dim + self.batch_dim
        mat_batch = self.batch_shape + self.event_shape[:-2]
        nmb = len(mat_batch)
        lead = tuple(torch.broadcast_shapes(EX.shape[:-2], EY.shape[:-2]))
        nsd = len(lead) - nmb
        sample_shape = lead[:nsd]
        full = sample_shape + mat_batch
        px, n = EX.shape[-2], EY.shape[-2]
        z = torch.cat((EX.expand(lead + (px, 1)), EY.expand(lead + (n, 1))), dim=-2).squeeze(-1)
        pw = None if p is None else p.reshape(tuple(p.shape) + (1,) * (self.event_dim - 2))
        N, Sz, Szz = ops.weighted_moments(z, pw, nsd, mat_batch)
        SExx, SEyx, SEyy = Szz[..., :px, :px], Szz[..., px:, :px], Szz[..., px:, px:]
        SEx, SEy = Sz[..., :px].unsqueeze(-1), Sz[..., px:].unsqueeze(-1)

        def wsum(C):
            if C is None:
                return 0.0
            d = C.shape[-1]
            if C.ndim <= nmb + 2:  # no sample axes: shared by every sample
                return C * N.reshape(tuple(N.shape) + (1, 1))
            if C.ndim < len(full) + 2:
                C = C.reshape((1,) * (len(full) + 2 - C.ndim) + tuple(C.shape))
            shared = all(C.shape[nsd + i] == 1 for i in range(nmb))
            S = int(math.prod(sample_shape))
            NBm = int(math.prod(mat_batch))
            if (shared or nmb == 0) and NBm == 1:
                # a single expert: one streaming pass over the covariances (K5b)
                Cs = C.expand(sample_shape + (1,) * nmb + (d, d)).reshape(S, d, d)
                ws = None if pw is None else pw.expand(full).reshape(S)
                return ops.weighted_matsum(Cs, ws).reshape(mat_batch + (d, d))
            if shared and pw is not None and nmb > 0:
                Cv = C.expand(sample_shape + (1,) * nmb + (d, d))
                Wv = pw.expand(full)
                # sample axes along which the covariances do not vary (broadcast views, e.g. the observables of a DMBD that all
                # see one latent message): the sum is linear in the weights, so THEY are summed over those axes first and every
                # distinct covariance is read once
                const = [i for i in range(nsd) if sample_shape[i] > 1 and Cv.stride(i) == 0]
                if const:
                    Wv = Wv.sum(const, keepdim=True)
                    Cv = Cv[tuple(slice(0, 1) if i in const else slice(None) for i in range(nsd))]
                S2 = int(math.prod(Cv.shape[:nsd]))
                return ops.weighted_matsum_cols(Cv.reshape(S2, d * d), Wv.reshape(S2, -1)).reshape(mat_batch + (d, d))
            C = C.expand(full + (d, d))
            if pw is None:
                return C.sum(tuple(range(nsd)))
            return (C * pw.expand(full).reshape(full + (1, 1))).sum(tuple(range(nsd)))

        SExx = SExx + wsum(covX)
        SEyy = SEyy + wsum(covY)
        if self.pad_X:
            SExx = torch.cat((SExx, SEx), dim=-1)
            last = torch.cat((SEx, N.reshape(tuple(N.shape) + (1, 1))), dim=-2)
            SExx = torch.cat((SExx, _T(last)), dim=-2)
            SEyx = torch.cat((SEyx, SEy), dim=-1)
        return SExx, SEyx, SEyy, N

    def update(self, pX, pY, p=None, lr=1.0, beta=None):
        """Input / output distributions (+ responsibilities) -> ss_update (ref :143-172)."""
        SExx, SEyx, SEyy, N = self._moments(pX.EX(), pY.EX(), _cov_of(pX), _cov_of(pY), p)
        self.ss_update(SExx, SEyx, SEyy, N, lr=lr, beta=beta)

    def raw_update(self, X, Y, p=None, lr=1.0, beta=None):
        """Data (+ responsibilities) -> ss_update (ref :174-204)."""
        SExx, SEyx, SEyy, N = self._moments(X, Y, None, None, p)
        self.ss_update(SExx, SEyx, SEyy, N, lr=lr, beta=beta)

    def _mn_kl(self):
        """the matrix-normal part of KLqprior from one launch (K15), or None where the kernel does not serve (state off the GPU,
        n p beyond its LDS image)"""
        if not self.mu.is_cuda or not ops.mn_kl_serves(self.n, self.p):
            return None
        xm = None
        if self.X_mask is not None:
            xm = self.__dict__.get("_vbmp_xmask_count")
            if xm is None:  # the mask never changes: its set entries are counted once (per batch element)
                xm = self._vbmp_xmask_count = self.X_mask.sum((-1, -2)).to(self.dtype)
        lead = tuple(self.mu.shape[:-2])
        R = self.EinvSigma()
        KL = ops.mn_kl(self.mu, self.mu_0, self.invV_0, self.V, R.expand(lead + tuple(R.shape[-2:])), self.logdetinvV,
                       self.logdetinvV_0, xm)
        for i in range(self.event_dim - 2):
            KL = KL.sum(-1)
        return KL

    def KLqprior(self):
        KL = self._mn_kl()
        if KL is not None:
            return KL + self.invU.KLqprior()
        return self._KLqprior_composed()

    def _KLqprior_composed(self):
        KL = self.n / 2.0 * self.logdetinvV - self.n / 2.0 * self.logdetinvV_0 - self.n * self.p / 2.0
        if self.X_mask is not None:
            KL = KL + self.n / 2.0 * self.logdetinvV_0 * (self.X_mask).sum((-1, -2))
        KL = KL + 0.5 * self.n * (self.invV_0 * self.V).sum((-1, -2))
        d = self.mu - self.mu_0
        KL = KL + 0.5 * (self.invV_0 * (_T(d) @ self.invU.EinvSigma() @ d)).sum((-1, -2))
        for i in range(self.event_dim - 2):
            KL = KL.sum(-1)
        return KL + self.invU.KLqprior()

    # ------------------------------------------------------------------ likelihoods
    def _joint_quadratic(self):
        """(P, b, c): Elog_like(x, y) = -1/2 z^T P z + z^T b + c with z = [x; y] (bias column folded in)."""
        R, G, H = self.EinvSigma(), self.EinvUX(), self.EXTinvUX()
        c = 0.5 * self.ElogdetinvSigma() - 0.5 * self.n * _LOG2PI
        if self.pad_X:
            H11, G1 = H[..., :-1, :-1], G[..., :, :-1]
            b = torch.cat((-H[..., :-1, -1], G[..., :, -1]), dim=-1)
            c = c - 0.5 * H[..., -1, -1]
        else:
            H11, G1 = H, G
            b = torch.zeros(tuple(c.shape) + (self.p + self.n,), device=self.device, dtype=self.dtype)
        P = torch.cat((torch.cat((H11, -_T(G1)), dim=-1), torch.cat((-G1, R), dim=-1)), dim=-2)
        return P, b, c

    def _stack(self, X, Y):
        lead = torch.broadcast_shapes(X.shape[:-2], Y.shape[:-2])
        return torch.cat((X.expand(lead + tuple(X.shape[-2:])), Y.expand(lead + tuple(Y.shape[-2:]))), dim=-2).squeeze(-1)

    def Elog_like(self, X, Y):
        P, b, c = self._joint_quadratic()
        ELL = ops.quadform_loglike(self._stack(X, Y), P, b, c)
        for i in range(self.event_dim - 2):
            ELL = ELL.sum(-1)
        return ELL

    def Elog_like_given_pX_pY(self, pX, pY):
        P, b, c = self._joint_quadratic()
        ELL = ops.quadform_loglike(self._stack(pX.mean(), pY.mean()), P, b, c)
        px = self.p - 1 if self.pad_X else self.p
        cx, cy = _cov_of(pX), _cov_of(pY)
        nmb = P.ndim - 2

        def trace_term(C, Pb):
            """sum_ij C_ij Pb_ij per (sample, expert); one GEMM when C is shared by the experts"""
            d = C.shape[-1]
            if nmb > 0 and C.ndim >= nmb + 2 and all(C.shape[C.ndim - 2 - nmb + i] == 1 for i in range(nmb)):
                ns = C.ndim - 2 - nmb
                lead_s = tuple(C.shape[:ns])
                # sample axes along which C is a broadcast view (stride 0): the trace is the same along them -- computed once on the
                # compact covariances and expanded, instead of materialising the view for the GEMM
                idx = tuple(slice(0, 1) if (C.stride(i) == 0 and C.shape[i] > 1) else slice(None) for i in range(ns))
                Cc = C[idx]
                Cf, Pf = Cc.reshape(-1, d * d), Pb.reshape(-1, d * d)
                keep = self._xmask_union(d) if Pb is Px else None
                if keep is not None:  # entries that X_mask zeroes in EVERY expert's precision drop out of the inner dimension
                    Cf, Pf = Cf.index_select(1, keep), Pf.index_select(1, keep)
                out = Cf @ Pf.transpose(0, 1)
                return out.reshape(tuple(Cc.shape[:ns]) + tuple(Pb.shape[:-2])).expand(lead_s + tuple(Pb.shape[:-2]))
            return (C * Pb).sum((-1, -2))
        Px = P[..., :px, :px]
        if cx is not None:
            ELL = ELL - 0.5 * trace_term(cx, Px)
        if cy is not None:
            ELL = ELL - 0.5 * trace_term(cy, P[..., px:, px:])
        for i in range(self.event_dim - 2):
            ELL = ELL.sum(-1)
        return ELL

    def _xmask_union(self, px):
        """flat indices of the (px x px) entries of E[X' invU X] that X_mask leaves free in at least one batch element, or None when
        there is no mask / the union covers more than half of the matrix.  E[X' invU X] = n V + mu' R mu is EXACTLY zero outside
        X_mask (x) X_mask (V and mu are masked), so a trace against it only needs those entries: for the 25 roles of the flocking
        DMBD, each reading one object's 8 latent dimensions, 400 of 2 704 -- the trace-term GEMM of the role assignments was 0.15 ms,
        a fifth of the E-step.  The mask never changes: found once (one host synchronisation)."""
        if self.X_mask is None or self.X_mask.shape[-2] != 1:
            return None
        c = self.__dict__.get("_vbmp_xmask_union")
        if c is None or c[0] != px:
            xm = self.X_mask[..., 0, :px].reshape(-1, px).bool()               # (batch elements, px)
            free = (xm.unsqueeze(-1) & xm.unsqueeze(-2)).any(0).reshape(-1)    # union over the batch of xm (x) xm
            keep = torch.nonzero(free).reshape(-1)
            c = self._vbmp_xmask_union = (px, keep if 2 * keep.numel() <= px * px else None)
        return c[1]

    def _residual_y(self, Y):
        """-1/2 y' E[R] y - n/2 log 2pi + 1/2 E log|R|  (one K3a launch)"""
        R = self.EinvSigma()
        c = 0.5 * self.ElogdetinvSigma() - 0.5 * self.n * _LOG2PI
        zero = torch.zeros(tuple(c.shape) + (self.n,), device=self.device, dtype=self.dtype)
        return ops.quadform_loglike(Y.squeeze(-1), R, zero, c)

    def Elog_like_X(self, Y):
        H, Gt = self.EXTinvUX(), self.EXTinvU()
        Residual = self._residual_y(Y)
        if self.pad_X:
            return H[..., :-1, :-1], shared_matvec(Gt[..., :-1, :], Y) - H[..., :-1, -1:], Residual - 0.5 * H[..., -1, -1]
        return H, shared_matvec(Gt, Y), Residual

    def _joint_blocks(self, pY, bias_sign):
        R, G, H = self.EinvSigma(), self.EinvUX(), self.EXTinvUX()
        Jyy = pY.EinvSigma() + R
        if self.pad_X:
            return (Jyy, -G[..., :, :-1], H[..., :-1, :-1], pY.EinvSigmamu() + bias_sign * G[..., :, -1:],
                    -H[..., :-1, -1:], H[..., -1, -1])
        zero = torch.zeros(tuple(H.shape[:-1]) + (1,), device=self.device, dtype=self.dtype)
        return Jyy, -G, H, pY.EinvSigmamu(), zero, torch.zeros((), device=self.device, dtype=self.dtype)

    def _marginalise(self, pY, bias_sign, Res0):
        Jyy, Jyx, Jxx, jy, jx, J11 = self._joint_blocks(pY, bias_sign)
        Pyy, nBiD, nCiA, Pxx = matrix_utils.block_precision_marginalizer(Jyy, Jyx, _T(Jyx), Jxx)
        eta_y = jy + nBiD @ jx
        eta_x = jx + nCiA @ jy
        Syy, ld_yy = ops.spd_inv_logdet(Pyy)
        Res = Res0 + pY.Res() + 0.5 * _sq(_T(eta_y) @ Syy @ eta_y) - 0.5 * ld_yy + 0.5 * pY.dim * _LOG2PI \
            + 0.5 * self.ElogdetinvSigma() - 0.5 * J11
        return Pxx, eta_x, Res

    def Elog_like_X_given_pY(self, pY):
        Pxx, eta_x, Res = self._marginalise(pY, -1.0, 0.0)
        Sxx, ld = ops.spd_inv_logdet(Pxx)
        px = MultivariateNormal_vector_format(invSigma=Pxx, invSigmamu=eta_x, mu=Sxx @ eta_x, Sigma=Sxx,
                                              logdetinvSigma=ld)
        return px, Res - px.Res()

    # ------------------------------------------------------------------ messages
    def Eforward(self, pX):
        G = self.EinvUX()
        eta = G[..., :, :-1] @ pX.mean()
        if self.pad_X:
            eta = eta + G[..., :, -1:]
        return MultivariateNormal_vector_format(invSigma=self.EinvSigma(), invSigmamu=eta)

    def forward(self, pX):
        """Message x -> y for a Gaussian input (ref :303-328): returns (MVN_vf(mu, Sigma), Res)."""
        Px, etax = pX.EinvSigma(), pX.EinvSigmamu()
        nV, M = self.n * self.V, self.mean()
        if self.pad_X:
            nV11, eta, M1 = nV[..., :-1, :-1], etax - nV[..., :-1, -1:], M[..., :-1]
        else:
            nV11, eta, M1 = nV, etax, M
        bshape = self.batch_shape + self.event_shape[:-2]
        per_message = any(Px.shape[i] != 1 and Px.stride(i) != 0 for i in range(Px.ndim - 2 - len(bshape)))
        if per_message and ops.mnw_message_fusable(self.n, M1.shape[-1]):
            # one precision per message (BASELINE config 3): everything in ONE fused kernel (K7)
            # Res = -q1/2 + q2/2 - (ld2 - ld1)/2 (- nV[-1,-1]/2 for the bias column) is formed by the kernel's epilogue
            mu_y, Sigma_yy, sc, Res = ops.mnw_message(Px, etax.squeeze(-1), eta.squeeze(-1), None, nV11, None, M1,
                                                      self.invEinvSigma(), None, 1.0, bshape,
                                                      res_w=(-0.5, 0.5, 0.5, -0.5, 0.0, 0.0, 0.0, 0.0),
                                                      res_c=-0.5 * nV[..., -1, -1] if self.pad_X else None)
            mu_y = mu_y.unsqueeze(-1)
            if self.pad_X:
                mu_y = mu_y + M[..., -1:]
            return MultivariateNormal_vector_format(mu=mu_y, Sigma=Sigma_yy), Res
        # shared precision: one factorisation per expert, per-message work is GEMV (K1 + GEMMs)
        Sx, ld_x = ops.spd_inv_logdet(Px)
        S, ld_s = ops.spd_inv_logdet(nV11 + Px)
        mu_y = M1 @ (S @ eta)
        if self.pad_X:
            mu_y = mu_y + M[..., -1:]
        Sigma_yy = M1 @ S @ _T(M1) + self.invEinvSigma()
        # -1/2 mu_x' P mu_x = -1/2 eta_x' P^-1 eta_x ;  logdet(nV P^-1 + I) = logdet(nV + P) - logdet P
        Res = -0.5 * _sq(_T(etax) @ Sx @ etax) + 0.5 * _sq(_T(eta) @ S @ eta) - 0.5 * (ld_s - ld_x)
        if self.pad_X:
            Res = Res - 0.5 * nV[..., -1, -1]
        return MultivariateNormal_vector_format(mu=mu_y, Sigma=Sigma_yy), Res

    def backward(self, pY, Res=0.0):
        """Message y -> x (ref :352-375): returns (MVN_vf(invSigma, invSigmamu), Res)."""
        Py = pY.EinvSigma()
        bshape = self.batch_shape + self.event_shape[:-2]
        per_message = any(Py.shape[i] != 1 and Py.stride(i) != 0 for i in range(Py.ndim - 2 - len(bshape)))
        px_dim = self.p - 1 if self.pad_X else self.p
        if per_message and ops.mnw_message_fusable(px_dim, self.n):
            # one precision per message: the block marginalisation and all four residual terms in ONE kernel (K8)
            Rm, G, H = self.EinvSigma(), self.EinvUX(), self.EXTinvUX()
            etay = pY.EinvSigmamu()
            if self.pad_X:
                G1, H11 = G[..., :, :-1], H[..., :-1, :-1]
                jy, jx, J11 = etay + G[..., :, -1:], -H[..., :-1, -1:], H[..., -1, -1]
            else:
                G1, H11, jy = G, H, etay
                jx = torch.zeros(tuple(H.shape[:-1]) + (1,), device=self.device, dtype=self.dtype)
                J11 = 0.0
            H11inv, ld_H = ops.spd_inv_logdet(H11)                  # per expert
            eta_y = jy + (G1 @ H11inv) @ jx if self.pad_X else jy   # (no bias column: j_x = 0, no pass over the messages)
            # the marginal precision of y, Rm - G H^-1 G' + P_y, is never eliminated: its log-determinant and quadratic
            # form follow from the two eliminations the kernel runs anyway (Schur mode: scal[4] = q3, scal[5] = ld3 + ld_H)
            # the kernel's epilogue forms the residual 1/2 (-q1 + ld1 + q3 - (ld3 - ld_H) + q4 - ld4) + per-expert constants and
            # stores invSigmamu_x = ovec + jx directly
            const = 0.5 * ld_H + 0.5 * self.ElogdetinvSigma() - 0.5 * J11 + 0.5 * px_dim * _LOG2PI
            eta_x, Pxx, sc, R = ops.mnw_message(Py, etay.squeeze(-1), jy.squeeze(-1), eta_y.squeeze(-1), Rm, None, _T(G1), H11,
                                                jx.squeeze(-1), -1.0, bshape, res_w=(-0.5, 0.5, 0.0, 0.0, 0.5, -0.5, 0.5, -0.5),
                                                res_c=const, add_cvec=True)
            if not (isinstance(Res, float) and Res == 0.0):
                R = Res + R
            return MultivariateNormal_vector_format(invSigma=Pxx, invSigmamu=eta_x.unsqueeze(-1), logdetinvSigma=sc[..., 7]), R
        Pxx, eta_x, R = self._marginalise(pY, +1.0, Res)
        pX = MultivariateNormal_vector_format(invSigma=Pxx, invSigmamu=eta_x)
        return pX, R - pX.Res()

    def Ebackward(self, pY):
        raise NotImplementedError

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
        pY = MultivariateNormal_vector_format(invSigma=self.EinvSigma(), invSigmamu=eta)
        return pY, Res - pY.Res()

    def postdict(self, Y):
        P, eta, Residual = self.Elog_like_X(Y)
        pX = MultivariateNormal_vector_format(invSigma=P, invSigmamu=eta)
        return pX, Residual - pX.Res()

    def predict_given_pX(self, pX):
        return self.forward(pX)

    # ------------------------------------------------------------------ expectations
    def mean(self):
        return self.mu

    def bias(self):
        return self.mu[..., -1:] if self.pad_X is True else torch.tensor(0.0, device=self.device, dtype=self.dtype)

    def weights(self):
        return self.mu[..., :-1] if self.pad_X is True else self.mu

    def var(self):
        return self.ESigma().diagonal(dim1=-1, dim2=-2).unsqueeze(-1) * self.V.diagonal(dim1=-1, dim2=-2).unsqueeze(-2)

    def _expectations(self):
        """(EinvSigma, EinvUX, EXTinvUX, ElogdetinvSigma) from ONE launch (K14), kept until a tensor they were computed from is
        rebound or written (composed from the getters: 14 launches wherever a caller asks for them); None where the noise model is
        not a plain Wishart (the diagonal sibling overrides the getters) or the state is not on the GPU.  Callers treat the
        returned tensors as read-only."""
        W = self.invU
        if type(W) is not Wishart or not self.mu.is_cuda:
            return None
        src = (self.mu, W.U, W.nu, self.V, W.logdet_invU)
        key = derived_key(*src)
        c = self.__dict__.get("_vbmp_expect")
        if c is None or c[0] != key:
            c = self._vbmp_expect = (key, ops.mnw_expectations(*src), src)  # src kept alive: ids stay unique
        return c[1]

    def EinvUX(self):
        e = self._expectations()
        return e[1] if e is not None else self.invU.EinvSigma() @ self.mu

    def EXTinvU(self):
        e = self._expectations()
        return _T(e[1]) if e is not None else _T(self.mu) @ self.invU.EinvSigma()

    def EXTAX(self, A):
        return self.V * (self.invU.ESigma() * A).sum((-1, -2)) + _T(self.mu) @ A @ self.mu

    def EXmMUTAXmMU(self, A):
        return self.V * (self.invU.ESigma() * A).sum((-1, -2))

    def EXAXT(self, A):
        return self.ESigma() * (self.V * A).sum((-1, -2)) + self.mu @ A @ _T(self.mu)

    def EXmMUAXmMUT(self, A):
        return self.ESigma() * (self.V * A).sum((-1, -2))

    def EXTinvUX(self):
        e = self._expectations()
        return e[2] if e is not None else self.n * self.V + _T(self.mu) @ self.invU.EinvSigma() @ self.mu

    def EXinvVXT(self):
        return self.p * self.invU.ESigma() + self.mu @ self.invV @ _T(self.mu)

    def EXmMUTinvUXmMU(self):
        return self.n * self.V

    def EXmMUinvVXmMUT(self):
        return self.p * self.invU.ESigma()

    def EXTX(self):
        return self.V * self.invU.ESigma().diagonal().sum() + _T(self.mu) @ self.mu

    def EXXT(self):
        return self.V.diagonal().sum() * self.invU.ESigma() + self.mu @ _T(self.mu)

    def ElogdetinvU(self):
        return self.invU.ElogdetinvSigma()

    def logdetEinvSigma(self):
        return self.invU.logdetEinvSigma()

    def ElogdetinvSigma(self):
        e = self._expectations()
        return e[3] if e is not None else self.invU.ElogdetinvSigma()

    def EinvSigma(self):
        e = self._expectations()
        return e[0] if e is not None else self.invU.EinvSigma()

    def invEinvSigma(self):
        return self.invU.invEinvSigma()

    def ESigma(self):
        return self.invU.ESigma()
