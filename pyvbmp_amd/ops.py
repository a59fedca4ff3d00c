"""Tensor-level entry points: shape/stride marshalling around the C-ABI kernels.

Everything here takes torch tensors that already live on the HIP device, flattens the leading
batch dims, turns broadcast (stride-0 expanded) operands into stride-0 kernel arguments instead of
materialising them, allocates the outputs and enqueues ONE kernel on the current stream.
"""
import ctypes
import math

import torch

from . import _lib as L


CHECK_SPD = False  # pyvbmp_amd.debug.check_spd(): read the kernels' non-SPD counter back after every K1 / K2 / K2a launch


def _prod(shape):
    return int(math.prod(shape))


def _spd_counter(dev, given):
    """the device counter handed to the kernel: the caller's, or a fresh one under debug.check_spd()"""
    if given is not None or not CHECK_SPD:
        return given, False
    return torch.zeros(1, dtype=torch.int32, device=dev), True


def _spd_verify(counter, own, what):
    if own:
        n = int(counter.item())  # synchronises: debug mode only
        if n:
            raise L.VbmpHipError(f"{what}: {n} matrices with a non-positive pivot (not symmetric positive definite); the "
                                 "elimination does not pivot and their log-determinants are NaN / -inf as in the reference")


def batch_operand(t, batch_shape, inner_shape):
    """Return (tensor, element stride between batch entries) for an operand that is broadcastable
    to batch_shape + inner_shape.  Fully shared operands (every batch stride 0, dense inner block)
    are passed as stride 0; everything else is made dense."""
    full = tuple(batch_shape) + tuple(inner_shape)
    ni = len(inner_shape)
    if t.ndim < len(full):
        t = t.reshape((1,) * (len(full) - t.ndim) + tuple(t.shape))
    if tuple(t.shape) != full:
        t = t.expand(full)
    nb = len(batch_shape)
    shared = all(t.stride(i) == 0 or t.shape[i] == 1 for i in range(nb))
    if shared and nb > 0 and _prod(batch_shape) > 0:
        # one block for the whole batch: densify just that block (e.g. a scalar prior mean expanded
        # over (B, D) has inner stride 0 too) and hand it to the kernel with batch stride 0
        return t[(0,) * nb].contiguous(), 0
    t = t.contiguous()
    return t, _prod(inner_shape)


def spd_inv_logdet(A, want_logdet=True, nonspd=None):
    """(A^-1, log det A) for a (..., D, D) stack of SPD matrices.  One K1 launch."""
    dev = L.require_device(A)
    lib = L.load()
    D = A.shape[-1]
    assert A.shape[-2] == D
    bshape = tuple(A.shape[:-2])
    B = _prod(bshape)
    if D > L.MAX_DIM:
        # beyond the one-wave-per-matrix kernels (VBMP_MAX_DIM = 64): the device library's factorisations (rocSOLVER through
        # torch.linalg, still on the GPU); Tensor.logdet keeps the reference's NaN / -inf semantics
        nonspd, own = _spd_counter(dev, nonspd)
        if nonspd is not None and B > 0:
            # the library route has no pivot counter: count the matrices whose Cholesky factorisation fails (no host sync)
            nonspd += (torch.linalg.cholesky_ex(A).info.reshape(-1) != 0).sum().to(torch.int32)
            _spd_verify(nonspd, own, "spd_inv_logdet")
        return torch.linalg.inv(A), (torch.logdet(A) if want_logdet else None)
    Ac = A.contiguous()
    Ainv = torch.empty_like(Ac)
    logdet = torch.empty(bshape, dtype=A.dtype, device=dev) if want_logdet else None
    if B > 0:
        nonspd, own = _spd_counter(dev, nonspd)
        fn = getattr(lib, "vbmp_spd_inv_logdet_" + L.suffix(A.dtype))
        L.call(fn, "vbmp_spd_inv_logdet", L.ptr(Ac), D * D, L.ptr(Ainv), L.ptr(logdet), B, D, L.ptr(nonspd), L.stream_ptr(dev))
        _spd_verify(nonspd, own, "spd_inv_logdet")
    return Ainv, logdet


def spd_inverse(A):
    return spd_inv_logdet(A, want_logdet=False)[0]


def wishart_ss_update(SExx, N, invU0, nu0, invU_old, nu_old, lr, nonspd=None):
    """K2a.  SExx: batch+(D,D); N: batch.  Returns (invU, nu, U, logdet) dense, shaped like SExx / N."""
    dev = L.require_device(SExx, N, invU0, nu0)
    lib = L.load()
    D = SExx.shape[-1]
    bshape = tuple(SExx.shape[:-2])
    B = _prod(bshape)
    dt = SExx.dtype
    SExx_c, sS = batch_operand(SExx, bshape, (D, D))
    N_c, sN = batch_operand(N.to(dt), bshape, ())
    i0, si0 = batch_operand(invU0.to(dt), bshape, (D, D))
    n0, sn0 = batch_operand(nu0.to(dt), bshape, ())
    blend = float(lr) != 1.0
    if blend:
        io, sio = batch_operand(invU_old.to(dt), bshape, (D, D))
        no, sno = batch_operand(nu_old.to(dt), bshape, ())
    else:
        io = no = None
        sio = sno = 0
    invU = torch.empty(bshape + (D, D), dtype=dt, device=dev)
    U = torch.empty_like(invU)
    nu = torch.empty(bshape, dtype=dt, device=dev)
    logdet = torch.empty(bshape, dtype=dt, device=dev)
    if B > 0:
        suf = L.suffix(dt)
        fn = getattr(lib, "vbmp_wishart_ss_update_" + suf)
        cT = L.DTYPES[suf][1]
        nonspd, own = _spd_counter(dev, nonspd)
        L.call(fn, "vbmp_wishart_ss_update", L.ptr(SExx_c), sS, L.ptr(N_c), sN, L.ptr(i0), si0, L.ptr(n0), sn0, L.ptr(io), sio, L.ptr(no), sno,
                   cT(float(lr)), L.ptr(invU), L.ptr(nu), L.ptr(U), L.ptr(logdet), B, D, L.ptr(nonspd),
                   L.stream_ptr(dev))
        _spd_verify(nonspd, own, "wishart_ss_update")
    return invU, nu, U, logdet


def niw_ss_update(SExx, SEx, N, lam0, mu0, invU0, nu0, lam_old, mu_old, invU_old, nu_old, lr, fixed_precision=False,
                  nonspd=None):
    """K2 (headline op).  SExx: batch+(D,D); SEx: batch+(D,); N, lam*: batch.
    Returns (lam, mu, invU, nu, U, logdet); the last four are None when fixed_precision."""
    dev = L.require_device(SExx, SEx, N, lam0, mu0, invU0, nu0)
    lib = L.load()
    D = SExx.shape[-1]
    bshape = tuple(SExx.shape[:-2])
    B = _prod(bshape)
    dt = SExx.dtype
    SExx_c, sS = batch_operand(SExx, bshape, (D, D))
    SEx_c, sx = batch_operand(SEx.to(dt), bshape, (D,))
    N_c, sN = batch_operand(N.to(dt), bshape, ())
    l0, sl0 = batch_operand(lam0.to(dt), bshape, ())
    m0, sm0 = batch_operand(mu0.to(dt), bshape, (D,))
    i0, si0 = batch_operand(invU0.to(dt), bshape, (D, D))
    n0, sn0 = batch_operand(nu0.to(dt), bshape, ())
    blend = float(lr) != 1.0
    lo = mo = io = no = None
    slo = smo = sio = sno = 0
    if blend:
        lo, slo = batch_operand(lam_old.to(dt), bshape, ())
        mo, smo = batch_operand(mu_old.to(dt), bshape, (D,))
        if not fixed_precision:
            io, sio = batch_operand(invU_old.to(dt), bshape, (D, D))
            no, sno = batch_operand(nu_old.to(dt), bshape, ())
    lam = torch.empty(bshape, dtype=dt, device=dev)
    mu = torch.empty(bshape + (D,), dtype=dt, device=dev)
    if fixed_precision:
        invU = U = nu = logdet = None
    else:
        invU = torch.empty(bshape + (D, D), dtype=dt, device=dev)
        U = torch.empty_like(invU)
        nu = torch.empty(bshape, dtype=dt, device=dev)
        logdet = torch.empty(bshape, dtype=dt, device=dev)
    if B > 0:
        suf = L.suffix(dt)
        fn = getattr(lib, "vbmp_niw_ss_update_" + suf)
        cT = L.DTYPES[suf][1]
        nonspd, own = _spd_counter(dev, nonspd)
        L.call(fn, "vbmp_niw_ss_update", L.ptr(SExx_c), sS, L.ptr(SEx_c), sx, L.ptr(N_c), sN, L.ptr(l0), sl0, L.ptr(m0), sm0, L.ptr(i0), si0,
                   L.ptr(n0), sn0, L.ptr(lo), slo, L.ptr(mo), smo, L.ptr(io), sio, L.ptr(no), sno, cT(float(lr)),
                   L.ptr(lam), L.ptr(mu), L.ptr(invU), L.ptr(nu), L.ptr(U), L.ptr(logdet), B, D,
                   1 if fixed_precision else 0, L.ptr(nonspd), L.stream_ptr(dev))
        _spd_verify(nonspd, own, "niw_ss_update")
    return lam, mu, invU, nu, U, logdet


def _split_components(x_comp_shape, mat_batch):
    """X carries one axis per component axis, each of size 1 (broadcast) or full.  Return k such that
    axes [:k] are broadcast ("outer": every outer component sees the same sample) and axes [k:] are
    dense ("inner"); None when the pattern is mixed and X has to be materialised."""
    k = 0
    n = len(mat_batch)
    while k < n and x_comp_shape[k] == 1:
        k += 1
    if tuple(x_comp_shape[k:]) != tuple(mat_batch[k:]):
        return None
    return k


def _dense_samples(X, mat_batch, D):
    """X: sample + comp* + (D,) -> (X2 dense (S,Bi,D), sample_shape, Bo, Bi)."""
    nb = len(mat_batch)
    sample_shape = tuple(X.shape[:X.ndim - nb - 1])
    comp = tuple(X.shape[X.ndim - nb - 1:-1])
    k = _split_components(comp, mat_batch)
    if k is None:
        X = X.expand(sample_shape + tuple(mat_batch) + (D,))
        k = 0
    S, Bo, Bi = _prod(sample_shape), _prod(mat_batch[:k]), _prod(mat_batch[k:])
    return X.reshape(S, Bi, D).contiguous(), sample_shape, Bo, Bi


def quadform_loglike(X, P, b, c):
    """K3a: -1/2 x^T P x + x^T b + c for every (sample, component).
    X: sample + comp* + (D,), comp* broadcastable to c.shape; P: c.shape+(D,D); b: c.shape+(D,)."""
    dev = L.require_device(X, P, b, c)
    lib = L.load()
    D = P.shape[-1]
    mat_batch = tuple(c.shape)
    dt = P.dtype
    X2, sample_shape, Bo, Bi = _dense_samples(X.to(dt), mat_batch, D)
    S = X2.shape[0]
    Pc = P.expand(mat_batch + (D, D)).contiguous()
    bc = b.expand(mat_batch + (D,)).contiguous()
    cc = c.contiguous()
    out = torch.empty((S, Bo, Bi), dtype=dt, device=dev)
    if out.numel() > 0:
        fn = getattr(lib, "vbmp_quadform_loglike_" + L.suffix(dt))
        L.call(fn, "vbmp_quadform_loglike", L.ptr(X2), S, Bo, Bi, D, L.ptr(Pc), L.ptr(bc), L.ptr(cc), L.ptr(out), L.stream_ptr(dev))
    return out.reshape(sample_shape + mat_batch)


def estep_sym_serves(S, K, D, dt):
    """does the symmetric-packed form of K3 (`vbmp_mixture_estep_sym`, the only form that also returns the per-sample evidence)
    serve S samples of K components in D dimensions?  Callers test this BEFORE assembling operands they would throw away."""
    return S > 0 and (D in (4, 8, 16) or (D == 32 and dt == torch.float32)) and K <= ESTEP_SYM_MAX_K and S >= 4096 \
        and not _estep_sym_off


def mixture_estep(X, P, b, c, want_lse=False):
    """K3: fused responsibilities for a K-component mixture over dense samples X (S,D).
    c must already include E log pi.  Returns p (S,K), NA (K), logZ ().
    want_lse: also the per-sample evidence lse (S,) as a fourth value -- or None (nothing launched) when the shape is outside the
    kernel form that provides it (the caller then composes the E-step)."""
    dev = L.require_device(X, P, b, c)
    lib = L.load()
    K, D = P.shape[0], P.shape[-1]
    dt = P.dtype
    Xc = X.to(dt).contiguous()
    S = Xc.shape[0]
    sym = estep_sym_serves(S, K, D, dt)
    if want_lse and not sym:
        return None
    p = torch.empty((S, K), dtype=dt, device=dev)
    acc = torch.zeros(K + 1, dtype=dt, device=dev)
    lse = torch.empty(S, dtype=dt, device=dev) if want_lse else None
    Pc, bc, cc = P.contiguous(), b.contiguous(), c.contiguous()  # named: a temporary's block could be reused before the launch
    if sym:
        # few components, many samples: symmetric-packed precisions as scalar operands (half the multiply-adds of x' P x,
        # several samples per thread), see k_estep_sym
        iu = _triu_idx(D, dev)
        Ps = Pc + Pc.transpose(-1, -2)
        Ps.diagonal(dim1=-2, dim2=-1).mul_(0.5)
        Qc = Ps[:, iu[0], iu[1]].contiguous()
        fn = getattr(lib, "vbmp_mixture_estep_sym_" + L.suffix(dt))
        L.call(fn, "vbmp_mixture_estep", L.ptr(Xc), S, K, D, L.ptr(Qc), L.ptr(bc), L.ptr(cc), L.ptr(p),
               L.ptr(acc), ctypes.c_void_p(acc.data_ptr() + K * acc.element_size()), L.ptr(lse), L.stream_ptr(dev))
    elif S > 0:
        fn = getattr(lib, "vbmp_mixture_estep_" + L.suffix(dt))
        L.call(fn, "vbmp_mixture_estep", L.ptr(Xc), S, K, D, L.ptr(Pc), L.ptr(bc), L.ptr(cc), L.ptr(p),
                   L.ptr(acc), ctypes.c_void_p(acc.data_ptr() + K * acc.element_size()), L.stream_ptr(dev))
    return (p, acc[:K], acc[K], lse) if want_lse else (p, acc[:K], acc[K])


ESTEP_SYM_MAX_K = 8
_estep_sym_off = False  # tests / A-B timing: force the other E-step forms
_TRIU = {}


def _triu_idx(D, dev):
    key = (D, str(dev))
    if key not in _TRIU:
        _TRIU[key] = torch.triu_indices(D, D, device=dev)
    return _TRIU[key]


def weighted_moments(X, pv, n_sample_dims, mat_batch):
    """K4: N = sum w, SEx = sum w x, SExx = sum w x x^T over the sample axes.
    X: sample + comp* + (D,); pv: sample + (axes broadcastable to mat_batch) or None (unit weights)."""
    dev = L.require_device(X, pv)
    lib = L.load()
    D = X.shape[-1]
    mat_batch = tuple(mat_batch)
    dt = X.dtype
    if pv is None:
        sample_shape = tuple(X.shape[:n_sample_dims])
        X = X.expand(sample_shape + mat_batch + (D,))
    X2, sample_shape, Bo, Bi = _dense_samples(X, mat_batch, D)
    assert len(sample_shape) == n_sample_dims
    S = X2.shape[0]
    p2 = None
    if pv is not None:
        p2 = pv.to(dt).expand(sample_shape + mat_batch).reshape(S, Bo, Bi).contiguous()
    fn = getattr(lib, "vbmp_weighted_moments_" + L.suffix(dt))

    def launch(Xc, pc, bo, bi):
        nB = bo * bi
        buf = torch.zeros(nB * (1 + D + D * D), dtype=dt, device=dev)
        Nk, SEx, SExx = buf[:nB], buf[nB:nB * (1 + D)], buf[nB * (1 + D):]
        if S > 0 and nB > 0:
            L.call(fn, "vbmp_weighted_moments", L.ptr(Xc), L.ptr(pc), S, bo, bi, D, L.ptr(Nk), L.ptr(SEx), L.ptr(SExx), L.stream_ptr(dev))
        return Nk.reshape(bo, bi), SEx.reshape(bo, bi, D), SExx.reshape(bo, bi, D, D)
    MAXI = 65535  # inner components per launch (grid.y of the kernel): a longer component axis goes in slices
    if Bi <= MAXI:
        Nk, SEx, SExx = launch(X2, p2, Bo, Bi)
    else:
        parts = []
        for c0 in range(0, Bi, MAXI):
            c1 = min(Bi, c0 + MAXI)
            pc = None if p2 is None else p2[:, :, c0:c1].contiguous()
            parts.append(launch(X2[:, c0:c1].contiguous(), pc, Bo, c1 - c0))
        Nk, SEx, SExx = (torch.cat([q[i] for q in parts], dim=1) for i in range(3))
    return Nk.reshape(mat_batch), SEx.reshape(mat_batch + (D,)), SExx.reshape(mat_batch + (D, D))


def _aligned(t):
    t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


def _norm3(X, T, sample_shape, bo_shape, inner):
    """Operand broadcastable to (T,)+sample+bo+inner -> (dense tensor, (st_t, st_s, st_b)) with stride 0 on
    the axis groups (time / sample / batch) it does not depend on."""
    lead = (T,) + tuple(sample_shape) + tuple(bo_shape)
    ni = len(inner)
    if X.ndim < len(lead) + ni:
        X = X.reshape((1,) * (len(lead) + ni - X.ndim) + tuple(X.shape))
    ns, nb = len(sample_shape), len(bo_shape)
    idx, tgt, dep = [], [], []
    for lo, hi in ((0, 1), (1, 1 + ns), (1 + ns, 1 + ns + nb)):
        d = any(X.shape[i] != 1 and X.stride(i) != 0 for i in range(lo, hi))
        dep.append(d)
        for i in range(lo, hi):
            idx.append(slice(None) if d else slice(0, 1))
            tgt.append(lead[i] if d else 1)
    Xc = X[tuple(idx)].expand(tuple(tgt) + tuple(inner))
    n_in = _prod(inner)
    Ss = _prod(sample_shape) if dep[1] else 1
    Nb = _prod(bo_shape) if dep[2] else 1
    Xc = _aligned(Xc.reshape(-1))
    return Xc, (Ss * Nb * n_in if dep[0] else 0, Nb * n_in if dep[1] else 0, n_in if dep[2] else 0)


def lds_smoother(T, sample_shape, bo_shape, H, invQ, ATQA_xx, QA_xp_x, A_Elogdet, x0_P, x0_eta, x0_res,
                 like_P, like_eta, like_res, cu1, cu2, cu3, sums_only=False, y=None, fixed_point="auto"):
    """K9: one persistent launch of the information filter + smoother.
    sums_only=True (what update_latents needs): only slot T-1 of the returned Sigma_t_tp1 is meaningful (the rest is the
    sweeps' work buffer) and logZ has ONE time step holding its sum over time.
    y (observations, broadcastable to (T,)+sample+bo+(nobs,), nobs <= 16): where the device form supports it the result also
    holds "sum_mu" = sum_t mu[t] and "sum_xy" = sum_t mu[t] y[t]' (absent otherwise: the caller then uses K10).
    fixed_point (include/vbmp_hip.h, "Fixed-point shortcut"): "auto" = stop the matrix recursions once they have converged
    (time-independent likelihood precision only; outputs equal the literal recursion's to its last-bit wander), "exact" = only
    at a bitwise repeat (bit for bit the literal recursion), "off" = the literal recursion at every step.
    System / prior parameters: bo_shape + (...).  Per-step operands: broadcastable to (T,)+sample+bo+(...).
    Returns dict of dense outputs shaped (T,)+sample+bo+(...) (and sample+bo+(...) for the x0 terms)."""
    dev = L.require_device(invQ, like_eta)
    lib = L.load()
    dt = like_eta.dtype
    if H > L.LDS_MAX_H and not (H <= L.LDS_MAX_H_BLOCK and L.lds_block_fits(H, dt.itemsize)):
        raise L.VbmpHipError(f"hidden_dim {H}: beyond the persistent smoother kernels (compose the recursion instead)")
    suf = L.suffix(dt)
    sample_shape, bo_shape = tuple(sample_shape), tuple(bo_shape)
    NB = _prod(bo_shape)
    S = _prod(sample_shape) * NB

    def per_b(t, inner):
        return _aligned(t.to(dt).expand(bo_shape + tuple(inner)).reshape((NB,) + tuple(inner)))
    keep = [per_b(invQ, (H, H)), per_b(ATQA_xx, (H, H)), per_b(QA_xp_x, (H, H)), per_b(A_Elogdet, ()),
            per_b(x0_P, (H, H)), per_b(x0_eta, (H,)), per_b(x0_res, ())]
    steps = [_norm3(like_P.to(dt), T, sample_shape, bo_shape, (H, H)), _norm3(like_eta.to(dt), T, sample_shape, bo_shape, (H,)),
             _norm3(like_res.to(dt), T, sample_shape, bo_shape, ()), _norm3(cu1.to(dt), T, sample_shape, bo_shape, (H,)),
             _norm3(cu2.to(dt), T, sample_shape, bo_shape, (H,)), _norm3(cu3.to(dt), T, sample_shape, bo_shape, ())]
    lead = (T,) + sample_shape + bo_shape
    out = {"invSigma": torch.empty(lead + (H, H), dtype=dt, device=dev),
           "invSigmamu": torch.empty(lead + (H,), dtype=dt, device=dev),
           "Sigma": torch.empty(lead + (H, H), dtype=dt, device=dev),
           "mu": torch.empty(lead + (H,), dtype=dt, device=dev),
           "Sigma_t_tp1": torch.empty(lead + (H, H), dtype=dt, device=dev),
           "logZ": torch.empty(((1,) + lead[1:]) if sums_only else lead, dtype=dt, device=dev),
           "Sigma_x0_x0": torch.empty(lead[1:] + (H, H), dtype=dt, device=dev),
           "mu_x0": torch.empty(lead[1:] + (H,), dtype=dt, device=dev),
           # time-integrated second moments, accumulated in the backward sweep's registers
           "sum_xx": torch.empty(lead[1:] + (H, H), dtype=dt, device=dev),
           "sum_xpx": torch.empty(lead[1:] + (H, H), dtype=dt, device=dev)}
    a = L.LDS_ARGS[suf]()
    a.T, a.S, a.NB, a.H = T, S, NB, H
    if fixed_point not in L.LDS_FIXED_POINT:
        raise ValueError(f"fixed_point must be one of {sorted(L.LDS_FIXED_POINT)}, not {fixed_point!r}")
    a.flags = ((L.LDS_CROSS_WORK | L.LDS_LOGZ_SUM) if sums_only else 0) | L.LDS_FIXED_POINT[fixed_point]
    for name, t in zip(("invQ", "ATQA_xx", "QA_xp_x", "A_Elogdet", "x0_P", "x0_eta", "x0_res"), keep):
        setattr(a, name, t.data_ptr())
    for (name, pre), (t, st) in zip((("like_P", "lP"), ("like_eta", "le"), ("like_res", "lr"), ("cu1", "c1"),
                                     ("cu2", "c2"), ("cu3", "c3")), steps):
        setattr(a, name, t.data_ptr())
        setattr(a, pre + "_t", st[0])
        setattr(a, pre + "_s", st[1])
        setattr(a, pre + "_b", st[2])
    for name, t in out.items():
        setattr(a, name, t.data_ptr())
    if S > 0 and T > 0 and (getattr(lib, "vbmp_lds_smoother_caps_" + suf)(ctypes.byref(a)) & L.LDS_CAP_OBS_SUMS):
        out["sum_mu"] = torch.empty(lead[1:] + (H,), dtype=dt, device=dev)
        a.sum_mu = out["sum_mu"].data_ptr()
        if y is not None and 1 <= y.shape[-1] <= 16:
            nobs = y.shape[-1]
            yk, yst = _norm3(y.to(dt), T, sample_shape, bo_shape, (nobs,))
            steps.append(yk)  # kept alive until the launch
            out["sum_xy"] = torch.empty(lead[1:] + (H, nobs), dtype=dt, device=dev)
            a.y, a.y_t, a.y_s, a.y_b, a.nobs = yk.data_ptr(), yst[0], yst[1], yst[2], nobs
            a.sum_xy = out["sum_xy"].data_ptr()
    if S > 0 and T > 0:
        fn = getattr(lib, "vbmp_lds_smoother_" + suf)
        L.call(fn, "vbmp_lds_smoother", ctypes.byref(a), L.stream_ptr(dev))
    return out


def _ts_view(X, T, series_shape, d):
    """X broadcastable to (T,)+series_shape+(d,) -> (tensor kept alive, data_ptr, st_t, st_s): dense along
    the axes it depends on, stride 0 elsewhere (time and/or series)."""
    lead = (T,) + tuple(series_shape)
    if X.ndim < len(lead) + 1:
        X = X.reshape((1,) * (len(lead) + 1 - X.ndim) + tuple(X.shape))
    dep_t = X.shape[0] != 1 and X.stride(0) != 0
    ns = len(series_shape)
    dep_s = any(X.shape[i] != 1 and X.stride(i) != 0 for i in range(1, 1 + ns))
    idx = (slice(None) if dep_t else slice(0, 1),) + tuple(slice(None) if dep_s else slice(0, 1) for _ in range(ns))
    tgt = ((T,) if dep_t else (1,)) + (tuple(series_shape) if dep_s else (1,) * ns) + (d,)
    Xc = X[idx].expand(tgt).contiguous()
    S = _prod(series_shape) if dep_s else 1
    return Xc, (S * d if dep_t else 0), (d if dep_s else 0)


def tsum_outer(a, b, M=None, a_from=0, b_from=0, steps=None, series_shape=None):
    """K10: out[series,i,j] = sum_t a[t+a_from, series, i] * b[t+b_from, series, j] (+ sum_t M[t, series, i, j]).
    a: (T,)+series+(da,), b: (T,)+series+(db,) (either may be broadcast along time or series)."""
    dev = L.require_device(a, b, M)
    lib = L.load()
    dt = a.dtype
    T = max(a.shape[0], b.shape[0])
    if series_shape is None:
        series_shape = tuple(torch.broadcast_shapes(a.shape[1:-1], b.shape[1:-1]))
    da, db = a.shape[-1], b.shape[-1]
    S = _prod(series_shape)
    ac, sa_t, sa_s = _ts_view(a, T, series_shape, da)
    bc, sb_t, sb_s = _ts_view(b.to(dt), T, series_shape, db)
    if steps is None:
        steps = T - max(a_from, b_from)
    esz = ac.element_size()
    Mc, sM_t, sM_s, mptr = None, 0, 0, 0
    if M is not None:
        Mc = M.expand((M.shape[0],) + tuple(series_shape) + (da, db)).contiguous()
        sM_t, sM_s, mptr = S * da * db, da * db, Mc.data_ptr()
    out = torch.empty(tuple(series_shape) + (da, db), dtype=dt, device=dev)
    if out.numel() > 0:
        fn = getattr(lib, "vbmp_tsum_outer_" + L.suffix(dt))
        L.call(fn, "vbmp_tsum_outer", ctypes.c_void_p(ac.data_ptr() + a_from * sa_t * esz), sa_t, sa_s, da,
               ctypes.c_void_p(bc.data_ptr() + b_from * sb_t * esz), sb_t, sb_s, db, ctypes.c_void_p(mptr), sM_t, sM_s,
               steps, S, L.ptr(out), L.stream_ptr(dev))
    return out


def _norm2(X, sample_shape, bshape, inner):
    """Operand broadcastable to sample+bshape+inner -> (dense tensor, st_s, st_b); stride 0 along the group
    of axes (samples / experts) it does not depend on."""
    lead = tuple(sample_shape) + tuple(bshape)
    ni = len(inner)
    if X.ndim < len(lead) + ni:
        X = X.reshape((1,) * (len(lead) + ni - X.ndim) + tuple(X.shape))
    ns, nb = len(sample_shape), len(bshape)
    idx, tgt, dep = [], [], []
    for lo, hi in ((0, ns), (ns, ns + nb)):
        d = any(X.shape[i] != 1 and X.stride(i) != 0 for i in range(lo, hi))
        dep.append(d)
        for i in range(lo, hi):
            idx.append(slice(None) if d else slice(0, 1))
            tgt.append(lead[i] if d else 1)
    Xc = _aligned(X[tuple(idx)].expand(tuple(tgt) + tuple(inner)).reshape(-1))
    n_in = _prod(inner)
    Nb = _prod(bshape) if dep[1] else 1
    return Xc, (Nb * n_in if dep[0] else 0), (n_in if dep[1] else 0)


def mnw_message_fusable(m, d):
    """output dim m, elimination dim d"""
    n, p = m, d
    def pad(d):
        return 1 if d <= 1 else 2 if d <= 2 else 4 if d <= 4 else 8 if d <= 8 else 16 if d <= 16 else 32 if d <= 32 else 64
    return pad(n) <= pad(p) <= L.MNW_MAX_DIM


def mnw_message(P, e1, e2, e3, Add1, Add2, M, C, cvec, sign, bshape, res_w=None, res_c=None, add_cvec=False):
    """K7/K8 sandwich kernel (see include/vbmp_hip.h).  P: sample+bshape*+(d,d); e1/e2/e3: sample+bshape*+(d,);
    Add1/Add2 (bshape,d,d); M (bshape,m,d); C (bshape,m,m); cvec (bshape,m) or None.
    Returns ovec lead+(m,), omat lead+(m,m), scal lead+(8,) with lead = sample+bshape.
    res_w (8 Python floats): also returns res lead = res_c (bshape, or None) + sum_k res_w[k] scal[..., k], formed in the kernel;
    add_cvec: ovec + cvec is returned in place of ovec."""
    dev = L.require_device(P, e1, e2, Add1, M, C)
    lib = L.load()
    dt = P.dtype
    bshape = tuple(bshape)
    m, d = M.shape[-2], M.shape[-1]
    nb = len(bshape)
    shapes = [P.shape[:-2], e1.shape[:-1], e2.shape[:-1], bshape] + ([e3.shape[:-1]] if e3 is not None else [])
    lead = tuple(torch.broadcast_shapes(*shapes))
    sample_shape = lead[:len(lead) - nb]
    S, NB = _prod(sample_shape), _prod(bshape)
    Pc, sP_s, sP_b = _norm2(P, sample_shape, bshape, (d, d))
    e1c, s1_s, s1_b = _norm2(e1.to(dt), sample_shape, bshape, (d,))
    e2c, s2_s, s2_b = _norm2(e2.to(dt), sample_shape, bshape, (d,))
    e3c, s3_s, s3_b = (None, 0, 0) if e3 is None else _norm2(e3.to(dt), sample_shape, bshape, (d,))
    A1 = _aligned(Add1.to(dt).expand(bshape + (d, d)))
    A2 = None if Add2 is None else _aligned(Add2.to(dt).expand(bshape + (d, d)))
    Mc = _aligned(M.to(dt).expand(bshape + (m, d)))
    Cc = _aligned(C.to(dt).expand(bshape + (m, m)))
    cv = None if cvec is None else _aligned(cvec.to(dt).expand(bshape + (m,)))
    ovec = torch.empty(lead + (m,), dtype=dt, device=dev)
    omat = torch.empty(lead + (m, m), dtype=dt, device=dev)
    scal = torch.empty(lead + (8,), dtype=dt, device=dev)
    res = torch.empty(lead, dtype=dt, device=dev) if res_w is not None else None
    rc = None if (res_c is None or res_w is None) else res_c.to(dt).expand(bshape).contiguous()
    if S > 0 and NB > 0:
        suf = L.suffix(dt)
        cT = L.DTYPES[suf][1]
        if res_w is None and not add_cvec:
            fn = getattr(lib, "vbmp_mnw_message_" + suf)
            L.call(fn, "vbmp_mnw_message", L.ptr(Pc), sP_s, sP_b, L.ptr(e1c), s1_s, s1_b, L.ptr(e2c), s2_s, s2_b, L.ptr(e3c),
                   s3_s, s3_b, L.ptr(A1), L.ptr(A2), L.ptr(Mc), L.ptr(Cc), L.ptr(cv), cT(float(sign)), L.ptr(ovec),
                   L.ptr(omat), L.ptr(scal), S, NB, m, d, L.stream_ptr(dev))
        else:
            fn = getattr(lib, "vbmp_mnw_message_res_" + suf)
            w8 = (cT * 8)(*[float(v) for v in (res_w if res_w is not None else [0.0] * 8)])
            L.call(fn, "vbmp_mnw_message", L.ptr(Pc), sP_s, sP_b, L.ptr(e1c), s1_s, s1_b, L.ptr(e2c), s2_s, s2_b, L.ptr(e3c),
                   s3_s, s3_b, L.ptr(A1), L.ptr(A2), L.ptr(Mc), L.ptr(Cc), L.ptr(cv), cT(float(sign)), L.ptr(ovec),
                   L.ptr(omat), L.ptr(scal), S, NB, m, d, ctypes.cast(w8, ctypes.c_void_p) if res_w is not None else None,
                   L.ptr(rc), L.ptr(res), 1 if add_cvec else 0, L.stream_ptr(dev))
    return (ovec, omat, scal) if res_w is None else (ovec, omat, scal, res)


def hmm_forward_backward(logits, trans, init, batch_shape, ptemp=1.0):
    """K11.  logits: (T,)+sample+batch+(K,); trans: batch+(K,K); init: batch+(K,).
    Returns p (same shape as logits), SEzz sample+batch+(K,K), SEz0 sample+batch+(K,), logZ sample+batch."""
    dev = L.require_device(logits, trans, init)
    lib = L.load()
    dt = logits.dtype
    K = logits.shape[-1]
    T = logits.shape[0]
    lead = tuple(logits.shape[1:-1])
    C, NB = _prod(lead), _prod(batch_shape)
    lg = logits.contiguous()
    tr = trans.to(dt).expand(tuple(batch_shape) + (K, K)).contiguous()
    ini = init.to(dt).expand(tuple(batch_shape) + (K,)).contiguous()
    p = torch.empty_like(lg)
    SEzz = torch.empty(lead + (K, K), dtype=dt, device=dev)
    SEz0 = torch.empty(lead + (K,), dtype=dt, device=dev)
    logZ = torch.empty(lead, dtype=dt, device=dev)
    if C > 0:
        suf = L.suffix(dt)
        fn = getattr(lib, "vbmp_hmm_forward_backward_" + suf)
        cT = L.DTYPES[suf][1]
        L.call(fn, "vbmp_hmm_forward_backward", L.ptr(lg), L.ptr(tr), L.ptr(ini), T, C, NB, K, cT(float(ptemp)), L.ptr(p),
               L.ptr(SEzz), L.ptr(SEz0), L.ptr(logZ), L.stream_ptr(dev))
    return p, SEzz, SEz0, logZ


def weighted_matsum(C, w=None):
    """K5b: sum_s w[s] * C[s] for C (S, ...) dense and w (S,) or None; returns C.shape[1:]."""
    dev = L.require_device(C, w)
    lib = L.load()
    dt = C.dtype
    Cc = _aligned(C)
    S = Cc.shape[0]
    inner = tuple(Cc.shape[1:])
    E = _prod(inner)
    wc = None if w is None else w.to(dt).contiguous()
    out = torch.zeros(E, dtype=dt, device=dev)
    if S > 0 and E > 0:
        fn = getattr(lib, "vbmp_weighted_matsum_" + L.suffix(dt))
        L.call(fn, "vbmp_weighted_matsum", L.ptr(Cc), L.ptr(wc), S, E, L.ptr(out), L.stream_ptr(dev))
    return out.reshape(inner)


ROWS_MAX_DIM = 64


def rows_affine(X, M, c=None):
    """K12: out[s] = M @ X[s] (+ c) for X (S, k) dense, M (n, k), c (n,) or None; returns (S, n)."""
    dev = L.require_device(X, M, c)
    lib = L.load()
    dt = X.dtype
    Xc, Mc = X.contiguous(), M.to(dt).contiguous()
    cc = None if c is None else c.to(dt).contiguous()
    S, k = Xc.shape
    n = Mc.shape[0]
    assert Mc.shape == (n, k) and (cc is None or cc.shape == (n,))
    out = torch.empty(S, n, dtype=dt, device=dev)
    if S > 0:
        fn = getattr(lib, "vbmp_rows_affine_" + L.suffix(dt))
        L.call(fn, "vbmp_rows_affine", L.ptr(Xc), S, k, L.ptr(Mc), L.ptr(cc), n, L.ptr(out), L.stream_ptr(dev))
    return out


ROWS_QUAD_MAX_K = 16


def rows_affine_quad(X, M, c, P, b, c0):
    """K12 + the quadratic form of the same rows in one pass: returns (M @ X[s] + c, -1/2 X[s]' P X[s] + b' X[s] + c0) for X (S, k),
    M (n, k), c (n,) or None, P (k, k), b (k,) or None, c0 a 0-d / 1-element tensor or None; (S, n) and (S,)."""
    dev = L.require_device(X, M, c, P, b, c0)
    lib = L.load()
    dt = X.dtype
    Xc, Mc, Pc = X.contiguous(), M.to(dt).contiguous(), P.to(dt).contiguous()
    cc = None if c is None else c.to(dt).contiguous()
    bc = None if b is None else b.to(dt).contiguous()
    c0c = None if c0 is None else c0.to(dt).reshape(1).contiguous()
    S, k = Xc.shape
    n = Mc.shape[0]
    assert Mc.shape == (n, k) and Pc.shape == (k, k) and (cc is None or cc.shape == (n,)) and (bc is None or bc.shape == (k,))
    out = torch.empty(S, n, dtype=dt, device=dev)
    q = torch.empty(S, dtype=dt, device=dev)
    if S > 0:
        fn = getattr(lib, "vbmp_rows_affine_quad_" + L.suffix(dt))
        L.call(fn, "vbmp_rows_affine_quad", L.ptr(Xc), S, k, L.ptr(Mc), L.ptr(cc), n, L.ptr(out), L.ptr(Pc), L.ptr(bc), L.ptr(c0c),
               L.ptr(q), L.stream_ptr(dev))
    return out, q


def niw_estep_params(U, nu, mu, lam, logdet_invU, alpha=None):
    """K13: (P, b, c) with -1/2 x' P x + x' b + c = E log N(x | component k) (+ E log pi_k when the Dirichlet counts `alpha` are
    given) for K Normal-inverse-Wishart components, one launch.  U (K,D,D), nu / lam / logdet_invU / alpha (K), mu (K,D)."""
    dev = L.require_device(U, nu, mu, lam, logdet_invU, alpha)
    lib = L.load()
    dt = U.dtype
    K, D = mu.shape
    assert U.shape == (K, D, D)
    ops_in = [t.to(dt).reshape(sh).contiguous() for t, sh in ((U, (K, D, D)), (nu, (K,)), (mu, (K, D)), (lam, (K,)), (logdet_invU, (K,)))]
    al = None if alpha is None else alpha.to(dt).reshape(K).contiguous()
    P = torch.empty(K, D, D, dtype=dt, device=dev)
    b = torch.empty(K, D, dtype=dt, device=dev)
    c = torch.empty(K, dtype=dt, device=dev)
    if K > 0:
        fn = getattr(lib, "vbmp_niw_estep_params_" + L.suffix(dt))
        L.call(fn, "vbmp_niw_estep_params", *[L.ptr(t) for t in ops_in], L.ptr(al), K, D, L.ptr(P), L.ptr(b), L.ptr(c), L.stream_ptr(dev))
    return P, b, c


def _prior_operand(t, lead, tail, dt):
    """(tensor, batch stride in elements) of a prior operand of shape lead + tail for the K15 kernels: stride 0 when it is ONE
    element expanded over the batch (the reference's expanded priors), prod(tail) when dense; anything else is made dense"""
    t = t.to(dt) if t.dtype != dt else t
    t = t.expand(tuple(lead) + tuple(tail))
    nl = len(lead)
    inner = t
    if nl and all(st == 0 or sz == 1 for st, sz in zip(t.stride()[:nl], t.shape[:nl])):
        inner = t[(0,) * nl]
        if inner.is_contiguous():
            return inner, 0
    if t.is_contiguous():
        return t, _prod(tail)
    return t.contiguous(), _prod(tail)


def _dense(t, shape, dt):
    t = t.to(dt) if t.dtype != dt else t
    return t.expand(tuple(shape)).contiguous()


def dirichlet_kl(alpha, alpha0, event_dim):
    """K15: Dirichlet.KLqprior per batch element (the trailing event_dim axes are one event)"""
    dev = L.require_device(alpha, alpha0)
    lib, dt = L.load(), alpha.dtype
    lead, ev = tuple(alpha.shape[:alpha.ndim - event_dim]), tuple(alpha.shape[alpha.ndim - event_dim:])
    NB, K = _prod(lead), _prod(ev)
    a = _dense(alpha, lead + ev, dt)
    a0, s0 = _prior_operand(alpha0, lead, ev, dt)
    out = torch.empty(lead, dtype=dt, device=dev)
    if NB > 0:
        L.call(getattr(lib, "vbmp_dirichlet_kl_" + L.suffix(dt)), "vbmp_dirichlet_kl", L.ptr(a), L.ptr(a0), s0, NB, K, L.ptr(out),
               L.stream_ptr(dev))
    return out


def gamma_kl(alpha, beta, alpha0, beta0, event_dim):
    """K15: Gamma.KLqprior summed over the trailing event_dim axes"""
    dev = L.require_device(alpha, beta, alpha0, beta0)
    lib, dt = L.load(), alpha.dtype
    lead, ev = tuple(alpha.shape[:alpha.ndim - event_dim]), tuple(alpha.shape[alpha.ndim - event_dim:])
    NB, K = _prod(lead), _prod(ev)
    a, b = _dense(alpha, lead + ev, dt), _dense(beta, lead + ev, dt)
    a0, sa0 = _prior_operand(alpha0, lead, ev, dt)
    b0, sb0 = _prior_operand(beta0, lead, ev, dt)
    out = torch.empty(lead, dtype=dt, device=dev)
    if NB > 0:
        L.call(getattr(lib, "vbmp_gamma_kl_" + L.suffix(dt)), "vbmp_gamma_kl", L.ptr(a), L.ptr(b), L.ptr(a0), L.ptr(b0), sa0, sb0, NB,
               K, L.ptr(out), L.stream_ptr(dev))
    return out


def wishart_kl(invU0, U, nu, nu0, ld, ld0, mu=None, mu0=None, lam=None, lam0=None):
    """K15: Wishart.KLqprior per n x n element (lead = U.shape[:-2]); with mu .. lam0 also the Normal part of
    NormalInverseWishart.KLqprior (mu, mu0: lead + (n,), lam, lam0: lead)"""
    dev = L.require_device(invU0, U, nu, nu0, ld, ld0, mu, mu0, lam, lam0)
    lib, dt = L.load(), U.dtype
    lead, n = tuple(U.shape[:-2]), U.shape[-1]
    NB = _prod(lead)
    Ud, nud, ldd = _dense(U, lead + (n, n), dt), _dense(nu, lead, dt), _dense(ld, lead, dt)
    I0, sm0 = _prior_operand(invU0, lead, (n, n), dt)
    n0, sn0 = _prior_operand(nu0, lead, (), dt)
    l0, sl0 = _prior_operand(ld0, lead, (), dt)
    if mu is not None:
        mud, lamd = _dense(mu, lead + (n,), dt), _dense(lam, lead, dt)
        m0, smu0 = _prior_operand(mu0, lead, (n,), dt)
        la0, slam0 = _prior_operand(lam0, lead, (), dt)
        extra = (L.ptr(mud), L.ptr(m0), smu0, L.ptr(lamd), L.ptr(la0), slam0)
    else:
        extra = (None, None, 0, None, None, 0)
    out = torch.empty(lead, dtype=dt, device=dev)
    if NB > 0:
        L.call(getattr(lib, "vbmp_wishart_kl_" + L.suffix(dt)), "vbmp_wishart_kl", L.ptr(I0), sm0, L.ptr(Ud), L.ptr(nud), L.ptr(n0),
               sn0, L.ptr(ldd), L.ptr(l0), sl0, *extra, NB, n, L.ptr(out), L.stream_ptr(dev))
    return out


MN_KL_MAX_LDS = 8192  # doubles of LDS the matrix-normal KL kernel stages its operands in: n p + p max(n, p)


def mn_kl_serves(n, p):
    return n * p + p * max(n, p) <= MN_KL_MAX_LDS


def mn_kl(mu, mu0, invV0, V, R, ldV, ldV0, xm=None):
    """K15: the matrix-normal part of MatrixNormalWishart / MatrixNormalGamma.KLqprior; mu: lead + (n, p), R: lead + (n, n)"""
    dev = L.require_device(mu, mu0, invV0, V, R, ldV, ldV0)
    lib, dt = L.load(), mu.dtype
    lead, (n, p) = tuple(mu.shape[:-2]), mu.shape[-2:]
    NB = _prod(lead)
    mud, Vd, Rd, ldd = _dense(mu, lead + (n, p), dt), _dense(V, lead + (p, p), dt), _dense(R, lead + (n, n), dt), _dense(ldV, lead, dt)
    m0, smu0 = _prior_operand(mu0, lead, (n, p), dt)
    I0, sv0 = _prior_operand(invV0, lead, (p, p), dt)
    l0, sl0 = _prior_operand(ldV0, lead, (), dt)
    xmo, sxm = (None, 0) if xm is None else _prior_operand(xm, lead, (), dt)  # set entries of X_mask per batch element
    out = torch.empty(lead, dtype=dt, device=dev)
    if NB > 0:
        L.call(getattr(lib, "vbmp_mn_kl_" + L.suffix(dt)), "vbmp_mn_kl", L.ptr(mud), L.ptr(m0), smu0, L.ptr(I0), sv0, L.ptr(Vd),
               L.ptr(Rd), L.ptr(ldd), L.ptr(l0), sl0, L.ptr(xmo), sxm, NB, n, p, L.ptr(out), L.stream_ptr(dev))
    return out


def mnw_expectations(mu, U, nu, V, logdet_invU):
    """K14: (EinvSigma, EinvUX, EXTinvUX, ElogdetinvSigma) of a MatrixNormalWishart posterior, one launch.  mu: lead + (n, p),
    U: lead + (n, n), nu / logdet_invU: lead, V: lead + (p, p) (operands may be broadcast over lead)."""
    dev = L.require_device(mu, U, nu, V, logdet_invU)
    lib = L.load()
    dt = mu.dtype
    lead = tuple(mu.shape[:-2])
    n, p = mu.shape[-2:]
    NB = _prod(lead)
    ins = [mu.reshape(NB, n, p).contiguous(), U.to(dt).expand(lead + (n, n)).reshape(NB, n, n).contiguous(),
           nu.to(dt).expand(lead).reshape(NB).contiguous(), V.to(dt).expand(lead + (p, p)).reshape(NB, p, p).contiguous(),
           logdet_invU.to(dt).expand(lead).reshape(NB).contiguous()]
    R = torch.empty(lead + (n, n), dtype=dt, device=dev)
    G = torch.empty(lead + (n, p), dtype=dt, device=dev)
    H = torch.empty(lead + (p, p), dtype=dt, device=dev)
    El = torch.empty(lead, dtype=dt, device=dev)
    if NB > 0:
        fn = getattr(lib, "vbmp_mnw_expectations_" + L.suffix(dt))
        L.call(fn, "vbmp_mnw_expectations", *[L.ptr(t) for t in ins], NB, n, p, L.ptr(R), L.ptr(G), L.ptr(H), L.ptr(El), L.stream_ptr(dev))
    return R, G, H, El


MATSUM_MAX_COLS = 32


def weighted_matsum_cols(C, W):
    """K5b, several weight columns: out[b] = sum_s W[s, b] * C[s] for C (S, ...) dense and W (S, NB); returns
    (NB,) + C.shape[1:] (the library GEMM W^T @ C beyond MATSUM_MAX_COLS columns)."""
    dev = L.require_device(C, W)
    dt = C.dtype
    S, NB = W.shape
    inner = tuple(C.shape[1:])
    E = _prod(inner)
    if NB > MATSUM_MAX_COLS or S == 0:
        return (W.to(dt).transpose(0, 1) @ C.reshape(S, E)).reshape((NB,) + inner)
    lib = L.load()
    Cc = C.contiguous()
    Wc = W.to(dt).contiguous()
    out = torch.zeros(NB, E, dtype=dt, device=dev)
    fn = getattr(lib, "vbmp_weighted_matsum_cols_" + L.suffix(dt))
    L.call(fn, "vbmp_weighted_matsum_cols", L.ptr(Cc), L.ptr(Wc), S, E, NB, L.ptr(out), L.stream_ptr(dev))
    return out.reshape((NB,) + inner)
