"""GPU parity: K1 (batched SPD inverse + logdet) and K2 (fused Wishart / NIW ss_update) through the
product classes, against (a) the golden fixtures captured from the reference and (b) the CPU oracle
on seeded inputs.  Tolerances are the north-star ones: 1e-10 normwise in fp64, 1e-4 in fp32."""
import pytest
import torch

from tests.helpers import TOL32, TOL64, assert_close

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _spd(B, D, dtype, seed, jitter=0.5):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(B, D, D + 3, generator=g, dtype=torch.float64)
    S = A @ A.transpose(-2, -1) / (D + 3) + jitter * torch.eye(D, dtype=torch.float64)
    return S.to(dtype)


def _tol(dtype):
    return TOL64 if dtype == torch.float64 else TOL32


# ------------------------------------------------------------------------------------ K1
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("D", [1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 16, 17, 31, 32, 33, 52, 64])
def test_spd_inv_logdet_vs_oracle(D, dtype):
    from pyvbmp_amd import ops
    for B in (1, 7, 131):
        A = _spd(B, D, dtype, seed=1000 + D + B)
        Ainv, ld = ops.spd_inv_logdet(A.to(DEV))
        ref_inv = torch.linalg.inv(A.double())
        ref_ld = torch.logdet(A.double())
        assert_close(Ainv, ref_inv, _tol(dtype), what=f"inverse D={D} B={B}")
        assert_close(ld, ref_ld, _tol(dtype), what=f"logdet D={D} B={B}")


@pytest.mark.parametrize("D", [33, 47, 64])
def test_spd_inv_large_dim_both_forms(D):
    """D > 32 runs one block per matrix for small batches and one wave per matrix for large ones; the debug switch forces either.
    Both against LU, with a batch larger than the grid (grid-stride loop) and a non-SPD member."""
    import ctypes
    from pyvbmp_amd import ops, _lib
    lib = _lib.load()
    lib.vbmp_debug_set_flags.argtypes = [ctypes.c_int]
    lib.vbmp_debug_set_flags.restype = None
    B = 2500
    A = _spd(B, D, torch.float64, seed=77 + D)
    A[5] = -A[5]  # det sign (-1)^D
    ref_inv, ref_ld = torch.linalg.inv(A), torch.logdet(A)
    try:
        for flag in (0x80, 0x40):
            lib.vbmp_debug_set_flags(flag)
            cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
            Ainv, ld = ops.spd_inv_logdet(A.to(DEV), nonspd=cnt)
            assert_close(Ainv, ref_inv, TOL64, what=f"inverse flag={flag}")
            assert_close(ld, ref_ld, TOL64, what=f"logdet flag={flag}")
            assert int(cnt.item()) == 1
    finally:
        lib.vbmp_debug_set_flags(0)


def test_spd_inv_batch_shapes_and_empty():
    from pyvbmp_amd import ops
    A = _spd(30, 6, torch.float64, 5).reshape(2, 3, 5, 6, 6)
    Ainv, ld = ops.spd_inv_logdet(A.to(DEV))
    assert Ainv.shape == (2, 3, 5, 6, 6) and ld.shape == (2, 3, 5)
    assert_close(Ainv, torch.linalg.inv(A))
    E, l0 = ops.spd_inv_logdet(torch.empty(0, 4, 4, dtype=torch.float64, device=DEV))
    assert E.shape == (0, 4, 4) and l0.shape == (0,)


def test_spd_inv_non_spd_flags():
    """det < 0 -> logdet NaN like Tensor.logdet; counter incremented; inverse still algebraic."""
    from pyvbmp_amd import ops
    A = _spd(5, 4, torch.float64, 9)
    A[2] = torch.diag(torch.tensor([2.0, -1.0, 3.0, 1.5], dtype=torch.float64))
    cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
    Ainv, ld = ops.spd_inv_logdet(A.to(DEV), nonspd=cnt)
    torch.cuda.synchronize()
    assert int(cnt.item()) == 1
    ref = torch.logdet(A)
    assert torch.isnan(ref[2]) and torch.isnan(ld.cpu()[2])
    assert_close(ld, ref)
    assert_close(Ainv, torch.linalg.inv(A))


def test_cpu_tensor_is_refused():
    from pyvbmp_amd import _lib, ops
    with pytest.raises(_lib.VbmpHipError):
        ops.spd_inv_logdet(torch.eye(3, dtype=torch.float64).unsqueeze(0))


# --------------------------------------------------------------------------- Wishart class
def _beta(c):
    b = float(c["beta"])
    return None if b < 0 else b


@pytest.mark.parametrize("D", [2, 6, 16])
@pytest.mark.parametrize("lr", [1.0, 0.5])
@pytest.mark.parametrize("beta", [None, 0.9])
def test_wishart_golden(golden, D, lr, beta):
    from pyvbmp_amd.dists import Wishart
    c = golden("wishart")[f"w_d{D}_lr{lr}_beta{beta}"]
    w = Wishart(event_shape=(D, D), batch_shape=(6,), scale=float(c["scale"]), device=DEV, dtype=torch.float64)
    assert_close(w.invU_0, c["invU_0"])
    assert_close(w.logdet_invU_0, c["logdet_invU_0"])
    assert_close(w.U, c["init_U"])
    for step in (1, 2):
        w.ss_update(c[f"SExx{step}"].to(DEV), c[f"N{step}"].to(DEV), lr=lr, beta=beta)
        for f in ("invU", "U", "nu", "logdet_invU"):
            assert_close(getattr(w, f), c[f"s{step}_{f}"], what=f"step{step} {f}")
    for f in ("mean", "meaninv", "ESigma", "EinvSigma", "invEinvSigma", "ElogdetinvSigma", "logdetEinvSigma",
              "KLqprior", "logZ"):
        assert_close(getattr(w, f)(), c[f], what=f)


def test_wishart_extra_event_dims_golden(golden):
    from pyvbmp_amd.dists import Wishart
    c = golden("wishart")["w_event322"]
    w = Wishart(event_shape=(3, 2, 2), batch_shape=(5, 6), scale=1.3, device=DEV, dtype=torch.float64)
    w.ss_update(c["SExx1"].to(DEV), c["N1"].to(DEV), lr=0.8)
    for f in ("invU", "U", "nu", "logdet_invU"):
        assert_close(getattr(w, f), c[f"s1_{f}"], what=f)
    for f in ("ESigma", "EinvSigma", "ElogdetinvSigma", "KLqprior", "logZ"):
        assert_close(getattr(w, f)(), c[f], what=f)
    w.to_event(1)
    assert_close(w.KLqprior(), c["KLqprior_to_event1"])


# ------------------------------------------------------------------------------- NIW class
def _mk_niw(c, event_shape, batch_shape, **kw):
    from pyvbmp_amd.dists import NormalInverseWishart
    q = NormalInverseWishart(event_shape=event_shape, batch_shape=batch_shape, device=DEV, dtype=torch.float64, **kw)
    q.mu = c["init_mu"].to(DEV)  # the reference draws the initial mean at random: replay the stored draw
    return q


def _check_niw(q, c, pre, tol=TOL64):
    assert_close(q.lambda_mu, c[pre + "lambda_mu"], tol, what=pre + "lambda_mu")
    assert_close(q.mu, c[pre + "mu"], tol, what=pre + "mu")
    for f in ("invU", "U", "nu", "logdet_invU"):
        assert_close(getattr(q.invU, f), c[pre + f], tol, what=pre + f)


def _check_niw_expect(q, c, tol=TOL64):
    for f in ("mean", "EX", "EXXT", "ESigma", "ElogdetinvSigma", "EinvSigmamu", "EinvSigma", "EinvUX", "EXTinvUX",
              "KLqprior"):
        assert_close(getattr(q, f)(), c[f], tol, what=f)


@pytest.mark.parametrize("lr", [1.0, 0.5])
def test_niw_d16_golden(golden, lr):
    c = golden("niw")[f"niw_d16_lr{lr}"]
    q = _mk_niw(c, (16,), (8,))
    for step in (1, 2):
        q.ss_update(c[f"SExx{step}"].to(DEV), c[f"SEx{step}"].to(DEV), c[f"N{step}"].to(DEV), lr=lr)
        _check_niw(q, c, f"s{step}_")
    _check_niw_expect(q, c)
    assert_close(q.Elog_like(c["X_bcast"].to(DEV)), c["Elog_like_bcast"], what="Elog_like bcast")
    assert_close(q.Elog_like(c["X_full"].to(DEV)), c["Elog_like_full"], what="Elog_like full")


def test_niw_forgetting_golden(golden):
    c = golden("niw")["niw_beta0.9"]
    q = _mk_niw(c, (6,), (8,), scale=0.5)
    for step in (1, 2, 3):
        q.ss_update(c[f"SExx{step}"].to(DEV), c[f"SEx{step}"].to(DEV), c[f"N{step}"].to(DEV), lr=0.7, beta=0.9)
        _check_niw(q, c, f"s{step}_")
    _check_niw_expect(q, c)


def test_niw_raw_update_golden(golden):
    c = golden("niw")["niw_raw"]
    q = _mk_niw(c, (5,), (4,))
    q.raw_update(c["X"].to(DEV), c["p"].to(DEV), lr=1.0)
    _check_niw(q, c, "p_")
    q.raw_update(c["X"].to(DEV), c["p"].to(DEV), lr=0.3)
    _check_niw(q, c, "p2_")
    q.raw_update(c["X_full"].to(DEV), None, lr=1.0)
    _check_niw(q, c, "nop_")
    _check_niw_expect(q, c)


def test_niw_event32_batch56_golden(golden):
    c = golden("niw")["niw_e32_b56"]
    q = _mk_niw(c, (3, 2), (5, 6), scale=0.8)
    q.ss_update(c["SExx1"].to(DEV), c["SEx1"].to(DEV), c["N1"].to(DEV), lr=1.0)
    _check_niw(q, c, "s1_")
    assert_close(q.Elog_like(c["X"].to(DEV)), c["Elog_like"], what="Elog_like")
    q.raw_update(c["X"].to(DEV), c["p"].to(DEV), lr=0.6)
    _check_niw(q, c, "raw_")
    _check_niw_expect(q, c)


def test_niw_to_event_golden(golden):
    c = golden("niw")["niw_toevent"]
    q = _mk_niw(c, (2,), (5, 6))
    q.ss_update(c["SExx1"].to(DEV), c["SEx1"].to(DEV), c["N1"].to(DEV))
    q.to_event(1)
    assert_close(q.Elog_like(c["X"].to(DEV)), c["Elog_like"], what="Elog_like")
    assert_close(q.KLqprior(), c["KLqprior"], what="KL")


def test_niw_fixed_precision_and_prior_golden(golden):
    c = golden("niw")["niw_fixed_precision"]
    q = _mk_niw(c, (4,), (3,), fixed_precision=True)
    q.ss_update(c["SExx1"].to(DEV), c["SEx1"].to(DEV), c["N1"].to(DEV), lr=0.9)
    _check_niw(q, c, "s1_")
    c = golden("niw")["niw_prior"]
    prior = {"lambda_mu": c["prior_lambda_mu"], "mu": c["prior_mu"], "nu": c["prior_nu"].to(DEV),
             "invU": c["prior_invU"].to(DEV)}
    q = _mk_niw(c, (4,), (3,), prior_parms=prior)
    q.ss_update(c["SExx1"].to(DEV), c["SEx1"].to(DEV), c["N1"].to(DEV))
    _check_niw(q, c, "s1_")
    _check_niw_expect(q, c)


# ------------------------------------------------------------- K2 vs oracle on seeded inputs
def _niw_inputs(B, D, dtype, seed, n=32):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(B, D, n, generator=g, dtype=torch.float64)
    return (A @ A.transpose(-2, -1)).to(dtype), A.sum(-1).to(dtype), torch.full((B,), float(n), dtype=dtype)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("D", [2, 6, 16, 20, 40])
@pytest.mark.parametrize("lr", [1.0, 0.5])
def test_niw_ss_update_vs_oracle(D, dtype, lr):
    from oracle import niw as oniw
    from pyvbmp_amd.dists import NormalInverseWishart
    B = 1003 if D <= 16 else 67
    SExx, SEx, N = _niw_inputs(B, D, dtype, 77 + D)
    q = NormalInverseWishart((D,), (B,), device=DEV, dtype=dtype)
    st = oniw.niw_new((D,), (B,), mu_init=q.mu.cpu().double())
    for it in range(2):
        q.ss_update(SExx.to(DEV), SEx.to(DEV), N.to(DEV), lr=lr, beta=None)
        st = oniw.niw_ss_update(st, SExx.double(), SEx.double(), N.double(), lr=lr, beta=None)
    tol = _tol(dtype)
    assert_close(q.lambda_mu, st["lambda_mu"], tol, what="lambda")
    assert_close(q.mu, st["mu"], tol, what="mu")
    assert_close(q.invU.invU, st["W"]["invU"], tol, what="invU")
    assert_close(q.invU.nu, st["W"]["nu"], tol, what="nu")
    assert_close(q.invU.U, st["W"]["U"], tol, what="U")
    assert_close(q.invU.logdet_invU, st["W"]["logdet_invU"], tol, what="logdet")


def test_niw_full_size_properties():
    """BASELINE config 2 at full size (B=1e6, D=16, fp64): size-independent properties of the result
    (U is the inverse of invU, symmetric; natural parameters consistent with the inputs) plus an oracle
    comparison on a strided sample of the batch."""
    from oracle import niw as oniw
    from pyvbmp_amd.dists import NormalInverseWishart
    B, D, n = 1_000_000, 16, 32
    g = torch.Generator(device=DEV).manual_seed(0)
    SExx = torch.empty(B, D, D, dtype=torch.float64, device=DEV)
    SEx = torch.empty(B, D, dtype=torch.float64, device=DEV)
    for s in range(0, B, 100_000):
        A = torch.randn(100_000, D, n, generator=g, dtype=torch.float64, device=DEV)
        SExx[s:s + 100_000] = A @ A.transpose(-2, -1)
        SEx[s:s + 100_000] = A.sum(-1)
    N = torch.full((B,), float(n), dtype=torch.float64, device=DEV)
    q = NormalInverseWishart((D,), (B,), device=DEV, dtype=torch.float64)
    q.ss_update(SExx, SEx, N, lr=1.0, beta=None)
    W = q.invU
    eye = torch.eye(D, dtype=torch.float64, device=DEV)
    for s in range(0, B, 250_000):
        sl = slice(s, s + 250_000)
        resid = (W.U[sl] @ W.invU[sl] - eye).abs().amax()
        assert float(resid) < 1e-10, f"U @ invU != I: {float(resid):.3e}"
        assert float((W.U[sl] - W.U[sl].transpose(-2, -1)).abs().amax()) < 1e-12
    assert float((q.lambda_mu - (1.0 + n)).abs().amax()) == 0.0
    assert float((W.nu - (D + 2.0 + n)).abs().amax()) == 0.0
    assert_close(q.mu, SEx / (1.0 + n), 1e-14, what="mu = SEx/(lam0+N)")
    idx = torch.arange(0, B, 997, device=DEV)
    st = oniw.niw_new((D,), (len(idx),), mu_init=torch.zeros(len(idx), D, dtype=torch.float64))
    st = oniw.niw_ss_update(st, SExx[idx].cpu(), SEx[idx].cpu(), N[idx].cpu(), lr=1.0, beta=None)
    assert_close(W.invU[idx], st["W"]["invU"], what="invU sample")
    assert_close(W.U[idx], st["W"]["U"], what="U sample")
    assert_close(W.logdet_invU[idx], st["W"]["logdet_invU"], what="logdet sample")


def test_raw_update_dense_batch_beyond_one_launch():
    """NormalInverseWishart.raw_update with a DENSE sample per batch element over a batch longer than one launch's
    component axis (65535): ops.weighted_moments slices the axis; against the closed-form moments"""
    from pyvbmp_amd import ops
    S, B, D = 3, 70001, 4
    g = torch.Generator(device=DEV).manual_seed(2)
    X = torch.randn(S, B, D, generator=g, dtype=torch.float64, device=DEV)
    p = torch.rand(S, B, generator=g, dtype=torch.float64, device=DEV)
    Nk, SEx, SExx = ops.weighted_moments(X, p, 1, (B,))
    assert_close(Nk, p.sum(0), 1e-13, what="N")
    assert_close(SEx, (p.unsqueeze(-1) * X).sum(0), 1e-13, what="SEx")
    assert_close(SExx, torch.einsum("sb,sbi,sbj->bij", p, X, X), 1e-13, what="SExx")


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_spd_inv_logdet_beyond_kernel_size_takes_the_device_library(dtype):
    from pyvbmp_amd import ops
    g = torch.Generator().manual_seed(80)
    A = torch.randn(3, 80, 90, generator=g, dtype=torch.float64)
    P = (A @ A.transpose(-2, -1) / 90 + 0.5 * torch.eye(80, dtype=torch.float64))
    Ai, ld = ops.spd_inv_logdet(P.to(DEV, dtype))
    assert Ai.is_cuda and Ai.dtype == dtype
    tol = 1e-10 if dtype == torch.float64 else 1e-4
    assert_close(Ai, torch.linalg.inv(P), tol, what="inverse")
    assert_close(ld, torch.logdet(P), tol, what="logdet")


def test_debug_check_spd_raises_on_an_indefinite_matrix():
    """pyvbmp_amd.debug.check_spd(): the kernels' non-SPD counter is read back; without it the reference's silent NaN"""
    from pyvbmp_amd import _lib, debug, ops
    A = torch.eye(5, dtype=torch.float64, device=DEV).repeat(7, 1, 1)
    A[3, 2, 2] = -1.0
    Ai, ld = ops.spd_inv_logdet(A)            # default: silent, like Tensor.logdet
    assert torch.isnan(ld[3]) and torch.isfinite(ld[[0, 1, 2, 4, 5, 6]]).all()
    with debug.check_spd():
        ops.spd_inv_logdet(A[:3])             # all positive definite: passes
        with pytest.raises(_lib.VbmpHipError, match="1 matrices"):
            ops.spd_inv_logdet(A)
    assert ops.CHECK_SPD is False
    # beyond the kernels' matrix size (D > 64: the device library's factorisation) the counter / the debug check cover the call too
    B = torch.eye(70, dtype=torch.float64, device=DEV).repeat(4, 1, 1)
    B[1, 5, 5] = -2.0
    cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.spd_inv_logdet(B, nonspd=cnt)
    assert int(cnt) == 1
    with debug.check_spd():
        ops.spd_inv_logdet(B[2:])
        with pytest.raises(_lib.VbmpHipError, match="1 matrices"):
            ops.spd_inv_logdet(B)
