"""GPU parity of MatrixNormalWishart (updates, likelihoods, forward / backward messages) against golden
fixtures captured from the reference."""
import pytest
import torch

from tests.helpers import TOL64, assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"

MNW_CASES = ["mnw_4x3_b5", "mnw_4x3_b5_pad", "mnw_4x3_nobatch", "mnw_32x32", "mnw_6x7_b2_pad"]


def _make(c):
    from pyvbmp_amd.transforms import MatrixNormalWishart
    batch = tuple(int(v) for v in c["batch_shape"])
    mask = c["mask"].to(DEV) if "mask" in c else None
    X_mask = c["X_mask"].to(DEV) if "X_mask" in c else None
    m = MatrixNormalWishart((int(c["n"]), int(c["p"])), batch, pad_X=bool(int(c["pad_X"])), mask=mask, X_mask=X_mask,
                            device=DEV, dtype=torch.float64)
    m.mu = c["init_mu"].to(DEV)
    return m, batch


def _check(m, c, pre, tol=TOL64):
    for f in ("mu", "invV", "V", "logdetinvV"):
        assert_close(getattr(m, f), c[pre + f], tol, what=pre + f)
    for f in ("invU", "U", "nu", "logdet_invU"):
        assert_close(getattr(m.invU, f), c[pre + "invU_" + f], tol, what=pre + "invU_" + f)


def _vf(**kw):
    from pyvbmp_amd.dists import MultivariateNormal_vector_format
    return MultivariateNormal_vector_format(**{k: v.to(DEV).clone() for k, v in kw.items()})


@pytest.mark.parametrize("case", MNW_CASES)
def test_mnw_golden(golden, case):
    from pyvbmp_amd.dists import Delta
    c = golden("mnw")[case]
    m, batch = _make(c)
    X, Y = c["X"].to(DEV), c["Y"].to(DEV)
    pr = c["p_resp"].to(DEV) if "p_resp" in c else None
    N = X.shape[0]
    Xe = X.expand((N,) + batch + tuple(X.shape[-2:]))
    m.raw_update(Xe, Y, p=pr, lr=1.0)
    _check(m, c, "raw1_")
    m.raw_update(Xe, Y, p=pr, lr=0.5)
    _check(m, c, "raw2_")
    for f in ("EinvUX", "EXTinvU", "EXTinvUX", "EXinvVXT", "EXmMUTinvUXmMU", "EXmMUinvVXmMUT", "ElogdetinvU",
              "logdetEinvSigma", "ElogdetinvSigma", "EinvSigma", "invEinvSigma", "ESigma", "KLqprior", "mean",
              "weights", "var"):
        assert_close(getattr(m, f)(), c["raw2_" + f], what=f)
    if "raw2_EXTX" in c:
        assert_close(m.EXTX(), c["raw2_EXTX"])
        assert_close(m.EXXT(), c["raw2_EXXT"])
        A = torch.eye(m.p, dtype=torch.float64, device=DEV) * 0.5 + 0.1
        An = torch.eye(m.n, dtype=torch.float64, device=DEV) * 0.5 + 0.1
        assert_close(m.EXTAX(An), c["raw2_EXTAX"])
        assert_close(m.EXAXT(A), c["raw2_EXAXT"])
    assert_close(m.Elog_like(X, Y), c["Elog_like"], what="Elog_like")
    P, eta, R = m.Elog_like_X(Y)
    assert_close(P, c["ELX_invSigma"])
    assert_close(eta, c["ELX_invSigmamu"])
    assert_close(R, c["ELX_Res"], what="ELX Res")
    pY, R = m.predict(X)
    assert_close(pY.invSigma, c["predict_invSigma"])
    assert_close(pY.invSigmamu, c["predict_invSigmamu"])
    assert_close(R, c["predict_Res"], what="predict Res")
    pX, R = m.postdict(Y)
    assert_close(pX.invSigma, c["postdict_invSigma"])
    assert_close(pX.invSigmamu, c["postdict_invSigmamu"])
    assert_close(R, c["postdict_Res"], what="postdict Res")

    # messages: per-sample precision (BASELINE config 3 shape) and shared precision
    pYm, R = m.forward(_vf(invSigma=c["fw_in_invSigma"], invSigmamu=c["fw_in_invSigmamu"]))
    assert_close(pYm.mu, c["fw_mu"], what="fw mu")
    assert_close(pYm.Sigma, c["fw_Sigma"], what="fw Sigma")
    assert_close(R, c["fw_Res"], what="fw Res")
    pYs, R = m.forward(_vf(invSigma=c["fws_in_invSigma"], invSigmamu=c["fw_in_invSigmamu"]))
    assert_close(pYs.mu, c["fws_mu"], what="fws mu")
    assert_close(pYs.Sigma, c["fws_Sigma"], what="fws Sigma")
    assert_close(R, c["fws_Res"], what="fws Res")
    pXb, R = m.backward(_vf(invSigma=c["bw_in_invSigma"], invSigmamu=c["bw_in_invSigmamu"]))
    assert_close(pXb.invSigma, c["bw_invSigma"], what="bw P")
    assert_close(pXb.invSigmamu, c["bw_invSigmamu"], what="bw eta")
    assert_close(R, c["bw_Res"], what="bw Res")
    pXs, R = m.backward(_vf(invSigma=c["bws_in_invSigma"], invSigmamu=c["bw_in_invSigmamu"]), Res=0.25)
    assert_close(pXs.invSigma, c["bws_invSigma"], what="bws P")
    assert_close(pXs.invSigmamu, c["bws_invSigmamu"], what="bws eta")
    assert_close(R, c["bws_Res"], what="bws Res")
    px, R = m.Elog_like_X_given_pY(_vf(invSigma=c["bw_in_invSigma"], invSigmamu=c["bw_in_invSigmamu"]))
    assert_close(px.invSigma, c["ELXpY_invSigma"])
    assert_close(px.invSigmamu, c["ELXpY_invSigmamu"])
    assert_close(px.mu, c["ELXpY_mu"])
    assert_close(px.Sigma, c["ELXpY_Sigma"])
    assert_close(R, c["ELXpY_Res"], what="ELXpY Res")

    # update from distributions
    pxd = c["upd_x_mu"].shape[-2]
    ux_mu = c["upd_x_mu"].to(DEV).expand((N,) + batch + (pxd, 1)).clone()
    ux_S = c["upd_x_Sigma"].to(DEV).expand((N,) + batch + (pxd, pxd)).clone()
    from pyvbmp_amd.dists import MultivariateNormal_vector_format as VF
    pXu = VF(mu=ux_mu, Sigma=ux_S)
    assert_close(m.Elog_like_given_pX_pY(pXu, Delta(Y)), c["ELpXpY"], what="ELpXpY")
    m.update(pXu, Delta(Y), p=pr, lr=0.8)
    _check(m, c, "upd_")
    n = int(c["n"])
    pYu = VF(mu=Y.clone(), Sigma=c["upd_y_Sigma"].to(DEV).expand((N,) + batch + (n, n)).clone())
    m.update(pXu, pYu, p=pr, lr=1.0, beta=0.5)
    _check(m, c, "upd2_")
    m.update(pXu, pYu, p=pr, lr=1.0, beta=0.5)
    _check(m, c, "upd3_")
    assert_close(m.KLqprior(), c["KLqprior_end"], what="KL end")


@pytest.mark.parametrize("case", ["mnw_Xmask", "mnw_mask", "mnw_mask_pad"])
def test_mnw_masks_golden(golden, case):
    c = golden("mnw")[case]
    m, batch = _make(c)
    assert_close(m.mu_0, c["init_mu_0"])
    X, Y, pr = c["X"].to(DEV), c["Y"].to(DEV), c["p_resp"].to(DEV)
    Xe = X.expand((X.shape[0],) + batch + tuple(X.shape[-2:]))
    m.raw_update(Xe, Y, p=pr, lr=1.0)
    _check(m, c, "raw1_")
    m.raw_update(Xe, Y, p=pr, lr=0.5)
    _check(m, c, "raw2_")
    assert_close(m.KLqprior(), c["KLqprior"], what="KL")
    assert_close(m.Elog_like(X, Y), c["Elog_like"], what="Elog_like")


@pytest.mark.parametrize("keep,batch", [(0.2, ()), (0.8, ()), (0.3, (3,)), (0.7, (2,))])
def test_mnw_masked_mean_primal_and_dual(keep, batch):
    """constrained posterior mean for sparse masks (fewer free than zeroed entries: the primal system on the free
    entries is solved) and dense ones (the reference's dual system on the zeroed entries), against the oracle, which
    always takes the reference's route (LU solve of the dual system)"""
    from oracle import mnw as omnw
    from pyvbmp_amd.transforms import MatrixNormalWishart
    n, p, N = 7, 9, 200
    g = torch.Generator().manual_seed(int(keep * 100) + len(batch))
    mask = torch.rand(n, p, generator=g) < keep
    mask[torch.arange(n), torch.arange(n)] = True  # every row keeps an entry
    m = MatrixNormalWishart((n, p), batch, mask=mask.to(DEV), device=DEV, dtype=torch.float64)
    (zi, _), (fi, _) = m._mask_entries()
    assert (fi.numel() <= zi.numel()) == (keep < 0.5)  # the case really takes the branch it is meant for
    st = omnw.mnw_new((n, p), batch, mu_init=m.mu.cpu(), mask=mask)
    for lr in (1.0, 0.6):
        X = torch.randn((N,) + batch + (p, 1), generator=g, dtype=torch.float64)
        Wt = torch.randn(n, p, generator=g, dtype=torch.float64) * mask
        # noise large enough that the reference's noise update (SEyy - mu invV mu', not exact for a constrained mean)
        # stays positive definite -- otherwise the second round would solve an indefinite system on both sides
        Y = Wt @ X + 2.0 * torch.randn((N,) + batch + (n, 1), generator=g, dtype=torch.float64)
        SExx = (X @ X.transpose(-2, -1)).sum(0)
        SEyx = (Y @ X.transpose(-2, -1)).sum(0)
        SEyy = (Y @ Y.transpose(-2, -1)).sum(0)
        Nn = torch.full(batch, float(N), dtype=torch.float64)
        assert torch.linalg.eigvalsh(omnw._noise_expect(st["W"])["EinvSigma"]).min() > 0
        m.ss_update(SExx.to(DEV), SEyx.to(DEV), SEyy.to(DEV), Nn.to(DEV), lr=lr)
        st = omnw.mnw_ss_update(st, SExx, SEyx, SEyy, Nn, lr=lr)
        assert_close(m.mu, st["mu"], 1e-10, what=f"mu lr={lr}")
        assert_close(m.invV, st["invV"], 1e-10, what="invV")
        assert_close(m.invU.invU, st["W"]["invU"], 1e-10, what="invU")
        assert (m.mu[..., ~mask.to(DEV)] == 0).all()


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("S,k,n,bias", [(100003, 6, 6, False), (70000, 7, 5, True), (1000, 1, 3, True), (513, 33, 64, False),
                                        (300, 64, 2, True), (0, 4, 4, False)])
def test_rows_affine(S, k, n, bias, dtype):
    """K12 against the plain product (fp32 reference of the same op for fp32, as for every floating-point kernel)"""
    from pyvbmp_amd import ops
    g = torch.Generator().manual_seed(S + k)
    X = torch.randn(S, k, generator=g, dtype=torch.float64)
    M = torch.randn(n, k, generator=g, dtype=torch.float64)
    c = torch.randn(n, generator=g, dtype=torch.float64) if bias else None
    out = ops.rows_affine(X.to(dtype).to(DEV), M.to(dtype).to(DEV), None if c is None else c.to(dtype).to(DEV))
    ref = X.to(dtype).double() @ M.to(dtype).double().T + (0 if c is None else c.to(dtype).double())
    assert out.shape == (S, n)
    if S:
        assert_close(out, ref, 1e-12 if dtype == torch.float64 else 1e-5, what="rows_affine")


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("S,k,n,opt", [(1, 1, 1, True), (1000, 3, 5, True), (70001, 6, 6, True), (5000, 9, 20, False), (4097, 16, 3, True),
                                       (300, 6, 64, False)])
def test_rows_affine_quad(S, k, n, opt, dtype):
    """K12 with the likelihood's scalar in the same pass: M x + c and -1/2 x'Px + b'x + c0 per row against plain products;
    opt=False: without the optional bias / linear term / constant"""
    from pyvbmp_amd import ops
    g = torch.Generator().manual_seed(S + 7 * k)
    X = torch.randn(S, k, generator=g, dtype=torch.float64)
    M = torch.randn(n, k, generator=g, dtype=torch.float64)
    A = torch.randn(k, k + 2, generator=g, dtype=torch.float64)
    P = A @ A.T / k
    c = torch.randn(n, generator=g, dtype=torch.float64) if opt else None
    b = torch.randn(k, generator=g, dtype=torch.float64) if opt else None
    c0 = torch.randn((), generator=g, dtype=torch.float64) if opt else None
    dev = lambda t: None if t is None else t.to(dtype).to(DEV)  # noqa: E731
    out, q = ops.rows_affine_quad(dev(X), dev(M), dev(c), dev(P), dev(b), dev(c0))
    Xd, Md, Pd = X.to(dtype).double(), M.to(dtype).double(), P.to(dtype).double()
    ref = Xd @ Md.T + (0 if c is None else c.to(dtype).double())
    qref = -0.5 * ((Xd @ Pd) * Xd).sum(-1) + (0 if b is None else Xd @ b.to(dtype).double()) + (0 if c0 is None else c0.to(dtype).double())
    assert out.shape == (S, n) and q.shape == (S,)
    tol = 1e-12 if dtype == torch.float64 else 1e-5
    assert_close(out, ref, tol, what="M x + c")
    assert_close(q, qref, tol, what="quadratic form")


def test_shared_matvec_routes_many_rows_through_k12():
    from pyvbmp_amd._common import shared_matvec
    g = torch.Generator().manual_seed(3)
    G = torch.randn(5, 6, generator=g, dtype=torch.float64).to(DEV)
    Y = torch.randn(200, 400, 6, 1, generator=g, dtype=torch.float64).to(DEV)
    assert_close(shared_matvec(G, Y), G @ Y, 1e-12, what="shared_matvec")


def test_rows_matmul_routes_many_rows_through_k12():
    from pyvbmp_amd._common import rows_matmul
    g = torch.Generator().manual_seed(5)
    X = torch.randn(300, 400, 7, generator=g, dtype=torch.float64).to(DEV)
    W = torch.randn(7, 8, generator=g, dtype=torch.float64).to(DEV)
    assert_close(rows_matmul(X, W.transpose(0, 1).contiguous().transpose(0, 1)), X @ W, 1e-12, what="rows_matmul")
    assert_close(rows_matmul(X[:3], W), X[:3] @ W, 1e-12, what="rows_matmul small")


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("S,inner,NB", [(24000, (53, 53), 25), (25600, (7, 7), 4), (5000, (8, 8), 1), (70, (4, 4), 32), (333, (3,), 5),
                                        (100, (6, 6), 33),
                                        # the matrix-core form (> 8 columns, even element count): the flocking DMBD's 52 x 52 by 25 roles,
                                        # ragged sample / element / column counts, one and two row tiles, a single k step
                                        (24000, (52, 52), 25), (1003, (10, 10), 9), (64, (4, 4), 16), (67, (6, 6), 17), (4099, (32, 32), 32),
                                        (130, (2,), 12), (3001, (9, 14), 31)])
def test_weighted_matsum_cols(S, inner, NB, dtype):
    """K5b with a weight column per expert against W^T @ C (odd and even matrix sizes, up to and beyond the column limit)"""
    from pyvbmp_amd import ops
    g = torch.Generator().manual_seed(S + NB)
    C = torch.randn((S,) + inner, generator=g, dtype=torch.float64)
    W = torch.rand(S, NB, generator=g, dtype=torch.float64)
    out = ops.weighted_matsum_cols(C.to(dtype).to(DEV), W.to(dtype).to(DEV))
    ref = (W.to(dtype).double().T @ C.to(dtype).double().reshape(S, -1)).reshape((NB,) + inner)
    assert out.shape == (NB,) + inner
    assert_close(out, ref, 1e-12 if dtype == torch.float64 else 2e-5, what="matsum_cols")


# ------------------------------------------------------------------ BASELINE configs[2]: one precision per message
MNWMSG_CASES = ["msg_32x32", "msg_32x31_pad", "msg_32x32_pad", "msg_16x16_b3", "msg_24x32", "msg_32x20", "msg_8x40"]


def _fitted(c, dtype):
    """the product class carrying the fitted state stored in the fixture (rounded to `dtype`)"""
    from pyvbmp_amd.transforms import MatrixNormalWishart
    batch = tuple(int(v) for v in c["batch_shape"])
    m = MatrixNormalWishart((int(c["n"]), int(c["p"])), batch, pad_X=bool(int(c["pad_X"])), device=DEV, dtype=dtype)
    for f in ("mu", "invV", "V", "logdetinvV"):
        setattr(m, f, c["state_" + f].to(DEV, dtype))
    for f in ("invU", "U", "nu", "logdet_invU"):
        setattr(m.invU, f, c["state_invU_" + f].to(DEV, dtype))
    return m


def _msg_tol(dtype):
    return TOL64 if dtype == torch.float64 else 1e-4  # north_star: 1e-10 fp64 / 1e-4 fp32


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("case", MNWMSG_CASES)
def test_mnw_messages_per_message_golden(golden, case, dtype):
    """forward / backward with a precision per message (ref transforms/MatrixNormalWishart.py:303-328, :352-375) against
    the reference's fp64 outputs, in fp64 at 1e-10 and in fp32 at 1e-4 (the fixture's inputs are float32-representable);
    msg_32x32 in fp32 is exactly the kernel instance BASELINE configs[2] dispatches to, msg_8x40 (p = 40) takes the
    composed route beyond the fused kernel's size"""
    from pyvbmp_amd import _lib
    c = golden("mnwmsg")[case]
    m = _fitted(c, dtype)
    tol = _msg_tol(dtype)
    launched = []
    _lib.launch_hooks = (lambda name: launched.append(name), lambda name: None)
    try:
        pY, R = m.forward(_vfd(dtype, invSigma=c["fw_in_invSigma"], invSigmamu=c["fw_in_invSigmamu"]))
        pX, Rb = m.backward(_vfd(dtype, invSigma=c["bw_in_invSigma"], invSigmamu=c["bw_in_invSigmamu"]),
                            Res=c["bw_in_Res"].to(DEV, dtype))
    finally:
        _lib.launch_hooks = None
    fused = launched.count("vbmp_mnw_message")
    assert fused == (0 if case == "msg_8x40" else 2), launched  # the route this case is meant to take
    assert pY.mu.dtype == dtype and pX.invSigma.dtype == dtype
    assert_close(pY.mu, c["fw_mu"], tol, what="fw mu")
    assert_close(pY.Sigma, c["fw_Sigma"], tol, what="fw Sigma")
    assert_close(R, c["fw_Res"], tol, what="fw Res")
    assert_close(pX.invSigma, c["bw_invSigma"], tol, what="bw P")
    assert_close(pX.invSigmamu, c["bw_invSigmamu"], tol, what="bw eta")
    assert_close(Rb, c["bw_Res"], tol, what="bw Res")


def _vfd(dtype, **kw):
    from pyvbmp_amd.dists import MultivariateNormal_vector_format
    return MultivariateNormal_vector_format(**{k: v.to(DEV, dtype).clone() for k, v in kw.items()})


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("layout", ["offset", "strided"])
def test_mnw_messages_generic_instance_golden(golden, layout, dtype, monkeypatch):
    """the same configs[2] golden through the GENERIC kernel instance of the C-ABI: a precision operand that is not
    16-byte aligned (offset) or not dense over the messages (strided) cannot take the exact-size tile loads.  The host
    classes always hand the kernel dense aligned copies, so the operand is passed through as it lies in memory here."""
    from pyvbmp_amd import ops
    c = golden("mnwmsg")["msg_32x32"]
    m = _fitted(c, dtype)
    tol = _msg_tol(dtype)
    raw = {}

    def relayout(P):
        P = P.to(DEV, dtype)
        N, d = P.shape[0], P.shape[-1]
        if layout == "offset":
            out = torch.empty(N * d * d + 1, device=DEV, dtype=dtype)[1:].view(N, d, d)
        else:
            out = torch.empty(N, 2, d, d, device=DEV, dtype=dtype)[:, 1]
        out.copy_(P)
        assert (out.data_ptr() % 16 != 0) if layout == "offset" else (out.stride(0) == 2 * d * d)
        raw[out.data_ptr()] = out
        return out
    norm2 = ops._norm2

    def passthrough(X, sample_shape, bshape, inner):
        if X.data_ptr() in raw and len(inner) == 2:
            return X, X.stride(0), 0  # as it lies: (pointer, element stride per message, shared by the experts)
        return norm2(X, sample_shape, bshape, inner)
    monkeypatch.setattr(ops, "_norm2", passthrough)
    from pyvbmp_amd.dists import MultivariateNormal_vector_format as VF
    pY, R = m.forward(VF(invSigma=relayout(c["fw_in_invSigma"]), invSigmamu=c["fw_in_invSigmamu"].to(DEV, dtype)))
    assert_close(pY.mu, c["fw_mu"], tol, what="fw mu")
    assert_close(pY.Sigma, c["fw_Sigma"], tol, what="fw Sigma")
    assert_close(R, c["fw_Res"], tol, what="fw Res")
    pX, Rb = m.backward(VF(invSigma=relayout(c["bw_in_invSigma"]), invSigmamu=c["bw_in_invSigmamu"].to(DEV, dtype)),
                        Res=c["bw_in_Res"].to(DEV, dtype))
    assert_close(pX.invSigma, c["bw_invSigma"], tol, what="bw P")
    assert_close(pX.invSigmamu, c["bw_invSigmamu"], tol, what="bw eta")
    assert_close(Rb, c["bw_Res"], tol, what="bw Res")


def test_mnw_messages_full_size_fp32_properties(golden):
    """BASELINE configs[2] at full size (262 144 messages, n = p = 32, fp32): size-independent properties of the
    messages (Sigma_yy symmetric positive definite and >= the noise floor invEinvSigma; backward precision symmetric PD;
    forward-then-backward consistency of the fused kernels with the composed K1 + GEMM route on a slice) plus the oracle
    (fp64, reference algorithm) on a strided sample of the messages, at the fp32 tolerance"""
    from oracle import mnw as omnw
    from tests.test_oracle_golden import mnwmsg_oracle_state
    from pyvbmp_amd.dists import MultivariateNormal_vector_format as VF
    c = golden("mnwmsg")["msg_32x32"]
    dtype = torch.float32
    m = _fitted(c, dtype)
    N, d = 262144, 32
    g = torch.Generator(device=DEV).manual_seed(0)
    P = torch.empty(N, d, d, device=DEV, dtype=dtype)
    for s in range(0, N, 32768):
        A = torch.randn(32768, d, d + 2, generator=g, device=DEV, dtype=dtype)
        P[s:s + 32768] = A @ A.transpose(-2, -1) / (d + 2)
    P.diagonal(dim1=-2, dim2=-1).add_(0.5)
    eta = torch.randn(N, d, 1, generator=g, device=DEV, dtype=dtype)
    pY, R = m.forward(VF(invSigma=P, invSigmamu=eta))
    S = pY.Sigma
    assert tuple(S.shape) == (N, d, d) and tuple(pY.mu.shape) == (N, d, 1) and tuple(R.shape) == (N,)
    assert torch.isfinite(S).all() and torch.isfinite(pY.mu).all() and torch.isfinite(R).all()
    assert float((S - S.transpose(-2, -1)).abs().amax() / S.abs().amax()) < 1e-5
    floor = m.invEinvSigma()
    for s in range(0, N, 65536):  # Sigma_yy - invEinvSigma = M S* M' is PSD; Sigma_yy itself PD
        ev = torch.linalg.eigvalsh((S[s:s + 65536] - floor).double())
        assert float(ev.min()) > -1e-5 * float(ev.max())
    pX, Rb = m.backward(VF(invSigma=P, invSigmamu=eta))
    Q = pX.invSigma
    assert torch.isfinite(Q).all() and torch.isfinite(pX.invSigmamu).all() and torch.isfinite(Rb).all()
    assert float((Q - Q.transpose(-2, -1)).abs().amax() / Q.abs().amax()) < 1e-5
    for s in range(0, N, 65536):
        assert float(torch.linalg.eigvalsh(Q[s:s + 65536].double()).min()) > 0
    # oracle on a strided sample
    idx = torch.arange(0, N, 4099, device=DEV)
    st = mnwmsg_oracle_state(c, torch.float64)
    for f in ("mu", "invV", "V", "logdetinvV"):  # the state the fp32 product actually holds
        st[f] = getattr(m, f).cpu().double()
    for f in ("invU", "U", "nu", "logdet_invU"):
        st["W"][f] = getattr(m.invU, f).cpu().double()
    mu_o, S_o, R_o = omnw.mnw_forward(st, P[idx].cpu().double(), eta[idx].cpu().double())
    assert_close(pY.mu[idx], mu_o, 1e-4, what="fw mu sample")
    assert_close(S[idx], S_o, 1e-4, what="fw Sigma sample")
    assert_close(R[idx], R_o, 1e-4, what="fw Res sample")
    P_o, e_o, Rb_o = omnw.mnw_backward(st, P[idx].cpu().double(), eta[idx].cpu().double())
    assert_close(Q[idx], P_o, 1e-4, what="bw P sample")
    assert_close(pX.invSigmamu[idx], e_o, 1e-4, what="bw eta sample")
    assert_close(Rb[idx], Rb_o, 1e-4, what="bw Res sample")


def _composed_expectations(m):
    """the five getters composed from the Wishart's own (ref transforms/MatrixNormalWishart.py:419-471)"""
    R = m.invU.EinvSigma()
    return R, R @ m.mu, m.n * m.V + m.mu.mT @ R @ m.mu, m.invU.ElogdetinvSigma()


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("event, batch, pad, masked", [((3, 2), (), False, False), ((6, 6), (), True, False),
                                                       ((4, 5), (7,), True, False), ((2, 3, 4), (5,), False, False),
                                                       ((6, 6), (3,), True, True), ((33, 40), (2,), False, False)])
def test_mnw_expectations_one_launch(event, batch, pad, masked, dtype):
    """K14: EinvSigma / EinvUX / EXTinvU / EXTinvUX / ElogdetinvSigma from one launch == the composed getters, and the cached
    values follow every way the parameters can change (rebinding by an update, in-place writes, a replayed graph)."""
    from pyvbmp_amd.transforms.MatrixNormalWishart import MatrixNormalWishart
    torch.manual_seed(5)
    tol = dict(tol=1e-12) if dtype == torch.float64 else dict(tol=2e-5)
    mask = None
    if masked:
        mask = torch.rand(event, device="cuda") > 0.3  # event-shaped, shared by the batch
    m = MatrixNormalWishart(event, batch, pad_X=pad, mask=mask, device="cuda", dtype=dtype)
    n, p = m.n, m.p
    px = p - 1 if pad else p
    lead = batch + tuple(event[:-2])

    def check():
        want = _composed_expectations(m)
        got = (m.EinvSigma(), m.EinvUX(), m.EXTinvUX(), m.ElogdetinvSigma())
        for g, w in zip(got, want):
            assert g.shape == w.shape
            assert_close(g, w, **tol)
        assert_close(m.EXTinvU(), want[1].mT, **tol)

    check()
    assert m._expectations() is not None and m.EinvSigma() is m.EinvSigma()  # served from the cache
    for _ in range(2):  # an update rebinds the parameters
        X = torch.randn((50,) + (1,) * len(lead) + (px, 1), device="cuda", dtype=dtype)
        Y = torch.randn((50,) + lead + (n, 1), device="cuda", dtype=dtype)
        m.raw_update(X, Y, lr=0.7)
        check()
    m.mu.mul_(1.5)  # an in-place write bumps the tensor's version
    check()
    m.invU.nu.add_(2.0)
    check()


def test_mnw_expectations_cache_follows_graph_replays():
    """a replayed HIP graph rewrites the parameters in place without touching any tensor version: the graph advances the state
    epoch, so getters read eagerly between replays are recomputed"""
    from pyvbmp_amd import graph
    from pyvbmp_amd.transforms.MatrixNormalWishart import MatrixNormalWishart
    torch.manual_seed(6)
    m = MatrixNormalWishart((4, 3), (), pad_X=True, device="cuda", dtype=torch.float64)
    ref = MatrixNormalWishart((4, 3), (), pad_X=True, device="cuda", dtype=torch.float64)
    for k in ("mu", "mu_0"):
        setattr(ref, k, getattr(m, k).clone())
    X = torch.randn(200, 3, 1, device="cuda", dtype=torch.float64)
    Y = torch.randn(200, 4, 1, device="cuda", dtype=torch.float64)
    seen = []
    g = graph.GraphedStep(m, lambda mm: mm.raw_update(X, Y, lr=0.5), warmup=1, post=lambda mm: seen.append(mm.EXTinvUX().clone()))
    g.run(3)
    for i in range(5):
        ref.raw_update(X, Y, lr=0.5)
        assert_close(seen[i], _composed_expectations(ref)[2], tol=1e-10)
    assert_close(m.EinvUX(), _composed_expectations(ref)[1], tol=1e-10)
    g.close()


def test_trace_term_on_the_unmasked_entries_only(monkeypatch):
    """Elog_like_given_pX_pY with an X_mask: the covariance trace runs over the entries that the mask leaves free in some batch
    element (E[X' invU X] is exactly zero elsewhere) == the trace over all entries"""
    from pyvbmp_amd.dists.MultivariateNormal_vector_format import MultivariateNormal_vector_format as MVN
    from pyvbmp_amd.transforms.MatrixNormalWishart import MatrixNormalWishart
    torch.manual_seed(11)
    dt, K, n, px = torch.float64, 7, 3, 12
    xm = torch.zeros(K, 1, px, dtype=torch.bool, device="cuda")
    for k in range(K):
        xm[k, 0, (k % 4) * 3:(k % 4) * 3 + 3] = True  # every expert reads three of the twelve inputs
    m = MatrixNormalWishart((n, px), (K,), X_mask=xm, pad_X=False, device="cuda", dtype=dt)
    X = torch.randn(300, 1, px, 1, device="cuda", dtype=dt)
    Y = torch.randn(300, K, n, 1, device="cuda", dtype=dt)
    m.raw_update(X, Y, lr=1.0)
    A = torch.randn(40, 1, px, px, device="cuda", dtype=dt)
    pX = MVN(mu=torch.randn(40, 1, px, 1, device="cuda", dtype=dt), Sigma=A @ A.mT + torch.eye(px, device="cuda", dtype=dt))
    B = torch.randn(40, 1, n, n, device="cuda", dtype=dt)
    pY = MVN(mu=torch.randn(40, 1, n, 1, device="cuda", dtype=dt), Sigma=B @ B.mT + torch.eye(n, device="cuda", dtype=dt))
    keep = m._xmask_union(px)
    assert keep is not None and keep.numel() == 4 * 9  # four distinct 3 x 3 blocks of the 144 entries
    got = m.Elog_like_given_pX_pY(pX, pY)
    monkeypatch.setattr(MatrixNormalWishart, "_xmask_union", lambda self, d: None)
    want = m.Elog_like_given_pX_pY(pX, pY)
    assert got.shape == want.shape == (40, K)
    assert_close(got, want, tol=1e-12)
