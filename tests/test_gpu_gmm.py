"""GPU parity of the mixture path (K3 fused E-step, K3a quadratic log-likelihood, K4 weighted moments,
then K2) through Mixture / GaussianMixtureModel, against golden fixtures captured from the reference
(BASELINE configs[0]: GMM K=4, D=2 on two-moons data) and against the CPU oracle at larger sizes."""
import pytest
import torch

from tests.helpers import TOL64, assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_gmm_two_moons_golden(golden):
    from pyvbmp_amd.models import GaussianMixtureModel
    c = golden("gmm")["gmm_k4_d2"]
    g = GaussianMixtureModel(4, 2, device=DEV, dtype=torch.float64)
    g.dist.mu = c["init_mu"].to(DEV)
    g.pi.alpha = c["init_alpha"].to(DEV)
    assert_close(g.pi.alpha_0, c["alpha_0"])
    assert_close(g.dist.invU.invU_0, c["init_invU_0"])
    X = c["data"].to(DEV)
    for it in range(1, 21):
        g.update(X, iters=1, lr=1.0)
        if it in (1, 2, 5, 20):
            pre = f"it{it}_"
            tol = TOL64 if it <= 2 else 1e-8  # EM iterations amplify rounding differences
            assert_close(g.p, c[pre + "p"], tol, what=pre + "p")
            assert_close(g.NA, c[pre + "NA"], tol, what=pre + "NA")
            assert_close(g.logZ, c[pre + "logZ"], tol, what=pre + "logZ")
            assert_close(g.ELBO_last, c[pre + "ELBO"], tol, what=pre + "ELBO")
            assert_close(g.pi.alpha, c[pre + "alpha"], tol, what=pre + "alpha")
            assert_close(g.dist.mu, c[pre + "mu"], tol, what=pre + "mu")
            assert_close(g.dist.lambda_mu, c[pre + "lambda_mu"], tol, what=pre + "lambda")
            assert_close(g.dist.invU.invU, c[pre + "invU"], tol, what=pre + "invU")
            assert_close(g.dist.invU.U, c[pre + "U"], tol, what=pre + "U")
            assert_close(g.dist.invU.logdet_invU, c[pre + "logdet_invU"], tol, what=pre + "logdet")
    assert torch.equal(g.assignment().cpu(), c["final_assignment"])
    assert_close(g.KLqprior(), c["final_KLqprior"], 1e-8)


def test_mixture_batched_golden(golden):
    """Mixture with batch_shape (3,), 6 components, event (3,2): the generic (non-fused) E-step."""
    from pyvbmp_amd.dists import Mixture, NormalInverseWishart
    c = golden("gmm")["mixture_b3_k6_e32"]
    niw = NormalInverseWishart(event_shape=(3, 2), batch_shape=(3, 6), device=DEV, dtype=torch.float64)
    mix = Mixture(niw, event_shape=(6,))
    niw.mu = c["init_mu"].to(DEV)
    mix.pi.alpha = c["init_alpha"].to(DEV)
    X = c["X"].to(DEV)
    for it in (1, 2, 3):
        mix.update(X, iters=1, lr=0.9)
        pre = f"it{it}_"
        for f in ("p", "NA", "logZ"):
            assert_close(getattr(mix, f), c[pre + f], 1e-9, what=pre + f)
        assert_close(mix.ELBO_last, c[pre + "ELBO"], 1e-9)
        assert_close(mix.pi.alpha, c[pre + "alpha"], 1e-9)
        assert_close(niw.mu, c[pre + "mu"], 1e-9)
        assert_close(niw.invU.invU, c[pre + "invU"], 1e-9)
        assert_close(niw.invU.U, c[pre + "U"], 1e-9)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("K,D,N", [(4, 16, 20011), (7, 3, 5000), (16, 32, 3001), (3, 40, 777), (40, 4, 3000)])
def test_gmm_iteration_vs_oracle(K, D, N, dtype):
    """One full VB iteration (fused E-step K3, moments K4, update K2) against the oracle."""
    from oracle import mixture as omix
    from oracle import niw as oniw
    from pyvbmp_amd.models import GaussianMixtureModel
    g = torch.Generator().manual_seed(K * 100 + D)
    centers = 3.0 * torch.randn(K, D, generator=g, dtype=torch.float64)
    X = (centers[torch.randint(K, (N,), generator=g)] + torch.randn(N, D, generator=g, dtype=torch.float64)).to(dtype)
    m = GaussianMixtureModel(K, D, device=DEV, dtype=dtype)
    m.dist.mu = centers.to(dtype).to(DEV) + 0.1
    st = oniw.niw_new((D,), (K,), scale=1.0 / K ** (1.0 / D), mu_init=(centers.to(dtype) + 0.1).double())
    alpha_0 = m.pi.alpha_0.cpu().double()
    alpha = m.pi.alpha.cpu().double()
    tol = TOL64 if dtype == torch.float64 else 2e-4
    for it in range(2):
        m.update(X.to(DEV), iters=1, lr=1.0)
        st, alpha, out = omix.mixture_iteration(st, alpha_0, alpha, X.double(), 1.0, (K,), (), (D,))
        assert_close(m.p, out["p"], tol, what="p")
        assert_close(m.NA, out["NA"], tol, what="NA")
        assert_close(m.logZ, out["logZ"], tol, what="logZ")
        assert_close(m.dist.mu, st["mu"], tol, what="mu")
        assert_close(m.dist.invU.invU, st["W"]["invU"], tol, what="invU")
        assert_close(m.dist.invU.U, st["W"]["U"], tol * 10, what="U")
        assert_close(m.pi.alpha, alpha, tol, what="alpha")


def test_weighted_moments_linearity_large():
    """Size-independent property at scale: moments are linear in the weights and additive over
    sample chunks (N = 2e6 samples, K = 4, D = 16)."""
    from pyvbmp_amd import ops
    N, K, D = 2_000_000, 4, 16
    g = torch.Generator(device=DEV).manual_seed(3)
    X = torch.randn(N, 1, D, generator=g, dtype=torch.float64, device=DEV)
    p = torch.rand(N, K, generator=g, dtype=torch.float64, device=DEV)
    Nk, SEx, SExx = ops.weighted_moments(X, p, 1, (K,))
    h = N // 2
    N1, S1, Q1 = ops.weighted_moments(X[:h], p[:h], 1, (K,))
    N2, S2, Q2 = ops.weighted_moments(X[h:], 2.0 * p[h:], 1, (K,))
    assert_close(N1 + 0.5 * N2, Nk, 1e-11)
    assert_close(S1 + 0.5 * S2, SEx, 1e-9)  # sums of zero-mean terms: compare against the scale of SExx
    assert_close(Q1 + 0.5 * Q2, SExx, 1e-11)
    ref = torch.einsum("nk,ni,nj->kij", p[:100000], X[:100000, 0], X[:100000, 0])
    Q, = ops.weighted_moments(X[:100000], p[:100000], 1, (K,))[2:]
    assert_close(Q, ref, 1e-11)


def test_gmm_sample_sharded_matches_single(golden):
    """Mixture._update_sharded: two 'ranks' (threads sharing the GPU, a barrier-based stand-in for the
    all-reduce) each hold half of the samples; the result must equal the unsharded run."""
    import threading

    from pyvbmp_amd.models import GaussianMixtureModel
    K, D, N = 4, 16, 6000
    g = torch.Generator().manual_seed(5)
    centers = 3.0 * torch.randn(K, D, generator=g, dtype=torch.float64)
    X = (centers[torch.randint(K, (N,), generator=g)] + torch.randn(N, D, generator=g, dtype=torch.float64)).to(DEV)
    mu0 = (centers + 0.2).to(DEV)

    def make():
        m = GaussianMixtureModel(K, D, device=DEV, dtype=torch.float64)
        m.dist.mu = mu0.clone()
        m.pi.alpha = torch.full((K,), 0.7, dtype=torch.float64, device=DEV)
        return m

    ref = make()
    ref.update(X, iters=3, lr=1.0)

    class PairReducer:
        def __init__(self):
            self.bar = threading.Barrier(2)
            self.slots = [None, None]
            self.calls = 0

        def all_reduce(self, rank, tensors):
            self.slots[rank] = [t.clone() for t in tensors]
            self.bar.wait()
            out = [a + b for a, b in zip(*self.slots)]
            self.bar.wait()
            return out

    shared = PairReducer()

    class RankView:
        def __init__(self, rank):
            self.rank = rank
            self.calls = 0

        def all_reduce(self, tensors):
            self.calls += 1
            return shared.all_reduce(self.rank, tensors)

    models, errs = [make(), make()], []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            m = models[rank]
            m.reducer = RankView(rank)
            m.update(X[rank * N // 2:(rank + 1) * N // 2], iters=3, lr=1.0)
        except Exception as e:  # pragma: no cover
            errs.append(e)
            shared.bar.abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for m in models:
        assert m.reducer.calls == 3  # one collective per VB iteration
        assert_close(m.dist.mu, ref.dist.mu, 1e-10)
        assert_close(m.dist.invU.invU, ref.dist.invU.invU, 1e-10)
        assert_close(m.dist.invU.U, ref.dist.invU.U, 1e-10)
        assert_close(m.pi.alpha, ref.pi.alpha, 1e-10)
        assert_close(m.logZ, ref.logZ, 1e-10)


@pytest.mark.parametrize("K,D,N", [(1, 64, 50001), (3, 40, 20000), (4, 16, 8191 * 2), (2, 33, 4097), (9, 16, 8192), (4, 2, 70001), (3, 4, 5000)])
def test_weighted_moments_mfma_fp32(K, D, N):
    """fp32 K4 on the matrix cores (v_mfma_f32_32x32x2_f32 over the sample axis) against an fp64 einsum"""
    from pyvbmp_amd import ops
    g = torch.Generator().manual_seed(K + D)
    X = torch.randn(N, 1, D, generator=g, dtype=torch.float64)
    p = torch.rand(N, K, generator=g, dtype=torch.float64)
    Nk, SEx, SExx = ops.weighted_moments(X.float().to(DEV), p.float().to(DEV), 1, (K,))
    Xs = X[:, 0]
    assert_close(Nk, p.sum(0), 1e-5, what="N")
    assert_close(SExx, torch.einsum("nk,ni,nj->kij", p, Xs, Xs), 2e-5, what="SExx")
    ref_x = torch.einsum("nk,ni->ki", p, Xs)
    assert float((SEx.cpu().double() - ref_x).abs().max()) < 2e-5 * float(p.sum(0).max())
    # unit weights
    Nk, SEx, SExx = ops.weighted_moments(X.float().to(DEV).expand(N, 1, D), None, 1, (1,))
    assert_close(SExx[0], Xs.T @ Xs, 2e-5, what="SExx unit w")
    assert abs(float(Nk[0]) - N) < 1e-3 * N


@pytest.mark.parametrize("K,D,N", [(1, 32, 30001), (4, 16, 50000), (3, 2, 9000), (2, 21, 4099), (11, 16, 8191), (6, 24, 5000), (5, 1, 7000),
                                   (7, 3, 6001), (4, 4, 100003), (3, 40, 5000), (5, 57, 4100), (2, 64, 4096)])
def test_weighted_moments_mfma_fp64(K, D, N):
    """fp64 K4 on v_mfma_f64_16x16x4_f64 against an einsum, at the fp64 parity tolerance"""
    from pyvbmp_amd import ops
    g = torch.Generator().manual_seed(K * 3 + D)
    X = torch.randn(N, 1, D, generator=g, dtype=torch.float64) + 0.5
    p = torch.rand(N, K, generator=g, dtype=torch.float64)
    Nk, SEx, SExx = ops.weighted_moments(X.to(DEV), p.to(DEV), 1, (K,))
    Xs = X[:, 0]
    assert_close(Nk, p.sum(0), 1e-12, what="N")
    assert_close(SExx, torch.einsum("nk,ni,nj->kij", p, Xs, Xs), 1e-12, what="SExx")
    assert_close(SEx, torch.einsum("nk,ni->ki", p, Xs), 1e-12, what="SEx")


def _quad_ref(X, P, b, c):
    """X (S,Bi,D), P (Bo,Bi,D,D), b (Bo,Bi,D), c (Bo,Bi) -> (S,Bo,Bi) in fp64 on the CPU"""
    X, P, b, c = X.double(), P.double(), b.double(), c.double()
    q = torch.einsum("sid,oide,sie->soi", X, P, X)
    return -0.5 * q + torch.einsum("sid,oid->soi", X, b) + c


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("S,Bo,Bi,D", [(4099, 4, 1, 16), (3000, 3, 2, 9), (2500, 33, 1, 12), (2100, 2, 3, 40),
                                        (2304, 5, 1, 64), (2050, 1, 1, 57), (5000, 7, 1, 8)])
def test_quadform_mfma_and_valu_forms(S, Bo, Bi, D, dtype, smoother_flags):
    """K3a on the matrix cores (D >= 8, enough samples) and its VALU form (debug switch 0x100) against an einsum:
    padded feature blocks, component chunks (Bo > 32), inner component axes with their own X slice."""
    from pyvbmp_amd import ops
    g = torch.Generator().manual_seed(S + D)
    X = torch.randn(S, Bi, D, generator=g, dtype=torch.float64)
    A = torch.randn(Bo, Bi, D, D + 2, generator=g, dtype=torch.float64)
    P = A @ A.transpose(-2, -1) / D + torch.randn(Bo, Bi, D, D, generator=g, dtype=torch.float64) * 0.05  # not symmetric
    b = torch.randn(Bo, Bi, D, generator=g, dtype=torch.float64)
    c = torch.randn(Bo, Bi, generator=g, dtype=torch.float64)
    ref = _quad_ref(X, P, b, c)
    tol = 1e-12 if dtype == torch.float64 else 2e-5
    for flag in (0, 0x100):
        smoother_flags(flag)
        out = ops.quadform_loglike(X[:, None].to(DEV, dtype), P.to(DEV, dtype), b.to(DEV, dtype), c.to(DEV, dtype))
        assert out.shape == (S, Bo, Bi)
        assert_close(out, ref, tol, what=f"flag={flag}")


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("S,K,D", [(4099, 4, 16), (2500, 33, 12), (3001, 6, 40), (2048, 2, 64),
                                   (3000, 1, 4), (4099, 1, 16), (2049, 1, 2), (5000, 2, 4),  # K = 1: single-component mixture
                                   (4100, 16, 16), (5000, 20, 8), (6000, 7, 12), (3333, 32, 8), (2100, 5, 24),  # fused softmax: KG = 4, 8, 2, 8, 2
                                   (8192, 8, 8), (4100, 3, 32), (70001, 5, 16), (4097, 8, 16), (9000, 7, 4)])  # symmetric-packed form
def test_mixture_estep_mfma_and_valu_forms(S, K, D, dtype, smoother_flags):
    from pyvbmp_amd import ops
    g = torch.Generator().manual_seed(S + K)
    X = torch.randn(S, D, generator=g, dtype=torch.float64)
    A = torch.randn(K, D, D + 2, generator=g, dtype=torch.float64)
    P = A @ A.transpose(-2, -1) / D
    b = torch.randn(K, D, generator=g, dtype=torch.float64)
    c = torch.randn(K, generator=g, dtype=torch.float64)
    l = _quad_ref(X[:, None], P[:, None], b[:, None], c[:, None])[..., 0]
    lse = torch.logsumexp(l, -1)
    pref = torch.exp(l - lse[:, None])
    tol = 1e-11 if dtype == torch.float64 else 1e-4
    # the product's choice (symmetric-packed VALU form for few components, D in {4, 8, 16, 32}, >= 4096 samples) / fused MFMA
    # form where it applies / VALU form / MFMA + separate softmax pass
    try:
        for sym_off, flag in ((False, 0), (True, 0), (True, 0x100), (True, 0x4000)):
            ops._estep_sym_off = sym_off
            smoother_flags(flag)
            p, NA, logZ = ops.mixture_estep(X.to(DEV, dtype), P.to(DEV, dtype), b.to(DEV, dtype), c.to(DEV, dtype))
            assert_close(p, pref, tol, what=f"p flag={flag} sym_off={sym_off}")
            assert_close(NA, pref.sum(0), tol, what=f"NA flag={flag} sym_off={sym_off}")
            assert_close(logZ, lse.sum(), tol, what=f"logZ flag={flag} sym_off={sym_off}")
    finally:
        ops._estep_sym_off = False


def test_gmm_graphed_update_matches_eager():
    """hipGraph replay of the VB iteration (pyvbmp_amd.graph) against the eager loop: same state after the same number
    of iterations (order of the atomic sums differs: 1e-9), also after an eager update in between and a second call."""
    from pyvbmp_amd.models import GaussianMixtureModel
    g = torch.Generator().manual_seed(3)
    X = torch.cat((torch.randn(300, 2, generator=g, dtype=torch.float64) + 3.0,
                   torch.randn(300, 2, generator=g, dtype=torch.float64) - 2.0)).to(DEV)
    models = []
    for graphed in (False, True):
        torch.manual_seed(5)
        m = GaussianMixtureModel(4, 2, device=DEV, dtype=torch.float64)
        m.update(X, iters=7, lr=1.0, graphed=graphed)
        m.update(X, iters=1, lr=1.0)               # eager step in between rebinds the state
        m.update(X, iters=4, lr=1.0, graphed=graphed)  # cached graph, state synced back in
        models.append(m)
    a, b = models
    for f in ("mu", "lambda_mu"):
        assert_close(getattr(b.dist, f), getattr(a.dist, f), 1e-9, what=f)
    assert_close(b.dist.invU.invU, a.dist.invU.invU, 1e-9, what="invU")
    assert_close(b.pi.alpha, a.pi.alpha, 1e-9, what="alpha")
    assert_close(b.p, a.p, 1e-9, what="p")
    assert_close(b.ELBO(), a.ELBO(), 1e-9, what="ELBO")


def test_graph_lifetime_is_deterministic_and_survives_a_delete_during_capture():
    """(1) no reference cycle: dropping the last reference to a graphed model releases its GraphedStep at once (no
    collector run); (2) a graphed model whose last reference disappears WHILE another model's iteration is being captured
    must not take its HIP graph down mid-capture (that aborts the process): the graph is parked until the capture ends"""
    import gc
    import weakref

    from pyvbmp_amd import graph
    from pyvbmp_amd.models import GaussianMixtureModel
    g = torch.Generator().manual_seed(4)
    X = torch.randn(500, 2, generator=g, dtype=torch.float64).to(DEV)
    gc.collect()
    gc.disable()
    try:
        a = GaussianMixtureModel(3, 2, device=DEV, dtype=torch.float64)
        a.update(X, iters=4, lr=1.0, graphed=True)
        step_ref = weakref.ref(next(iter(a.__dict__["_vbmp_graphs"].values())))
        model_ref = weakref.ref(a)
        del a
        assert model_ref() is None and step_ref() is None, "a graphed model must not sit in a reference cycle"

        victim = [GaussianMixtureModel(3, 2, device=DEV, dtype=torch.float64)]
        victim[0].update(X, iters=4, lr=1.0, graphed=True)
        vref = weakref.ref(victim[0])
        calls = []

        def step(m):
            calls.append(graph._capture_depth)
            if graph._capture_depth > 0 and victim:
                victim.pop()  # last reference to a graphed model goes away in the middle of this capture
                assert vref() is None and len(graph._graveyard) == 1
            m.update(X, iters=1, lr=1.0)
        torch.manual_seed(8)
        b = GaussianMixtureModel(3, 2, device=DEV, dtype=torch.float64)
        torch.manual_seed(8)
        ref = GaussianMixtureModel(3, 2, device=DEV, dtype=torch.float64)
        graph.run_iterations(b, step, 5, key="k")
        assert calls == [0, 0, 1] and not victim and not graph._graveyard
        ref.update(X, iters=5, lr=1.0)
        assert_close(b.dist.mu, ref.dist.mu, 1e-9, what="mu after the interrupted capture")
        graph.close(b)
        assert not b.__dict__["_vbmp_graphs"]
    finally:
        gc.enable()


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 2e-5)])
@pytest.mark.parametrize("K,D", [(1, 2), (4, 16), (7, 5), (70, 40), (3, 64)])
def test_niw_estep_params_kernel_against_the_getters(K, D, dtype, tol):
    """K13 (one launch: P = U nu, b = P mu, c with the digamma sums and E log pi) against the composition of the classes' getters
    (ref dists/NormalInverseWishart.py:91-97,107-132, dists/Wishart.py:82-83, dists/Dirichlet.py:52-53) after a real update"""
    from pyvbmp_amd import ops
    from pyvbmp_amd.dists import NormalInverseWishart
    g = torch.Generator().manual_seed(K * 100 + D)
    q = NormalInverseWishart((D,), (K,), device=DEV, dtype=dtype)
    A = torch.randn(K, D, D + 3, generator=g, dtype=torch.float64)
    SExx, SEx = (A @ A.transpose(-2, -1)).to(dtype).to(DEV), A.sum(-1).to(dtype).to(DEV)
    N = (D + 3.0 + torch.arange(K, dtype=torch.float64)).to(dtype).to(DEV)
    q.ss_update(SExx, SEx, N, lr=1.0, beta=None)
    alpha = (0.3 + torch.rand(K, generator=g, dtype=torch.float64) * 5).to(dtype).to(DEV)
    W = q.invU
    P0, b0 = W.EinvSigma(), q.EinvSigmamu()
    c0 = -0.5 * q.EXTinvUX() + 0.5 * W.ElogdetinvSigma() - 0.5 * D * 1.8378770664093453
    for al in (None, alpha):
        P, b, c = ops.niw_estep_params(W.U, W.nu, q.mu, q.lambda_mu, W.logdet_invU, al)
        cref = c0 if al is None else c0 + torch.digamma(al) - torch.digamma(al.sum())
        assert_close(P, P0, tol, what="P")
        assert_close(b, b0, tol, what="b")
        assert_close(c, cref, tol, what="c")
    # small arguments of the digamma recurrence (nu close to D - 1, tiny alpha)
    W2nu = torch.full((K,), D - 1 + 1e-3, dtype=dtype, device=DEV)
    al2 = torch.full((K,), 1e-3, dtype=dtype, device=DEV)
    _, _, c = ops.niw_estep_params(W.U, W2nu, q.mu, q.lambda_mu, W.logdet_invU, al2)
    ar = torch.arange(D, dtype=dtype, device=DEV)
    P2 = W.U * W2nu.reshape(K, 1, 1)
    quad = ((P2 @ q.mu.unsqueeze(-1)).squeeze(-1) * q.mu).sum(-1)
    cref = -0.5 * (quad + D / q.lambda_mu.reshape(K)) + 0.5 * (D * 0.6931471805599453 - W.logdet_invU.reshape(K)
           + torch.digamma(0.5 * W2nu.unsqueeze(-1) - 0.5 * ar).sum(-1)) - 0.5 * D * 1.8378770664093453 \
           + torch.digamma(al2) - torch.digamma(al2.sum())
    assert_close(c, cref, tol * 10, what="c at small arguments")
