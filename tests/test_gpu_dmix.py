"""GPU parity of the Polya-Gamma logistic gate (MultiNomialLogisticRegression over MVN_ard) and of the gated mixture
of linear transforms (dMixtureofLinearTransforms) against fixtures captured from the reference."""
import pytest
import torch

from tests.helpers import assert_close
from tests.test_oracle_dmix import DMIX_CASES, MNLR_CASES

pytestmark = pytest.mark.gpu
DEV = "cuda"


def set_ard(q, c, pre):
    for f in ("mu", "invSigma", "invSigmamu", "Sigma", "logdetinvSigma"):
        setattr(q, f, c[pre + f].to(DEV))
    q.alpha.alpha = c[pre + "alpha"].to(DEV)
    q.alpha.beta = c[pre + "beta"].to(DEV)


def check_ard(q, c, pre, tol=1e-9):
    for f in ("mu", "invSigma", "invSigmamu", "Sigma", "logdetinvSigma"):
        assert_close(getattr(q, f), c[pre + f], tol, what=pre + f)
    assert_close(q.alpha.alpha, c[pre + "alpha"], tol, what=pre + "alpha")
    assert_close(q.alpha.beta, c[pre + "beta"], tol, what=pre + "beta")


@pytest.mark.parametrize("case", MNLR_CASES)
def test_logistic_gate_golden(golden, case):
    from pyvbmp_amd.dists import MultivariateNormal_vector_format as VF
    from pyvbmp_amd.transforms import MultiNomialLogisticRegression
    c = golden("dmix")[case]
    lr = float(c["lr"])
    m = MultiNomialLogisticRegression(int(c["ncls"]), int(c["p"]), pad_X=bool(int(c["pad_X"])), device=DEV,
                                      dtype=torch.float64)
    set_ard(m.beta, c, "init_")
    X, Y, w = c["X"].to(DEV), c["Y"].to(DEV), c["w"].to(DEV)
    m.raw_update(X, Y, iters=2, lr=lr)
    check_ard(m.beta, c, "r1_")
    m.raw_update(X, Y, iters=3, p=w, lr=lr)
    check_ard(m.beta, c, "r2_")
    assert_close(m.KLqprior(), c["KLqprior"], 1e-9, what="KL")
    assert_close(m.Elog_like(X, Y), c["Elog_like"], 1e-9, what="Elog_like")
    assert_close(m.log_predict(X[:9]), c["log_predict"], 1e-9, what="log_predict")
    assert_close(m.predict(X[:9]), c["predict"], 1e-9, what="predict")
    assert_close(m.log_predict_1(X[:9]), c["log_predict_1"], 1e-9, what="log_predict_1")
    assert_close(m.log_predict_2(X[:9]), c["log_predict_2"], 1e-9, what="log_predict_2")
    assert_close(m.weights(), c["weights"], 1e-9, what="weights")
    SigX = c["SigX"].to(DEV)
    pX = VF(mu=X.unsqueeze(-1), Sigma=SigX)
    assert_close(m.Elog_like_given_pX_pY(pX, Y), c["ELpXpY"], 1e-9, what="ELpXpY")
    assert_close(m.log_forward(VF(mu=X[:9].unsqueeze(-1), Sigma=SigX[:9])), c["log_forward"], 1e-9, what="log_forward")
    px, Res = m.backward(c["bw_pY"].to(DEV))
    assert_close(px.invSigma, c["bw_invSigma"], 1e-9, what="bw P")
    assert_close(px.invSigmamu, c["bw_invSigmamu"], 1e-9, what="bw eta")
    assert_close(px.mu, c["bw_mu"], 1e-9, what="bw mu")
    assert_close(Res, c["bw_Res"], 1e-9, what="bw Res")
    m.update(pX, Y, iters=2, lr=lr)
    check_ard(m.beta, c, "u1_")


@pytest.mark.parametrize("case", DMIX_CASES)
def test_gated_mixture_golden(golden, case):
    from pyvbmp_amd.dists import MultivariateNormal_vector_format as VF
    from pyvbmp_amd.transforms import dMixtureofLinearTransforms
    from tests.test_oracle_lds import n_iters
    c = golden("dmix")[case]
    n, p, K = int(c["n"]), int(c["p"]), int(c["mix"])
    gamma = bool(int(c["gamma"]))
    lr = float(c["lr"])
    m = dMixtureofLinearTransforms(n, p, K, pad_X=True, type='Gamma' if gamma else 'Wishart', device=DEV,
                                   dtype=torch.float64)
    m.A.mu = c["init_A_mu"].to(DEV)
    if gamma:
        m.A.invU.gamma.alpha = c["init_A_alpha"].to(DEV)
        m.A.invU.gamma.beta = c["init_A_beta"].to(DEV)
    set_ard(m.pi.beta, c, "init_pi_")
    X, Y = c["X"].to(DEV), c["Y"].to(DEV)
    for it in range(1, n_iters(c) + 1):
        pre = f"it{it}_"
        m.raw_update(X, Y, iters=1, lr=lr)
        assert_close(m.A.mu, c[pre + "A_mu"], 1e-9, what=pre + "A_mu")
        assert_close(m.A.invV, c[pre + "A_invV"], 1e-9, what=pre + "A_invV")
        check_ard(m.pi.beta, c, pre + "pi_")
    assert_close(m.KLqprior(), c["KLqprior"], 1e-9, what="KL")
    assert_close(m.Elog_like(X, Y), c["Elog_like"], 1e-9, what="Elog_like")
    pY, pr = m.predict(X[:7])
    assert_close(pr, c["pred_p"], 1e-9, what="pred p")
    assert_close(pY.mean(), c["pred_mu"], 1e-9, what="pred mu")
    assert_close(pY.ESigma(), c["pred_Sigma"], 1e-9, what="pred Sigma")
    SigX, SigY = c["SigX"].to(DEV), c["SigY"].to(DEV)
    pX, pYd = VF(mu=X.unsqueeze(-1), Sigma=SigX), VF(mu=Y.unsqueeze(-1), Sigma=SigY)
    assert_close(m.Elog_like_given_pX_pY(pX, pYd), c["ELpXpY"], 1e-9, what="ELpXpY")
    if "fw_mu" in c:
        fw = m.forward(VF(mu=X[:5].unsqueeze(-1), Sigma=SigX[:5]))
        assert_close(fw.mean(), c["fw_mu"], 1e-9, what="fw mu")
        assert_close(fw.ESigma(), c["fw_Sigma"], 1e-9, what="fw Sigma")
    px, logZ, pp = m.postdict(Y[:5])
    assert_close(pp, c["post_p"], 1e-9, what="post p")
    assert_close(logZ, c["post_logZ"], 1e-9, what="post logZ")
    assert_close(px.invSigma, c["post_invSigma"], 1e-9, what="post P")
    assert_close(px.invSigmamu, c["post_invSigmamu"], 1e-9, what="post eta")
    m.update(pX, pYd, iters=1, lr=lr)
    assert_close(m.logZ, c["upd_logZ"], 1e-9, what="upd logZ")
    assert_close(m.NA, c["upd_NA"], 1e-9, what="upd NA")
    assert_close(m.A.mu, c["upd_A_mu"], 1e-9, what="upd A_mu")
    check_ard(m.pi.beta, c, "upd_pi_")
    assert_close(m.ELBO_last, c["upd_ELBO"], 1e-9, what="upd ELBO")


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 1e-4)])
@pytest.mark.parametrize("n,p,K,N,pad", [(8, 8, 8, 20000, False), (3, 4, 5, 9001, True), (4, 4, 3, 4096, False)])
def test_mixture_of_linear_transforms_fused_estep(n, p, K, N, pad, dtype, tol):
    """MixtureofLinearTransforms.update_assignments on many dense samples takes ONE fused launch (symmetric-packed K3 on the stacked
    z = [x; y], with the per-sample evidence): responsibilities and evidence against the composed route (K3a + softmax in torch)"""
    from pyvbmp_amd import ops
    from pyvbmp_amd.transforms import MixtureofLinearTransforms
    g = torch.Generator().manual_seed(n * 100 + K)
    X = torch.randn(N, p, 1, generator=g, dtype=torch.float64)
    Y = torch.randn(n, p, generator=g, dtype=torch.float64) @ X + 0.3 * torch.randn(N, n, 1, generator=g, dtype=torch.float64)
    torch.manual_seed(1)
    m = MixtureofLinearTransforms(n, p, K, pad_X=pad, device=DEV, dtype=dtype)
    Xd, Yd = X.to(DEV, dtype), Y.to(DEV, dtype)
    m.raw_update(Xd[:500], Yd[:500], iters=2)   # move the experts off their prior
    launched = []
    from pyvbmp_amd import _lib
    _lib.launch_hooks = (lambda name: launched.append(name), lambda name: None)
    try:
        m.update_assignments(Xd, Yd)
    finally:
        _lib.launch_hooks = None
    fusable = (n + p + (0 if not pad else 0)) in (4, 8, 16) or ((n + p) == 32 and dtype == torch.float32)
    assert ("vbmp_mixture_estep" in launched) == fusable, launched
    p_f, lz_f = m.p.clone(), m.logZ.clone()
    try:
        ops._estep_sym_off = True
        m.update_assignments(Xd, Yd)
    finally:
        ops._estep_sym_off = False
    assert_close(p_f, m.p, tol, what="responsibilities")
    assert_close(lz_f, m.logZ, tol, what="per-sample evidence")
