"""GPU parity of MatrixNormalGamma and of LinearDynamicalSystems with the reference's default
(MatrixNormalGamma) transition, against golden fixtures captured from the reference."""
import pytest
import torch

from tests.helpers import assert_close
from tests.test_oracle_mng import LDSG_CASES, MNG_CASES

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _make(c):
    from pyvbmp_amd.transforms import MatrixNormalGamma
    batch = tuple(int(v) for v in c["batch_shape"])
    mask = c["mask"].to(DEV) if "mask" in c else None
    m = MatrixNormalGamma((int(c["n"]), int(c["p"])), batch, pad_X=bool(int(c["pad_X"])), mask=mask, device=DEV,
                          dtype=torch.float64)
    m.mu = c["init_mu"].to(DEV)
    m.invU.gamma.alpha = c["init_alpha"].to(DEV)
    m.invU.gamma.beta = c["init_beta"].to(DEV)
    return m, batch


def _check(m, c, pre, tol=1e-10):
    for f in ("mu", "invV", "V", "logdetinvV"):
        assert_close(getattr(m, f), c[pre + f], tol, what=pre + f)
    assert_close(m.invU.gamma.alpha, c[pre + "alpha"], tol)
    assert_close(m.invU.gamma.beta, c[pre + "beta"], tol)


@pytest.mark.parametrize("case", MNG_CASES)
def test_mng_golden(golden, case):
    from pyvbmp_amd.dists import Delta
    from pyvbmp_amd.dists import MultivariateNormal_vector_format as VF
    c = golden("mng")[case]
    m, batch = _make(c)
    X, Y = c["X"].to(DEV), c["Y"].to(DEV)
    pr = c["p_resp"].to(DEV) if "p_resp" in c else None
    N = X.shape[0]
    Xe = X.expand((N,) + batch + tuple(X.shape[-2:]))
    m.raw_update(Xe, Y, p=pr, lr=1.0)
    _check(m, c, "raw1_")
    m.raw_update(Xe, Y, p=pr, lr=0.5)
    _check(m, c, "raw2_")
    for f in ("EinvUX", "EXTinvU", "EXTinvUX", "EXinvVXT", "ElogdetinvU", "ElogdetinvSigma", "EinvSigma", "ESigma",
              "KLqprior", "mean", "weights", "var"):
        assert_close(getattr(m, f)(), c["raw2_" + f], what=f)
    assert_close(m.Elog_like(X, Y), c["Elog_like"], what="Elog_like")
    P, eta, R = m.Elog_like_X(Y)
    assert_close(P, c["ELX_invSigma"])
    assert_close(eta, c["ELX_invSigmamu"])
    assert_close(R, c["ELX_Res"])
    pY, R = m.predict(X)
    assert_close(pY.invSigma, c["predict_invSigma"])
    assert_close(pY.invSigmamu, c["predict_invSigmamu"])
    assert_close(R, c["predict_Res"], what="predict Res")
    pYm = m.forward(VF(invSigma=c["fw_in_invSigma"].to(DEV), invSigmamu=c["fw_in_invSigmamu"].to(DEV)))
    assert_close(pYm.invSigma, c["fw_invSigma"], what="fw P")
    assert_close(pYm.invSigmamu, c["fw_invSigmamu"], what="fw eta")
    pXb, R = m.backward(VF(invSigma=c["bw_in_invSigma"].to(DEV), invSigmamu=c["bw_in_invSigmamu"].to(DEV)))
    assert_close(pXb.invSigma, c["bw_invSigma"], what="bw P")
    assert_close(pXb.invSigmamu, c["bw_invSigmamu"], what="bw eta")
    assert_close(R, c["bw_Res"], what="bw Res")
    pxd = c["upd_x_mu"].shape[-2]
    pXu = VF(mu=c["upd_x_mu"].to(DEV).expand((N,) + batch + (pxd, 1)).clone(),
             Sigma=c["upd_x_Sigma"].to(DEV).expand((N,) + batch + (pxd, pxd)).clone())
    assert_close(m.Elog_like_given_pX_pY(pXu, Delta(Y)), c["ELpXpY"], what="ELpXpY")
    m.update(pXu, Delta(Y), p=pr, lr=0.8)
    _check(m, c, "upd_")
    assert_close(m.KLqprior(), c["KLqprior_end"], what="KL end")


@pytest.mark.parametrize("case", LDSG_CASES)
def test_lds_default_transition_golden(golden, case):
    from pyvbmp_amd.models import LinearDynamicalSystems
    from tests.test_oracle_lds import n_iters
    c = golden("lds_mng")[case]
    h = int(c["hidden"])
    obs_shape = tuple(int(v) for v in c["obs_shape"])
    batch = tuple(int(v) for v in c["batch_shape"])
    m = LinearDynamicalSystems(obs_shape, h, control_dim=int(c["control"]), regression_dim=int(c["regression"]),
                               batch_shape=batch, device=DEV, dtype=torch.float64)  # default latent_noise
    m.x0.mu = c["init_x0_mu"].to(DEV)
    m.A.mu = c["init_A_mu"].to(DEV)
    m.A.invU.gamma.alpha = c["init_A_alpha"].to(DEV)
    m.A.invU.gamma.beta = c["init_A_beta"].to(DEV)
    m.obs_model.mu = c["init_obs_mu"].to(DEV)
    m.set_latent_parms()
    m.expand_to_batch = len(batch) > 0
    lr = float(c["lr"])
    dev = lambda k: c[k].to(DEV) if k in c else None  # noqa: E731
    y, u, r = m.reshape_inputs(dev("y"), dev("u"), dev("r"))
    for it in range(1, n_iters(c) + 1):
        pre = f"it{it}_"
        m.update_latents(y, u, r)
        for f in ("mu", "Sigma", "invSigma", "invSigmamu"):
            assert_close(getattr(m.px, f), c[pre + "px_" + f], 1e-9, what=pre + f)
        for f in ("SE_x_x", "SE_x0_x0", "SE_x0", "SE_y_xr", "SE_y_y", "SE_xpu_xpu", "SE_x_xpu", "SE_xr_xr", "logZ"):
            assert_close(getattr(m, f), c[pre + f], 1e-9, what=pre + f)
        assert_close(m.ELBO(), c[pre + "ELBO"], 1e-9, what=pre + "ELBO")
        m.ss_update(p=None, lr=lr)
        m.obs_model.ss_update(m.SE_xr_xr, m.SE_y_xr, m.SE_y_y, m.T, lr)
        assert_close(m.A.mu, c[pre + "A_mu"], 1e-9)
        assert_close(m.A.invU.gamma.alpha, c[pre + "A_alpha"], 1e-9)
        assert_close(m.A.invU.gamma.beta, c[pre + "A_beta"], 1e-9)
        assert_close(m.obs_model.mu, c[pre + "obs_mu"], 1e-9)
    assert_close(m.KLqprior(), c["KLqprior"], 1e-9)


@pytest.mark.parametrize("case", ["mix3_h3_o5", "mix2_h2_o4_ctrl_reg"])
def test_mixture_of_lds_golden(golden, case):
    """MixtureofLinearDynamicalSystems on the device against fixtures captured from the reference"""
    from pyvbmp_amd.models import MixtureofLinearDynamicalSystems
    from tests.test_oracle_lds import n_iters
    c = golden("mixlds")[case]
    K, h = int(c["K"]), int(c["hidden"])
    obs_shape = tuple(int(v) for v in c["obs_shape"])
    m = MixtureofLinearDynamicalSystems(K, obs_shape, h, int(c["control"]), int(c["regression"]), device=DEV,
                                        dtype=torch.float64)
    m.lds.x0.mu = c["init_x0_mu"].to(DEV)
    m.lds.A.mu = c["init_A_mu"].to(DEV)
    m.lds.A.invU.gamma.alpha = c["init_A_alpha"].to(DEV)
    m.lds.A.invU.gamma.beta = c["init_A_beta"].to(DEV)
    m.lds.obs_model.mu = c["init_obs_mu"].to(DEV)
    m.lds.set_latent_parms()
    m.pi.alpha = c["init_pi_alpha"].to(DEV)  # the constructor draws it from the global RNG
    lr = float(c["lr"])
    dev = lambda k: c[k].to(DEV) if k in c else None  # noqa: E731
    for it in range(1, n_iters(c) + 1):
        pre = f"it{it}_"
        m.update(dev("y"), dev("u"), dev("r"), iters=1, lr=lr, verbose=False)
        assert_close(m.p, c[pre + "p"], 1e-9, what=pre + "p")
        assert_close(m.NA, c[pre + "NA"], 1e-9, what=pre + "NA")
        assert_close(m.logZ, c[pre + "logZ"], 1e-9, what=pre + "logZ")
        assert_close(m.pi.alpha, c[pre + "pi_alpha"], 1e-9, what=pre + "alpha")
        assert_close(m.lds.A.mu, c[pre + "A_mu"], 1e-9, what=pre + "A_mu")
        assert_close(m.lds.A.invU.gamma.alpha, c[pre + "A_alpha"], 1e-9)
        assert_close(m.lds.A.invU.gamma.beta, c[pre + "A_beta"], 1e-9)
        assert_close(m.lds.obs_model.mu, c[pre + "obs_mu"], 1e-9, what=pre + "obs_mu")
        assert_close(m.lds.x0.mu, c[pre + "x0_mu"], 1e-9, what=pre + "x0_mu")
    assert_close(m.KLqprior(), c["KLqprior"], 1e-9, what="KL")
    assert torch.equal(m.assignment().cpu(), c["assignment"])


@pytest.mark.parametrize("case", ["mixlt_w_n3_p4_k3", "mixlt_w_nopad_lr", "mixlt_g_n3_p2_k2"])
def test_mixture_of_linear_transforms_golden(golden, case):
    """MixtureofLinearTransforms on the device against fixtures captured from the reference:
    raw_update iterations, predict, responsibility-weighted expectations, update(pX, pY)"""
    from pyvbmp_amd.dists import MultivariateNormal_vector_format as VF
    from pyvbmp_amd.transforms import MixtureofLinearTransforms
    from tests.test_oracle_lds import n_iters
    c = golden("mixlt")[case]
    n, p, dim = int(c["n"]), int(c["p"]), int(c["dim"])
    m = MixtureofLinearTransforms(n, p, dim, pad_X=bool(int(c["pad_X"])), type='Gamma' if int(c["gamma"]) else 'Wishart',
                                  device=DEV, dtype=torch.float64)
    m.W.mu = c["init_W_mu"].to(DEV)
    m.pi.alpha = c["init_pi_alpha"].to(DEV)
    if int(c["gamma"]):
        m.W.invU.gamma.alpha = c["init_W_alpha"].to(DEV)
        m.W.invU.gamma.beta = c["init_W_beta"].to(DEV)
    lr = float(c["lr"])
    X, Y = c["X"].to(DEV), c["Y"].to(DEV)
    for it in range(1, n_iters(c) + 1):
        pre = f"it{it}_"
        m.raw_update(X, Y, iters=1, lr=lr)
        assert_close(m.p, c[pre + "p"], 1e-9, what=pre + "p")
        assert_close(m.logZ, c[pre + "logZ"], 1e-9, what=pre + "logZ")
        assert_close(m.ELBO_last, c[pre + "ELBO"], 1e-9, what=pre + "ELBO")
        assert_close(m.pi.alpha, c[pre + "pi_alpha"], 1e-9, what=pre + "alpha")
        assert_close(m.W.mu, c[pre + "W_mu"], 1e-9, what=pre + "W_mu")
        assert_close(m.W.invV, c[pre + "W_invV"], 1e-9, what=pre + "W_invV")
        if int(c["gamma"]):
            assert_close(m.W.invU.gamma.alpha, c[pre + "W_alpha"], 1e-9)
            assert_close(m.W.invU.gamma.beta, c[pre + "W_beta"], 1e-9)
        else:
            assert_close(m.W.invU.invU, c[pre + "W_invU"], 1e-9)
            assert_close(m.W.invU.nu, c[pre + "W_nu"], 1e-9)
    assert_close(m.KLqprior(), c["KLqprior"], 1e-9, what="KL")
    pY, pr = m.predict(X[:7])
    assert_close(pr, c["pred_p"], 1e-9, what="pred p")
    assert_close(pY.mean(), c["pred_mu"], 1e-9, what="pred mu")
    assert_close(pY.ESigma(), c["pred_Sigma"], 1e-9, what="pred Sigma")
    for f in ("EinvUX", "EXTinvUX", "EinvSigma", "ElogdetinvSigma"):
        assert_close(getattr(m, f)(), c[f], 1e-9, what=f)
    m.update(VF(mu=X, Sigma=c["upd_SigX"].to(DEV)), VF(mu=Y, Sigma=c["upd_SigY"].to(DEV)), iters=1, lr=lr)
    assert_close(m.p, c["upd_p"], 1e-9, what="upd p")
    assert_close(m.logZ, c["upd_logZ"], 1e-9, what="upd logZ")
    assert_close(m.ELBO_last, c["upd_ELBO"], 1e-9, what="upd ELBO")
    assert_close(m.W.mu, c["upd_W_mu"], 1e-9, what="upd W_mu")
    assert_close(m.pi.alpha, c["upd_pi_alpha"], 1e-9, what="upd alpha")


@pytest.mark.parametrize("keep,batch", [(0.25, ()), (0.75, (3,))])
def test_mng_masked_mean_rowwise_equals_the_dense_systems(keep, batch):
    """diagonal noise: the row-separable constrained mean (one batched K1 launch) against the dense primal / dual
    systems of the parent class on the same posterior"""
    import torch
    from pyvbmp_amd.transforms import MatrixNormalGamma, MatrixNormalWishart
    n, p = 9, 11
    g = torch.Generator().manual_seed(int(100 * keep))
    mask = torch.rand(n, p, generator=g) < keep
    mask[torch.arange(n), torch.arange(n)] = True
    m = MatrixNormalGamma((n, p), batch, mask=mask.to(DEV), device=DEV, dtype=torch.float64)
    m.invU.gamma.alpha = m.invU.gamma.alpha * (1.0 + torch.rand(m.invU.gamma.alpha.shape, generator=g, dtype=torch.float64).to(DEV))
    A = torch.randn(batch + (p, p + 3), generator=g, dtype=torch.float64).to(DEV)
    invV = A @ A.transpose(-2, -1) + torch.eye(p, dtype=torch.float64, device=DEV)
    mu = torch.randn(batch + (n, p), generator=g, dtype=torch.float64).to(DEV)
    V = torch.linalg.inv(invV)
    got = m._constrain_mean(mu, invV, V)
    ref = MatrixNormalWishart._constrain_mean(m, mu, invV, V)
    assert_close(got, ref, 1e-11, what="constrained mean")
    assert (got[..., ~mask.to(DEV)] == 0).all()
