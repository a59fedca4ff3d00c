"""Pin the MatrixNormalGamma oracle (diagonal-noise sibling of MNW) and the LDS oracle with the reference's
DEFAULT transition to golden outputs of the reference.  CPU only."""
import pytest
import torch

from oracle import lds as olds
from oracle import mnw as omnw
from oracle import niw as oniw
from tests.helpers import assert_close

MNG_CASES = ["mng_4x3_b5", "mng_4x3_b5_pad", "mng_6x7_nobatch", "mng_mask"]
LDSG_CASES = ["ldsg_h6_o6", "ldsg_h3_o5_ctrl_reg", "ldsg_h4_o5_batch2"]


def mng_state(c):
    batch = tuple(int(v) for v in c["batch_shape"])
    st = omnw.mng_new((int(c["n"]), int(c["p"])), batch, mu_init=c["init_mu"], alpha_init=c["init_alpha"],
                      beta_init=c["init_beta"], pad_X=bool(int(c["pad_X"])), mask=c.get("mask"))
    return st, batch


def check_mng(st, c, pre, tol=1e-10):
    for f in ("mu", "invV", "V", "logdetinvV"):
        assert_close(st[f], c[pre + f], tol, what=pre + f)
    assert_close(st["W"]["alpha"], c[pre + "alpha"], tol)
    assert_close(st["W"]["beta"], c[pre + "beta"], tol)


@pytest.mark.parametrize("case", MNG_CASES)
def test_mng_oracle_golden(golden, case):
    c = golden("mng")[case]
    st, batch = mng_state(c)
    X, Y, pr = c["X"], c["Y"], c.get("p_resp")
    N = X.shape[0]
    Xe = X.expand((N,) + batch + X.shape[-2:])
    st = omnw.mnw_ss_update(st, *omnw.mnw_moments_data(st, Xe, Y, pr), lr=1.0)
    check_mng(st, c, "raw1_")
    st = omnw.mnw_ss_update(st, *omnw.mnw_moments_data(st, Xe, Y, pr), lr=0.5)
    check_mng(st, c, "raw2_")
    e = omnw.mnw_expectations(st)
    for f in ("EinvUX", "EXTinvU", "EXTinvUX", "EXinvVXT", "ElogdetinvU", "ElogdetinvSigma", "EinvSigma", "ESigma",
              "mean", "weights", "var"):
        assert_close(e[f], c["raw2_" + f], what=f)
    assert_close(omnw.mnw_kl(st), c["raw2_KLqprior"], what="KL")
    assert_close(omnw.mnw_elog_like(st, X, Y), c["Elog_like"])
    P, eta, R = omnw.mnw_elog_like_X(st, Y)
    assert_close(P, c["ELX_invSigma"])
    assert_close(eta, c["ELX_invSigmamu"])
    assert_close(R, c["ELX_Res"])
    Pyy, etay = omnw.mng_forward(st, c["fw_in_invSigma"], c["fw_in_invSigmamu"])
    assert_close(Pyy, c["fw_invSigma"], what="fw P")
    assert_close(etay, c["fw_invSigmamu"], what="fw eta")
    P, eta, R = omnw.mnw_backward(st, c["bw_in_invSigma"], c["bw_in_invSigmamu"])
    assert_close(P, c["bw_invSigma"])
    assert_close(eta, c["bw_invSigmamu"])
    assert_close(R, c["bw_Res"], what="bw Res")
    px = c["upd_x_mu"].shape[-2]
    EX = c["upd_x_mu"].expand((N,) + batch + (px, 1))
    EXXT = c["upd_x_Sigma"].expand((N,) + batch + (px, px)) + EX @ EX.transpose(-2, -1)
    EYYT = Y @ Y.transpose(-2, -1)
    assert_close(omnw.mnw_elog_like_dists(st, EX, EXXT, Y, EYYT), c["ELpXpY"])
    st = omnw.mnw_ss_update(st, *omnw.mnw_moments_dists(st, EX, EXXT, Y, EYYT, pr), lr=0.8)
    check_mng(st, c, "upd_")
    assert_close(omnw.mnw_kl(st), c["KLqprior_end"])


def ldsg_states(c):
    h = int(c["hidden"])
    obs_shape = tuple(int(v) for v in c["obs_shape"])
    batch = tuple(int(v) for v in c["batch_shape"])
    cd, rd = int(c["control"]) + 1, int(c["regression"]) + 1
    offset = (1,) * (len(obs_shape) - 1)
    x0 = oniw.niw_new(offset + (h,), batch, mu_init=c["init_x0_mu"])
    A = omnw.mng_new(offset + (h, h + cd), batch, mu_init=c["init_A_mu"], alpha_init=c["init_A_alpha"],
                     beta_init=c["init_A_beta"])
    obs = omnw.mnw_new(obs_shape + (h + rd,), batch, mu_init=c["init_obs_mu"])
    return x0, A, obs, h, obs_shape, batch, cd, rd


@pytest.mark.parametrize("case", LDSG_CASES)
def test_lds_default_transition_oracle_golden(golden, case):
    from tests.test_oracle_lds import n_iters
    c = golden("lds_mng")[case]
    x0, A, obs, h, obs_shape, batch, cd, rd = ldsg_states(c)
    nx = len(obs_shape) - 1
    lr = float(c["lr"])
    y, u, r = olds.reshape_inputs(c["y"], c.get("u"), c.get("r"), obs_shape, cd, rd, batch, len(batch) > 0)
    for it in range(1, n_iters(c) + 1):
        pre = f"it{it}_"
        sm = olds.smoother(olds.latent_parms(A, h), x0, h, y, u, r, obs, nx)
        for f in ("mu", "Sigma", "invSigma", "invSigmamu"):
            assert_close(sm[f], c[pre + "px_" + f], 1e-9, what=pre + f)
        st = olds.latent_stats(sm, y, u, r, obs_shape, cd, rd, batch, nx)
        for f in ("SE_x_x", "SE_x_xpu", "SE_xpu_xpu", "SE_xr_xr", "logZ"):
            assert_close(st[f], c[pre + f], 1e-9, what=pre + f)
        lz = st["logZ"]
        while lz.ndim > len(batch):
            lz = lz.sum(0)
        kl = oniw.niw_kl(x0) + omnw.mnw_kl(A)
        for _ in range(nx):
            kl = kl.squeeze(-1)
        assert_close(lz - (kl + omnw.mnw_kl(obs)), c[pre + "ELBO"], 1e-9, what=pre + "ELBO")
        st = olds.reduce_stats(st, len(batch), nx)
        x0 = oniw.niw_ss_update(x0, st["SE_x0_x0"], st["SE_x0"].squeeze(-1), st["N"], lr)
        A = omnw.mnw_ss_update(A, st["SE_xpu_xpu"], st["SE_x_xpu"], st["SE_x_x"], st["T"], lr)
        obs = omnw.mnw_ss_update(obs, st["SE_xr_xr"], st["SE_y_xr"], st["SE_y_y"], st["T"], lr)
        assert_close(A["mu"], c[pre + "A_mu"], 1e-9)
        assert_close(A["W"]["alpha"], c[pre + "A_alpha"], 1e-9)
        assert_close(A["W"]["beta"], c[pre + "A_beta"], 1e-9)
        assert_close(obs["mu"], c[pre + "obs_mu"], 1e-9)


MIXLDS_CASES = ["mix3_h3_o5", "mix2_h2_o4_ctrl_reg"]


@pytest.mark.parametrize("case", MIXLDS_CASES)
def test_mixture_of_lds_oracle_golden(golden, case):
    """MixtureofLinearDynamicalSystems (ref models/MixtureofLinearDynamicalSystems.py:12-34) restated on the
    functional LDS oracle: per-series evidences -> responsibilities -> weighted M-step, against reference fixtures."""
    from oracle import mixture as omix
    from tests.test_oracle_lds import n_iters
    c = golden("mixlds")[case]
    K = int(c["K"])
    c = dict(c)
    c["batch_shape"] = torch.tensor([K])
    x0, A, obs, h, obs_shape, batch, cd, rd = ldsg_states(c)
    nx = len(obs_shape) - 1
    lr = float(c["lr"])
    alpha_0 = torch.full((K,), 0.5, dtype=torch.float64)
    alpha = c["init_pi_alpha"]
    y, u, r = olds.reshape_inputs(c["y"], c.get("u"), c.get("r"), obs_shape, cd, rd, batch, True)
    for it in range(1, n_iters(c) + 1):
        pre = f"it{it}_"
        sm = olds.smoother(olds.latent_parms(A, h), x0, h, y, u, r, obs, nx)
        st = olds.latent_stats(sm, y, u, r, obs_shape, cd, rd, batch, nx)
        lz = st["logZ"]
        while lz.ndim > 2:
            lz = lz.sum(0)
        assert_close(lz, c[pre + "lds_logZ"], 1e-9, what=pre + "lds_logZ")
        log_p = lz + omix.dirichlet_loggeomean(alpha)
        logZ = torch.logsumexp(log_p, -1)
        p = torch.exp(log_p - logZ.unsqueeze(-1))
        assert_close(p, c[pre + "p"], 1e-9, what=pre + "p")
        assert_close(logZ, c[pre + "logZ"], 1e-9, what=pre + "logZ")
        NA = p.sum(0)
        assert_close(NA, c[pre + "NA"], 1e-9, what=pre + "NA")
        alpha = omix.dirichlet_ss_update(alpha_0, alpha, NA, lr)
        assert_close(alpha, c[pre + "pi_alpha"], 1e-9, what=pre + "alpha")
        st = olds.reduce_stats(st, len(batch), nx, p=p)
        x0 = oniw.niw_ss_update(x0, st["SE_x0_x0"], st["SE_x0"].squeeze(-1), st["N"], lr)
        A = omnw.mnw_ss_update(A, st["SE_xpu_xpu"], st["SE_x_xpu"], st["SE_x_x"], st["T"], lr)
        obs = omnw.mnw_ss_update(obs, st["SE_xr_xr"], st["SE_y_xr"], st["SE_y_y"], st["T"], lr)
        assert_close(A["mu"], c[pre + "A_mu"], 1e-9, what=pre + "A_mu")
        assert_close(A["W"]["alpha"], c[pre + "A_alpha"], 1e-9)
        assert_close(A["W"]["beta"], c[pre + "A_beta"], 1e-9)
        assert_close(obs["mu"], c[pre + "obs_mu"], 1e-9, what=pre + "obs_mu")
        assert_close(x0["mu"], c[pre + "x0_mu"], 1e-9, what=pre + "x0_mu")


MIXLT_CASES = ["mixlt_w_n3_p4_k3", "mixlt_w_nopad_lr", "mixlt_g_n3_p2_k2"]


def mixlt_state(c):
    n, p, dim = int(c["n"]), int(c["p"]), int(c["dim"])
    pad = bool(int(c["pad_X"]))
    scale = 1.0 / dim ** (1.0 / n)
    if int(c["gamma"]):
        st = omnw.mng_new((n, p), (dim,), mu_init=c["init_W_mu"], alpha_init=c["init_W_alpha"],
                          beta_init=c["init_W_beta"], scale=scale, pad_X=pad)
    else:
        st = omnw.mnw_new((n, p), (dim,), mu_init=c["init_W_mu"], scale=scale, pad_X=pad)
    return st, n, p, dim


@pytest.mark.parametrize("case", MIXLT_CASES)
def test_mixture_of_linear_transforms_oracle_golden(golden, case):
    """MixtureofLinearTransforms.raw_update / update / predict (ref transforms/MixtureofLinearTransforms.py:35-109)
    restated on the MNW oracle, against reference fixtures"""
    from oracle import mixture as omix
    from tests.test_oracle_lds import n_iters
    c = golden("mixlt")[case]
    st, n, p, dim = mixlt_state(c)
    lr = float(c["lr"])
    alpha_0 = torch.full((dim,), 0.5, dtype=torch.float64)
    alpha = c["init_pi_alpha"]
    X, Y = c["X"], c["Y"]
    Xe, Ye = X.unsqueeze(-3), Y.unsqueeze(-3)

    def estep(log_like):
        log_p = log_like + omix.dirichlet_loggeomean(alpha)
        logZ = torch.logsumexp(log_p, -1)
        return torch.exp(log_p - logZ.unsqueeze(-1)), logZ

    def elbo(logZ):
        return logZ.sum(0) - (omix.dirichlet_kl(alpha_0, alpha) + omnw.mnw_kl(st).sum(-1))
    for it in range(1, n_iters(c) + 1):
        pre = f"it{it}_"
        pr, logZ = estep(omnw.mnw_elog_like(st, Xe, Ye))
        assert_close(pr, c[pre + "p"], 1e-9, what=pre + "p")
        assert_close(logZ, c[pre + "logZ"], 1e-9, what=pre + "logZ")
        assert_close(elbo(logZ), c[pre + "ELBO"], 1e-9, what=pre + "ELBO")
        alpha = omix.dirichlet_ss_update(alpha_0, alpha, pr.sum(0), lr)
        N = X.shape[0]
        st = omnw.mnw_ss_update(st, *omnw.mnw_moments_data(st, Xe.expand(N, dim, p, 1), Ye, pr), lr=lr)
        assert_close(alpha, c[pre + "pi_alpha"], 1e-9, what=pre + "alpha")
        assert_close(st["mu"], c[pre + "W_mu"], 1e-9, what=pre + "W_mu")
        assert_close(st["invV"], c[pre + "W_invV"], 1e-9, what=pre + "W_invV")
    assert_close(omix.dirichlet_kl(alpha_0, alpha) + omnw.mnw_kl(st).sum(-1), c["KLqprior"], 1e-9, what="KL")
    # predict: moment-matched mixture of the experts' predictive Gaussians
    P, eta, res = omnw.mnw_predict(st, X[:7].unsqueeze(-3))
    if int(c["gamma"]):  # MatrixNormalGamma.predict hands back the un-normalised residual (ref MatrixNormalGamma.py:367-376)
        res = res + omnw._res_nat(P, eta)
    pg = torch.softmax(res + omix.dirichlet_loggeomean(alpha), -1)
    Sig = torch.linalg.inv(P).expand(7, dim, n, n)
    m = Sig @ eta
    pw = pg.unsqueeze(-1).unsqueeze(-1)
    mu = (m * pw).sum(-3)
    assert_close(pg, c["pred_p"], 1e-9, what="pred p")
    assert_close(mu, c["pred_mu"], 1e-9, what="pred mu")
    assert_close(((Sig + m @ m.transpose(-2, -1)) * pw).sum(-3) - mu @ mu.transpose(-2, -1), c["pred_Sigma"], 1e-9)
    # update(pX, pY): expected log-likelihood E-step, moments from distributions
    EX, EY = Xe, Ye
    EXXT = (c["upd_SigX"] + X @ X.transpose(-2, -1)).unsqueeze(-3)
    EYYT = (c["upd_SigY"] + Y @ Y.transpose(-2, -1)).unsqueeze(-3)
    pr, logZ = estep(omnw.mnw_elog_like_dists(st, EX, EXXT, EY, EYYT))
    assert_close(pr, c["upd_p"], 1e-9, what="upd p")
    assert_close(logZ, c["upd_logZ"], 1e-9, what="upd logZ")
    assert_close(elbo(logZ), c["upd_ELBO"], 1e-9, what="upd ELBO")
    alpha = omix.dirichlet_ss_update(alpha_0, alpha, pr.sum(0), lr)
    N = X.shape[0]
    st = omnw.mnw_ss_update(st, *omnw.mnw_moments_dists(st, EX.expand(N, dim, p, 1), EXXT.expand(N, dim, p, p), EY, EYYT,
                                                         pr), lr=lr)
    assert_close(st["mu"], c["upd_W_mu"], 1e-9, what="upd W_mu")
    assert_close(alpha, c["upd_pi_alpha"], 1e-9, what="upd alpha")
