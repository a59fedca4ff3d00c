"""Pin the CPU oracle to the reference: every oracle function is replayed on the inputs stored in
tests/golden/*.npz (captured from the real reference by tools/gen_golden.py) and must reproduce the
reference's outputs to 1e-10 normwise (fp64).  CPU only."""
import pytest
import torch

from oracle import mixture as omix
from oracle import mnw as omnw
from oracle import mutils as omu
from oracle import mvn as omvn
from oracle import niw as oniw
from tests.helpers import TOL64, assert_close


def _beta(c):
    b = float(c["beta"])
    return None if b < 0 else b


# ------------------------------------------------------------------ Wishart
@pytest.mark.parametrize("D", [2, 6, 16])
@pytest.mark.parametrize("lr", [1.0, 0.5])
@pytest.mark.parametrize("beta", [None, 0.9])
def test_wishart_ss_update(golden, D, lr, beta):
    c = golden("wishart")[f"w_d{D}_lr{lr}_beta{beta}"]
    st = oniw.wishart_new((D, D), (6,), float(c["scale"]))
    assert_close(st["invU_0"], c["invU_0"], what="invU_0")
    assert_close(st["U"], c["init_U"], what="init U")
    assert_close(st["logdet_invU"], c["init_logdet_invU"], what="init logdet")
    for step in (1, 2):
        st = oniw.wishart_ss_update(st, c[f"SExx{step}"], c[f"N{step}"], lr=lr, beta=beta)
        for f in ("invU", "U", "nu", "logdet_invU"):
            assert_close(st[f], c[f"s{step}_{f}"], what=f"step{step} {f}")
    e = oniw.wishart_expectations(st)
    for f in ("mean", "meaninv", "ESigma", "EinvSigma", "invEinvSigma", "ElogdetinvSigma", "logdetEinvSigma",
              "KLqprior", "logZ"):
        assert_close(e[f], c[f], what=f)


def test_wishart_extra_event_dims(golden):
    c = golden("wishart")["w_event322"]
    st = oniw.wishart_new((3, 2, 2), (5, 6), 1.3)
    st = oniw.wishart_ss_update(st, c["SExx1"], c["N1"], lr=0.8)
    for f in ("invU", "U", "nu", "logdet_invU"):
        assert_close(st[f], c[f"s1_{f}"], what=f)
    e = oniw.wishart_expectations(st)
    for f in ("ESigma", "EinvSigma", "ElogdetinvSigma", "KLqprior", "logZ"):
        assert_close(e[f], c[f], what=f)
    assert_close(oniw.wishart_kl(st, 4), c["KLqprior_to_event1"], what="KL to_event")


# ---------------------------------------------------------------------- NIW
def _check_niw(st, c, pre):
    assert_close(st["lambda_mu"], c[pre + "lambda_mu"], what=pre + "lambda_mu")
    assert_close(st["mu"], c[pre + "mu"], what=pre + "mu")
    for f in ("invU", "U", "nu", "logdet_invU"):
        assert_close(st["W"][f], c[pre + f], what=pre + f)


def _check_niw_expect(st, c, ed=None):
    e = oniw.niw_expectations(st)
    for f in ("mean", "EX", "EXXT", "ESigma", "ElogdetinvSigma", "EinvSigmamu", "EinvSigma", "EinvUX", "EXTinvUX"):
        assert_close(e[f], c[f], what=f)
    assert_close(oniw.niw_kl(st, ed), c["KLqprior"], what="KLqprior")


@pytest.mark.parametrize("lr", [1.0, 0.5])
def test_niw_d16(golden, lr):
    c = golden("niw")[f"niw_d16_lr{lr}"]
    st = oniw.niw_new((16,), (8,), mu_init=c["init_mu"])
    for step in (1, 2):
        st = oniw.niw_ss_update(st, c[f"SExx{step}"], c[f"SEx{step}"], c[f"N{step}"], lr=lr)
        _check_niw(st, c, f"s{step}_")
    _check_niw_expect(st, c)
    assert_close(oniw.niw_elog_like(st, c["X_bcast"]), c["Elog_like_bcast"], what="Elog_like bcast")
    assert_close(oniw.niw_elog_like(st, c["X_full"]), c["Elog_like_full"], what="Elog_like full")


def test_niw_forgetting(golden):
    c = golden("niw")["niw_beta0.9"]
    st = oniw.niw_new((6,), (8,), scale=0.5, mu_init=c["init_mu"])
    for step in (1, 2, 3):
        st = oniw.niw_ss_update(st, c[f"SExx{step}"], c[f"SEx{step}"], c[f"N{step}"], lr=0.7, beta=0.9)
        _check_niw(st, c, f"s{step}_")
    _check_niw_expect(st, c)


def test_niw_raw_update(golden):
    c = golden("niw")["niw_raw"]
    st = oniw.niw_new((5,), (4,), mu_init=c["init_mu"])
    st = oniw.niw_raw_update(st, c["X"], c["p"], (4,), (5,), lr=1.0)
    _check_niw(st, c, "p_")
    st = oniw.niw_raw_update(st, c["X"], c["p"], (4,), (5,), lr=0.3)
    _check_niw(st, c, "p2_")
    st = oniw.niw_raw_update(st, c["X_full"], None, (4,), (5,), lr=1.0)
    _check_niw(st, c, "nop_")
    _check_niw_expect(st, c)


def test_niw_event32_batch56(golden):
    c = golden("niw")["niw_e32_b56"]
    st = oniw.niw_new((3, 2), (5, 6), scale=0.8, mu_init=c["init_mu"])
    st = oniw.niw_ss_update(st, c["SExx1"], c["SEx1"], c["N1"], lr=1.0)
    _check_niw(st, c, "s1_")
    assert_close(oniw.niw_elog_like(st, c["X"]), c["Elog_like"], what="Elog_like")
    st = oniw.niw_raw_update(st, c["X"], c["p"], (5, 6), (3, 2), lr=0.6)
    _check_niw(st, c, "raw_")
    _check_niw_expect(st, c)


def test_niw_to_event(golden):
    c = golden("niw")["niw_toevent"]
    st = oniw.niw_new((2,), (5, 6), mu_init=c["init_mu"])
    st = oniw.niw_ss_update(st, c["SExx1"], c["SEx1"], c["N1"])
    assert_close(oniw.niw_elog_like(st, c["X"], event_dim=2), c["Elog_like"], what="Elog_like")
    assert_close(oniw.niw_kl(st, event_dim=2), c["KLqprior"], what="KL")


def test_niw_fixed_precision_and_prior(golden):
    c = golden("niw")["niw_fixed_precision"]
    st = oniw.niw_new((4,), (3,), mu_init=c["init_mu"], fixed_precision=True)
    st = oniw.niw_ss_update(st, c["SExx1"], c["SEx1"], c["N1"], lr=0.9)
    _check_niw(st, c, "s1_")
    c = golden("niw")["niw_prior"]
    st = oniw.niw_new((4,), (3,), mu_init=c["init_mu"], lambda_mu_0=c["prior_lambda_mu"], mu_0=c["prior_mu"],
                      nu_0=c["prior_nu"], invU_0=c["prior_invU"])
    st = oniw.niw_ss_update(st, c["SExx1"], c["SEx1"], c["N1"])
    _check_niw(st, c, "s1_")
    _check_niw_expect(st, c)


# ---------------------------------------------------------------------- MVN
@pytest.mark.parametrize("vf", [False, True])
def test_mvn_conversions(golden, vf):
    g = golden("mvn")
    c = g["vf_from_moments" if vf else "mvn_from_moments"]
    eta, P = omvn.natural_from_moments(c["mu"], c["Sigma"], vf)
    assert_close(P, c["EinvSigma"], what="EinvSigma")
    assert_close(eta, c["EinvSigmamu"], what="EinvSigmamu")
    ld = omvn.logdet_precision(Sigma=c["Sigma"])
    assert_close(ld, c["ElogdetinvSigma"], what="logdet")
    assert_close(omvn.second_moment(c["mu"], c["Sigma"], vf), c["EXXT"], what="EXXT")
    assert_close(omvn.trace_second_moment(c["mu"], c["Sigma"], vf), c["EXTX"], what="EXTX")
    if vf:
        assert_close(omvn.residual(c["mu"], eta, omvn.logdet_precision(invSigma=P), 5), c["Res"], what="Res")
    ld_used = omvn.logdet_precision(invSigma=P) if vf else ld
    assert_close(omvn.elog_like(c["X"], c["mu"], P, ld_used, vf), c["Elog_like"], what="Elog_like")
    c = g["vf_from_natural" if vf else "mvn_from_natural"]
    mu, S = omvn.moments_from_natural(c["invSigma"], c["invSigmamu"], vf)
    assert_close(mu, c["mean"], what="mean")
    assert_close(S, c["ESigma"], what="ESigma")
    assert_close(omvn.logdet_precision(invSigma=c["invSigma"]), c["ElogdetinvSigma"], what="logdet")
    assert_close(omvn.second_moment(mu, S, vf), c["EXXT"], what="EXXT")
    if vf:
        P2, e2 = c["invSigma"] + c["other_invSigma"], c["invSigmamu"] + c["other_invSigmamu"]
        assert_close(P2, c["nat_invSigma"])
        mu2, S2 = omvn.moments_from_natural(P2, e2, True)
        assert_close(mu2, c["nat_mean"], what="nat mean")
        assert_close(omvn.residual(mu2, e2, torch.logdet(P2), 5), c["nat_Res"], what="nat Res")
        P3 = P2 + c["other_invSigma"]
        assert_close(P3, c["comb_invSigma"])
        assert_close(torch.linalg.inv(P3), c["comb_ESigma"], what="comb ESigma")


@pytest.mark.parametrize("vf", [False, True])
def test_mvn_updates(golden, vf):
    c = golden("mvn")["vf_updates" if vf else "mvn_updates"]
    mu, S = omvn.raw_update(c["X"], c["p"], (3,), vf)
    assert_close(mu, c["p_mu"], what="p mu")
    assert_close(S, c["p_Sigma"], what="p Sigma")
    mu, S = omvn.raw_update(c["X_full"], None, (3,), vf)
    assert_close(mu, c["nop_mu"], what="nop mu")
    assert_close(S, c["nop_Sigma"], what="nop Sigma")
    if not vf:
        eta, P = omvn.natural_from_moments(mu, S, vf)
        assert_close(P, c["nop_EinvSigma"])
        assert_close(eta, c["nop_EinvSigmamu"])


# ------------------------------------------------------------- matrix_utils
@pytest.mark.parametrize("case", ["mu_4_3", "mu_6_6", "mu_16_8"])
def test_matrix_utils(golden, case):
    c = golden("matrix_utils")[case]
    A, B, C, D = c["A"], c["B"], c["C"], c["D"]
    assert_close(omu.block_diag(A, D), c["block_diag"])
    assert_close(omu.block_build(A, B, C, D), c["block_build"])
    for form in ("left", "right", "True"):
        for i, o in enumerate(omu.block_inverse(A, B, C, D, form)):
            assert_close(o, c[f"inv_{form}_{i}"], what=f"{form}[{i}]")
    assert_close(omu.block_inverse(A, B, C, D, False), c["inv_full"])
    assert_close(omu.block_inverse(A, B, C, D), c["inv_default"])
    for i, o in enumerate(omu.precision_marginalizer(A, B, C, D)):
        assert_close(o, c[f"marg_{i}"], what=f"marg[{i}]")
    assert_close(omu.block_logdet(A, B, C, D), c["logdet"])
    assert_close(omu.block_logdet(A, B, C, D, "A"), c["logdet_A"])
    assert_close(omu.block_logdet(A, B, C, D, "D"), c["logdet_D"])


# ---------------------------------------------------------------------- MNW
def _mnw_state(c, with_mu0=False):
    batch = tuple(int(v) for v in c["batch_shape"])
    st = omnw.mnw_new((int(c["n"]), int(c["p"])), batch, mu_init=c["init_mu"], pad_X=bool(int(c["pad_X"])),
                      mask=c.get("mask"), X_mask=c.get("X_mask"))
    return st, batch


def _check_mnw(st, c, pre, tol=TOL64):
    for f in ("mu", "invV", "V", "logdetinvV"):
        assert_close(st[f], c[pre + f], tol, what=pre + f)
    for f in ("invU", "U", "nu", "logdet_invU"):
        assert_close(st["W"][f], c[pre + "invU_" + f], tol, what=pre + "invU_" + f)


MNW_CASES = ["mnw_4x3_b5", "mnw_4x3_b5_pad", "mnw_4x3_nobatch", "mnw_32x32", "mnw_6x7_b2_pad"]


@pytest.mark.parametrize("case", MNW_CASES)
def test_mnw_updates_and_messages(golden, case):
    c = golden("mnw")[case]
    st, batch = _mnw_state(c)
    nb = len(batch)
    X, Y, pr = c["X"], c["Y"], c.get("p_resp")
    N = X.shape[0]
    Xe = X.expand((N,) + batch + X.shape[-2:])
    st = omnw.mnw_ss_update(st, *omnw.mnw_moments_data(st, Xe, Y, pr), lr=1.0)
    _check_mnw(st, c, "raw1_")
    st = omnw.mnw_ss_update(st, *omnw.mnw_moments_data(st, Xe, Y, pr), lr=0.5)
    _check_mnw(st, c, "raw2_")
    e = omnw.mnw_expectations(st)
    for f in ("EinvUX", "EXTinvU", "EXTinvUX", "EXinvVXT", "EXmMUTinvUXmMU", "EXmMUinvVXmMUT", "ElogdetinvU",
              "logdetEinvSigma", "ElogdetinvSigma", "EinvSigma", "invEinvSigma", "ESigma", "mean", "weights", "var"):
        assert_close(e[f], c["raw2_" + f], what=f)
    assert_close(omnw.mnw_kl(st), c["raw2_KLqprior"], what="KL")
    assert_close(omnw.mnw_elog_like(st, X, Y), c["Elog_like"], what="Elog_like")
    P, eta, R = omnw.mnw_elog_like_X(st, Y)
    assert_close(P, c["ELX_invSigma"])
    assert_close(eta, c["ELX_invSigmamu"])
    assert_close(R, c["ELX_Res"])
    P, eta, R = omnw.mnw_predict(st, X)
    assert_close(P, c["predict_invSigma"])
    assert_close(eta, c["predict_invSigmamu"])
    assert_close(R, c["predict_Res"], what="predict Res")
    P, eta, R = omnw.mnw_postdict(st, Y)
    assert_close(P, c["postdict_invSigma"])
    assert_close(eta, c["postdict_invSigmamu"])
    assert_close(R, c["postdict_Res"], what="postdict Res")
    # messages
    mu_y, Syy, R = omnw.mnw_forward(st, c["fw_in_invSigma"], c["fw_in_invSigmamu"])
    assert_close(mu_y, c["fw_mu"], what="fw mu")
    assert_close(Syy, c["fw_Sigma"], what="fw Sigma")
    assert_close(R, c["fw_Res"], what="fw Res")
    mu_y, Syy, R = omnw.mnw_forward(st, c["fws_in_invSigma"], c["fw_in_invSigmamu"])
    assert_close(mu_y, c["fws_mu"], what="fws mu")
    assert_close(Syy, c["fws_Sigma"], what="fws Sigma")
    assert_close(R, c["fws_Res"], what="fws Res")
    P, eta, R = omnw.mnw_backward(st, c["bw_in_invSigma"], c["bw_in_invSigmamu"])
    assert_close(P, c["bw_invSigma"], what="bw P")
    assert_close(eta, c["bw_invSigmamu"], what="bw eta")
    assert_close(R, c["bw_Res"], what="bw Res")
    P, eta, R = omnw.mnw_backward(st, c["bws_in_invSigma"], c["bw_in_invSigmamu"], Res=0.25)
    assert_close(P, c["bws_invSigma"], what="bws P")
    assert_close(eta, c["bws_invSigmamu"], what="bws eta")
    assert_close(R, c["bws_Res"], what="bws Res")
    P, eta, mu, S, R = omnw.mnw_elog_like_X_given_pY(st, c["bw_in_invSigma"], c["bw_in_invSigmamu"])
    assert_close(P, c["ELXpY_invSigma"])
    assert_close(eta, c["ELXpY_invSigmamu"])
    assert_close(mu, c["ELXpY_mu"])
    assert_close(S, c["ELXpY_Sigma"])
    assert_close(R, c["ELXpY_Res"], what="ELXpY Res")
    # update from distributions
    px = c["upd_x_mu"].shape[-2]
    EX = c["upd_x_mu"].expand((N,) + batch + (px, 1))
    Sx = c["upd_x_Sigma"].expand((N,) + batch + (px, px))
    EXXT = Sx + EX @ EX.transpose(-2, -1)
    EYYT = Y @ Y.transpose(-2, -1)
    assert_close(omnw.mnw_elog_like_dists(st, EX, EXXT, Y, EYYT), c["ELpXpY"], what="ELpXpY")
    st = omnw.mnw_ss_update(st, *omnw.mnw_moments_dists(st, EX, EXXT, Y, EYYT, pr), lr=0.8)
    _check_mnw(st, c, "upd_")
    n = int(c["n"])
    EYYT2 = c["upd_y_Sigma"].expand((N,) + batch + (n, n)) + Y @ Y.transpose(-2, -1)
    st = omnw.mnw_ss_update(st, *omnw.mnw_moments_dists(st, EX, EXXT, Y, EYYT2, pr), lr=1.0, beta=0.5)
    _check_mnw(st, c, "upd2_")
    st = omnw.mnw_ss_update(st, *omnw.mnw_moments_dists(st, EX, EXXT, Y, EYYT2, pr), lr=1.0, beta=0.5)
    _check_mnw(st, c, "upd3_")
    assert_close(omnw.mnw_kl(st), c["KLqprior_end"], what="KL end")


@pytest.mark.parametrize("case", ["mnw_Xmask", "mnw_mask", "mnw_mask_pad"])
def test_mnw_masks(golden, case):
    c = golden("mnw")[case]
    st, batch = _mnw_state(c)
    assert_close(st["mu_0"], c["init_mu_0"])
    X, Y, pr = c["X"], c["Y"], c["p_resp"]
    Xe = X.expand((X.shape[0],) + batch + X.shape[-2:])
    st = omnw.mnw_ss_update(st, *omnw.mnw_moments_data(st, Xe, Y, pr), lr=1.0)
    _check_mnw(st, c, "raw1_")
    st = omnw.mnw_ss_update(st, *omnw.mnw_moments_data(st, Xe, Y, pr), lr=0.5)
    _check_mnw(st, c, "raw2_")
    assert_close(omnw.mnw_kl(st), c["KLqprior"], what="KL")
    assert_close(omnw.mnw_elog_like(st, X, Y), c["Elog_like"], what="Elog_like")


# ---------------------------------------------------------------------- GMM
def test_gmm_two_moons(golden):
    c = golden("gmm")["gmm_k4_d2"]
    K, D = 4, 2
    st = oniw.niw_new((D,), (K,), scale=1.0 / K ** (1.0 / D), mu_init=c["init_mu"])
    assert_close(st["W"]["invU_0"], c["init_invU_0"])
    alpha_0, alpha = c["alpha_0"], c["init_alpha"]
    X = c["data"]
    for it in range(1, 21):
        st, alpha, out = omix.mixture_iteration(st, alpha_0, alpha, X, 1.0, (K,), (), (D,))
        if it in (1, 2, 5, 20):
            pre = f"it{it}_"
            # 20 EM iterations amplify rounding: allow 1e-9 at the end, 1e-10 early
            tol = TOL64 if it <= 2 else 1e-8
            for f in ("p", "NA", "logZ", "ELBO"):
                assert_close(out[f], c[pre + f], tol, what=pre + f)
            assert_close(alpha, c[pre + "alpha"], tol)
            assert_close(st["mu"], c[pre + "mu"], tol)
            assert_close(st["W"]["invU"], c[pre + "invU"], tol)
            assert_close(st["W"]["U"], c[pre + "U"], tol)
    assert_close(omix.mixture_kl(st, alpha_0, alpha), c["final_KLqprior"], 1e-8)


def test_mixture_batched(golden):
    c = golden("gmm")["mixture_b3_k6_e32"]
    st = oniw.niw_new((3, 2), (3, 6), mu_init=c["init_mu"])
    alpha_0 = torch.tensor(0.5, dtype=torch.float64).expand(3, 6)
    alpha = c["init_alpha"]
    for it in (1, 2, 3):
        st, alpha, out = omix.mixture_iteration(st, alpha_0, alpha, c["X"], 0.9, (6,), (3,), (3, 2))
        pre = f"it{it}_"
        for f in ("p", "NA", "logZ", "ELBO"):
            assert_close(out[f], c[pre + f], 1e-9, what=pre + f)
        assert_close(alpha, c[pre + "alpha"], 1e-9)
        assert_close(st["mu"], c[pre + "mu"], 1e-9)
        assert_close(st["W"]["invU"], c[pre + "invU"], 1e-9)


# ---------------------------------------------------------------------- MNW messages with a precision per message (BASELINE configs[2])
MNWMSG_CASES = ["msg_32x32", "msg_32x31_pad", "msg_32x32_pad", "msg_16x16_b3", "msg_24x32", "msg_32x20", "msg_8x40"]


def mnwmsg_oracle_state(c, dtype=torch.float64):
    """oracle state of the fitted transform stored by tools/gen_golden.py:mnwmsg_case"""
    batch = tuple(int(v) for v in c["batch_shape"])
    st = omnw.mnw_new((int(c["n"]), int(c["p"])), batch, mu_init=c["state_mu"].to(dtype), pad_X=bool(int(c["pad_X"])), dtype=dtype)
    for f in ("invV", "V", "logdetinvV"):
        st[f] = c["state_" + f].to(dtype)
    for f in ("invU", "U", "nu", "logdet_invU"):
        st["W"][f] = c["state_invU_" + f].to(dtype)
    return st


@pytest.mark.parametrize("case", MNWMSG_CASES)
def test_mnw_messages_per_message_precision(golden, case):
    c = golden("mnwmsg")[case]
    st = mnwmsg_oracle_state(c)
    mu_y, Syy, R = omnw.mnw_forward(st, c["fw_in_invSigma"].double(), c["fw_in_invSigmamu"].double())
    assert_close(mu_y, c["fw_mu"], what="fw mu")
    assert_close(Syy, c["fw_Sigma"], what="fw Sigma")
    assert_close(R, c["fw_Res"], what="fw Res")
    P, eta, R = omnw.mnw_backward(st, c["bw_in_invSigma"].double(), c["bw_in_invSigmamu"].double(), Res=c["bw_in_Res"].double())
    assert_close(P, c["bw_invSigma"], what="bw P")
    assert_close(eta, c["bw_invSigmamu"], what="bw eta")
    assert_close(R, c["bw_Res"], what="bw Res")


# ---------------------------------------------------------------------- HMM forward-backward
HMM_CASES = ["hmm_k25_roles", "hmm_k25_roles_ptemp", "hmm_k4", "hmm_k3_b2", "hmm_k9_T1", "hmm_k2_T2"]


@pytest.mark.parametrize("case", HMM_CASES)
def test_hmm_forward_backward(golden, case):
    from oracle import hmm as ohmm
    c = golden("hmm")[case]
    p, SEzz, SEz0, logZ = ohmm.forward_backward(c["logits"], c["trans"], c["init"], float(c["ptemp"]))
    assert_close(p, c["p"], what="p")
    assert_close(SEzz, c["SEzz"], what="SEzz")
    assert_close(SEz0, c["SEz0"], what="SEz0")
    assert_close(logZ, c["logZ"], what="logZ")
