"""Pin the oracle of the Polya-Gamma logistic gate (oracle/mnlr.py) and of the gated mixture of linear transforms
(built from oracle/mnw.py + oracle/mnlr.py) to fixtures captured from the reference (tests/golden/dmix.npz).  CPU only."""
import math

import pytest
import torch

from oracle import mnlr as omn
from oracle import mnw as omnw
from tests.helpers import assert_close

MNLR_CASES = ["mnlr_c4_p3", "mnlr_c3_p9_lr"]
DMIX_CASES = ["dmix_w_n2_p3_k3", "dmix_g_n3_p2_k2"]


def ard_state(c, pre):
    return omn.ard_from(c[pre + "mu"], c[pre + "invSigma"], c[pre + "invSigmamu"], c[pre + "Sigma"],
                        c[pre + "logdetinvSigma"], c[pre + "alpha"], c[pre + "beta"])


def check_ard(st, c, pre, tol=1e-9):
    for f in ("mu", "invSigma", "invSigmamu", "Sigma", "logdetinvSigma", "alpha", "beta"):
        assert_close(st[f], c[pre + f], tol, what=pre + f)


@pytest.mark.parametrize("case", MNLR_CASES)
def test_logistic_gate_oracle_golden(golden, case):
    c = golden("dmix")[case]
    lr = float(c["lr"])
    st = ard_state(c, "init_")
    X, Y, w = c["X"], c["Y"], c["w"]
    st = omn.mnlr_raw_update(st, X, Y, iters=2, lr=lr)
    check_ard(st, c, "r1_")
    st = omn.mnlr_raw_update(st, X, Y, iters=3, p=w, lr=lr)
    check_ard(st, c, "r2_")
    assert_close(omn.ard_kl(st), c["KLqprior"], 1e-9, what="KL")
    assert_close(omn.mnlr_elog_like(st, X, Y), c["Elog_like"], 1e-9, what="Elog_like")
    lp = omn.mnlr_log_predict(st, X[:9])
    assert_close(lp, c["log_predict"], 1e-9, what="log_predict")
    assert_close(torch.softmax(lp, -1), c["predict"], 1e-9, what="predict")
    assert_close(omn.mnlr_log_predict_1(st, X[:9]), c["log_predict_1"], 1e-9, what="log_predict_1")
    assert_close(omn.mnlr_log_predict_2(st, X[:9]), c["log_predict_2"], 1e-9, what="log_predict_2")
    assert_close(omn.mnlr_weights(st), c["weights"], 1e-9, what="weights")
    mu = X.unsqueeze(-1)
    assert_close(omn.mnlr_elog_like_dist(st, mu, c["SigX"], Y), c["ELpXpY"], 1e-9, what="ELpXpY")
    assert_close(omn.mnlr_log_forward(st, mu[:9], c["SigX"][:9]), c["log_forward"], 1e-9, what="log_forward")
    p0 = X.shape[-1]
    pY = c["bw_pY"]
    P, eta, Sig, m, Res = omn.mnlr_elog_like_X(st, torch.eye(p0, dtype=torch.float64).expand(1, p0, p0),
                                               torch.zeros(1, p0, 1, dtype=torch.float64), pY)
    assert_close(P, c["bw_invSigma"], 1e-9, what="bw P")
    assert_close(eta, c["bw_invSigmamu"], 1e-9, what="bw eta")
    assert_close(m, c["bw_mu"], 1e-9, what="bw mu")
    assert_close(Res, c["bw_Res"], 1e-9, what="bw Res")
    st = omn.mnlr_update(st, mu, c["SigX"], Y, iters=2, lr=lr)
    check_ard(st, c, "u1_")


def dmix_states(c):
    n, p, K = int(c["n"]), int(c["p"]), int(c["mix"])
    scale = 1.0 / K ** (1.0 / n)
    if int(c["gamma"]):
        A = omnw.mng_new((n, p), (K,), mu_init=c["init_A_mu"], alpha_init=c["init_A_alpha"], beta_init=c["init_A_beta"],
                         scale=scale, pad_X=True)
    else:
        A = omnw.mnw_new((n, p), (K,), mu_init=c["init_A_mu"], scale=scale, pad_X=True)
    return A, ard_state(c, "init_pi_"), n, p, K


def resp(log_p):
    logZ = torch.logsumexp(log_p, -1)
    return torch.exp(log_p - logZ.unsqueeze(-1)), logZ


@pytest.mark.parametrize("case", DMIX_CASES)
def test_gated_mixture_oracle_golden(golden, case):
    """dMixtureofLinearTransforms.raw_update / predict / Elog_like / postdict / update
    (ref transforms/dMixtureofLinearTransforms.py:35-121)"""
    from tests.test_oracle_lds import n_iters
    c = golden("dmix")[case]
    A, pi, n, p, K = dmix_states(c)
    lr = float(c["lr"])
    gamma = bool(int(c["gamma"]))
    X, Y = c["X"], c["Y"]
    N = X.shape[0]
    AX, AY = X.unsqueeze(-1).unsqueeze(-3), Y.unsqueeze(-1).unsqueeze(-3)
    for it in range(1, n_iters(c) + 1):
        pre = f"it{it}_"
        pa, _ = resp(omnw.mnw_elog_like(A, AX, AY) + omn.mnlr_log_predict(pi, X))
        pi = omn.mnlr_raw_update(pi, X, pa, iters=2, lr=lr)
        A = omnw.mnw_ss_update(A, *omnw.mnw_moments_data(A, AX.expand(N, K, p, 1), AY, pa), lr=lr)
        assert_close(A["mu"], c[pre + "A_mu"], 1e-9, what=pre + "A_mu")
        assert_close(A["invV"], c[pre + "A_invV"], 1e-9, what=pre + "A_invV")
        check_ard(pi, c, pre + "pi_")
    assert_close(omnw.mnw_kl(A).sum(-1) + omn.ard_kl(pi), c["KLqprior"], 1e-9, what="KL")
    assert_close(torch.logsumexp(omnw.mnw_elog_like(A, AX, AY) + omn.mnlr_log_predict(pi, X), -1), c["Elog_like"], 1e-9)
    # predict
    pg = torch.softmax(omn.mnlr_log_predict(pi, X[:7]), -1)
    P, eta, _ = omnw.mnw_predict(A, AX[:7])
    Sig = torch.linalg.inv(P).expand(7, K, n, n)
    m = Sig @ eta
    pv = pg.reshape(pg.shape + (1, 1))
    mu = (m * pv).sum(-3)
    assert_close(pg, c["pred_p"], 1e-9, what="pred p")
    assert_close(mu, c["pred_mu"], 1e-9, what="pred mu")
    assert_close(((Sig + m @ m.transpose(-2, -1)) * pv).sum(-3) - mu @ mu.transpose(-2, -1), c["pred_Sigma"], 1e-9)
    # expected log-likelihood with Gaussian inputs / outputs
    mx, my = X.unsqueeze(-1), Y.unsqueeze(-1)
    EXXT = (c["SigX"] + mx @ mx.transpose(-2, -1)).unsqueeze(-3)
    EYYT = (c["SigY"] + my @ my.transpose(-2, -1)).unsqueeze(-3)
    log_p = omnw.mnw_elog_like_dists(A, AX, EXXT, AY, EYYT) + omn.mnlr_log_forward(pi, mx, c["SigX"])
    assert_close(torch.logsumexp(log_p, -1), c["ELpXpY"], 1e-9, what="ELpXpY")
    # postdict: expert messages to x, gate message, mixed by evidence
    Pm, etam, Res = omnw.mnw_elog_like_X(A, Y[:5].unsqueeze(-2).unsqueeze(-1))
    like_P, like_eta = Pm.unsqueeze(0).movedim(-3, -3), etam.movedim(-3, -3)
    Z = torch.eye(K, dtype=torch.float64)
    P2, eta2, Sig2, mu2, Rz = omn.mnlr_elog_like_X(pi, like_P, like_eta, Z, iters=4)
    R = Res + Rz + 0.5 * (mu2 * eta2).sum(-2).squeeze(-1) - 0.5 * torch.logdet(P2) + p / 2.0 * math.log(2 * math.pi)
    logZ = R.logsumexp(-1, True)
    pp = (R - logZ).exp()
    pv = pp.reshape(pp.shape + (1, 1))
    assert_close(pp, c["post_p"], 1e-9, what="post p")
    assert_close(logZ.squeeze(-1), c["post_logZ"], 1e-9, what="post logZ")
    assert_close((P2 * pv).sum(-3), c["post_invSigma"], 1e-9, what="post P")
    assert_close((eta2 * pv).sum(-3), c["post_invSigmamu"], 1e-9, what="post eta")
    # update(pX, pY)
    pa, logZ = resp(log_p)
    assert_close(logZ, c["upd_logZ"], 1e-9, what="upd logZ")
    assert_close(pa.sum(0), c["upd_NA"], 1e-9, what="upd NA")
    pi = omn.mnlr_update(pi, mx, c["SigX"], pa, iters=2, lr=lr)
    A = omnw.mnw_ss_update(A, *omnw.mnw_moments_dists(A, AX.expand(N, K, p, 1), EXXT.expand(N, K, p, p), AY, EYYT, pa), lr=lr)
    assert_close(A["mu"], c["upd_A_mu"], 1e-9, what="upd A_mu")
    check_ard(pi, c, "upd_pi_")
    assert_close(logZ.sum() - (omnw.mnw_kl(A).sum(-1) + omn.ard_kl(pi)), c["upd_ELBO"], 1e-9, what="upd ELBO")
