"""Oracle: linear-dynamical-system VB E-step (information filter + smoother) and the M-step glue
(torch CPU, LU route).  TEST INFRASTRUCTURE ONLY.

Reference: models/LinearDynamicalSystems.py (latent_noise='shared': the transition is a
MatrixNormalWishart).  Shapes follow the reference: time first, then sample dims, batch dims, one
singleton "offset" axis per extra observation axis, then the (dim, 1) vector-format event.
"""
import math

import torch

from . import mnw as _mnw
from . import niw as _niw

LOG2PI = math.log(2.0 * math.pi)
inv = torch.linalg.inv
logdet = torch.logdet
T_ = lambda a: a.transpose(-2, -1)  # noqa: E731


def _sq(a):
    return a.squeeze(-1).squeeze(-1)


def reshape_inputs(y, u, r, obs_shape, control_dim, regression_dim, batch_shape=(), expand_to_batch=False):
    """vector format + appended ones.  control_dim / regression_dim INCLUDE the appended one.
    ref models/LinearDynamicalSystems.py:56-83"""
    sample_shape = y.shape[: y.ndim - len(obs_shape)]
    y = y.unsqueeze(-1)
    one = torch.ones((), dtype=y.dtype)
    if u is None:
        u = one.expand(sample_shape + (control_dim, 1))
    else:
        u = torch.cat((u, torch.ones(u.shape[:-1] + (1,), dtype=y.dtype)), -1).unsqueeze(-1)
    if r is None:
        r = one.expand(sample_shape + tuple(obs_shape[:-1]) + (regression_dim, 1))
    else:
        r = torch.cat((r, torch.ones(r.shape[:-1] + (1,), dtype=y.dtype)), -1).unsqueeze(-1)
    if expand_to_batch:
        for _ in batch_shape:
            y, u, r = (v.unsqueeze(len(sample_shape)) for v in (y, u, r))
        y = y.expand(sample_shape + tuple(batch_shape) + tuple(obs_shape) + (1,))
        u = u.expand(sample_shape + tuple(batch_shape) + (control_dim, 1))
        r = r.expand(sample_shape + tuple(batch_shape) + tuple(obs_shape[:-1]) + (regression_dim, 1))
    for _ in range(len(obs_shape) - 1):
        u = u.unsqueeze(-3)
    return y, u, r


def latent_parms(A_state, h):
    """ref :230-242 (set_latent_parms)"""
    e = _mnw.mnw_expectations(A_state)
    ATQA, QA = e["EXTinvUX"], e["EinvUX"]
    xx = ATQA[..., :h, :h]
    return {"invQ": e["EinvSigma"], "ATQA_x_x": xx, "invATQA_x_x": inv(xx), "logdetATQA_x_x": logdet(xx),
            "ATQA_x_u": ATQA[..., :h, h:], "ATQA_u_u": ATQA[..., h:, h:], "QA_xp_x": QA[..., :, :h],
            "QA_xp_u": QA[..., :, h:], "A_Elogdet": e["ElogdetinvSigma"]}


def log_likelihood(obs_state, Y, R, h, n_extra_obs):
    """Likelihood of x_t as natural parameters.  ref :244-266"""
    e = _mnw.mnw_expectations(obs_state)
    BTRB, BTR = e["EXTinvUX"], e["EXTinvU"]
    P = BTRB[..., :h, :h]
    eta = BTR[..., :h, :] @ Y - BTRB[..., :h, h:] @ R
    res = -0.5 * T_(Y) @ e["EinvSigma"] @ Y - 0.5 * T_(R) @ BTRB[..., h:, h:] @ R + T_(R) @ BTR[..., h:, :] @ Y
    res = _sq(res) + 0.5 * e["ElogdetinvSigma"] - 0.5 * obs_state["n"] * LOG2PI
    for i in range(n_extra_obs):
        P = P.sum(-3 - i, True)
        eta = eta.sum(-3 - i, True)
        res = res.sum(-1 - i, True)
    P = P.expand(eta.shape[:-2] + (h, h))
    return P, eta, res


def forward_step(lp, h, P, eta, res, P_like, eta_like, res_like, U):
    """ref :268-288"""
    S = inv(P + lp["ATQA_x_x"])
    eta_t = eta_like + lp["QA_xp_u"] @ U
    eta_m = eta - lp["ATQA_x_u"] @ U
    Pn = P_like + lp["invQ"] - lp["QA_xp_x"] @ S @ T_(lp["QA_xp_x"])
    etan = eta_t + lp["QA_xp_x"] @ S @ eta_m
    res = res + res_like - 0.5 * _sq(T_(U) @ lp["ATQA_u_u"] @ U) + 0.5 * lp["A_Elogdet"]
    res = res + 0.5 * _sq(T_(eta_m) @ S @ eta_m) + 0.5 * logdet(S)
    mu = inv(Pn) @ etan
    post = -0.5 * (mu * etan).squeeze(-1).sum(-1) + 0.5 * logdet(Pn) - 0.5 * h * LOG2PI
    return Pn, etan, post, res - post, S


def backward_step(lp, G, g, P_like, eta_like, U):
    """ref :296-302"""
    S = inv(lp["invQ"] + P_like + G)
    Gn = lp["ATQA_x_x"] - T_(lp["QA_xp_x"]) @ S @ lp["QA_xp_x"]
    gn = -lp["ATQA_x_u"] @ U + T_(lp["QA_xp_x"]) @ S @ (lp["QA_xp_u"] @ U + eta_like + g)
    return Gn, gn


def smoother(lp, x0_state, h, y, u, r, obs_state, n_extra_obs, like=None):
    """forward_backward_loop, ref :332-383.  Returns dict with px.* (T first), Sigma_t_tp1, Sigma_x0_x0,
    mu_x0, logZ."""
    Tn = y.shape[0]
    P_like, eta_like, res_like = like if like is not None else log_likelihood(obs_state, y, r, h, n_extra_obs)
    lead = eta_like.shape[:-2]  # (T, sample, batch, offset)
    dt = y.dtype
    pP = torch.zeros(lead + (h, h), dtype=dt)
    pe = torch.zeros(lead + (h, 1), dtype=dt)
    pS = torch.zeros(lead + (h, h), dtype=dt)
    pm = torch.zeros(lead + (h, 1), dtype=dt)
    logZ = torch.zeros(lead, dtype=dt)
    cross = torch.zeros(lead + (h, h), dtype=dt)
    x0e = _niw.niw_expectations(x0_state)
    pP[-1] = x0e["EinvSigma"]
    pe[-1] = x0e["EinvSigmamu"].unsqueeze(-1)
    res = -0.5 * x0e["EXTinvUX"] + 0.5 * x0e["ElogdetinvSigma"] - 0.5 * h * LOG2PI
    for t in range(Tn):
        pP[t], pe[t], res, logZ[t], cross[t - 1] = forward_step(lp, h, pP[t - 1], pe[t - 1], res, P_like[t],
                                                                eta_like[t], res_like[t], u[t])
    pS[-1] = inv(pP[-1])
    pm[-1] = pS[-1] @ pe[-1]
    G = torch.zeros(pP.shape[1:], dtype=dt)
    g = torch.zeros(pe.shape[1:], dtype=dt)
    QA = lp["QA_xp_x"]
    for t in range(Tn - 2, -1, -1):
        # NB the elementwise `*` in the last factor is the reference's (:372), kept for parity
        cross[t] = cross[t] @ T_(QA) @ inv(G + P_like[t + 1] + lp["invQ"] - QA @ cross[t] * T_(QA))
        G, g = backward_step(lp, G, g, P_like[t + 1], eta_like[t + 1], u[t + 1])
        pP[t] = pP[t] + G
        pe[t] = pe[t] + g
        pS[t] = inv(pP[t])
        pm[t] = pS[t] @ pe[t]
    cross[-1] = cross[-1] @ T_(QA) @ inv(G + P_like[0] + lp["invQ"] - QA @ cross[-1] * T_(QA))
    G, g = backward_step(lp, G, g, P_like[0], eta_like[0], u[0])
    S00 = inv(G + x0e["EinvSigma"])
    m0 = S00 @ (g + x0e["EinvSigmamu"].unsqueeze(-1))
    return {"mu": pm, "Sigma": pS, "invSigma": pP, "invSigmamu": pe, "Sigma_t_tp1": cross, "Sigma_x0_x0": S00,
            "mu_x0": m0, "logZ": logZ}


def latent_stats(sm, y, u, r, obs_shape, control_dim, regression_dim, batch_shape, n_extra_obs):
    """Time-integrated sufficient statistics.  ref :173-216"""
    mu, Sg, cross, S00, m0 = sm["mu"], sm["Sigma"], sm["Sigma_t_tp1"], sm["Sigma_x0_x0"], sm["mu_x0"]
    SE_x0_x0 = S00 + m0 @ T_(m0)
    SE_x_x = (mu @ T_(mu) + Sg).sum(0)
    SE_xp_xp = SE_x_x - (mu[-1] @ T_(mu[-1]) + Sg[-1]) + SE_x0_x0
    SE_x_u = (mu @ T_(u)).sum(0)
    SE_xp_u = (mu[:-1] @ T_(u[1:])).sum(0) + m0 @ T_(u[0])
    SE_xp_x = (mu[:-1] @ T_(mu[1:])).sum(0) + cross[:-1].sum(0) + m0 @ T_(mu[0]) + cross[-1]
    SE_x_r = (mu @ T_(r)).sum(0)
    SE_x_y = (mu @ T_(y)).sum(0)
    SE_u_u = (u @ T_(u)).sum(0)
    SE_r_r = (r @ T_(r)).sum(0)
    SE_y_y = (y @ T_(y)).sum(0)
    SE_y_r = (y @ T_(r)).sum(0)
    ed, bd = len(obs_shape), len(batch_shape)
    sample_shape = tuple(y.shape[1: y.ndim - ed - bd - 1])
    offset = (1,) * n_extra_obs
    bs, os_ = tuple(batch_shape), tuple(obs_shape)
    SE_y_r = SE_y_r.expand(sample_shape + bs + os_ + (regression_dim,))
    SE_u_u = SE_u_u.expand(sample_shape + bs + offset + (control_dim, control_dim))
    SE_r_r = SE_r_r.expand(sample_shape + bs + os_[:-1] + (regression_dim, regression_dim))
    out = {"T": y.shape[0] * torch.ones(sample_shape + bs + offset, dtype=y.dtype),
           "N": torch.ones(sample_shape + bs + offset, dtype=y.dtype),
           "SE_x_x": SE_x_x, "SE_x0_x0": SE_x0_x0, "SE_x0": m0,
           "SE_y_xr": torch.cat((T_(SE_x_y), SE_y_r), -1), "SE_y_y": SE_y_y,
           "SE_xpu_xpu": torch.cat((torch.cat((SE_xp_xp, SE_xp_u), -1), torch.cat((T_(SE_xp_u), SE_u_u), -1)), -2),
           "SE_x_xpu": torch.cat((T_(SE_xp_x), SE_x_u), -1)}
    xx = SE_x_x.expand(SE_x_r.shape[:-2] + SE_x_x.shape[-2:])
    out["SE_xr_xr"] = torch.cat((torch.cat((xx, SE_x_r), -1), torch.cat((T_(SE_x_r), SE_r_r), -1)), -2)
    lz = sm["logZ"]
    for _ in range(n_extra_obs):
        lz = lz.squeeze(-1)
    out["logZ"] = lz.sum(0)
    return out


def reduce_stats(st, batch_dim, n_extra_obs, p=None):
    """(per-series weights p, ref :106-121, then) sum over the sample axes + symmetrise.  ref :135-150"""
    st = dict(st)
    keys = ("SE_x0_x0", "SE_x0", "SE_xpu_xpu", "SE_x_xpu", "SE_x_x", "SE_xr_xr", "SE_y_xr", "SE_y_y", "T", "N")
    if p is not None:
        for _ in range(n_extra_obs):
            p = p.unsqueeze(-1)
        st["T"], st["N"] = st["T"] * p, st["N"] * p
        pm = p.unsqueeze(-1).unsqueeze(-1)
        for k in keys[:-2]:
            st[k] = st[k] * pm
    while st["SE_x_x"].ndim > batch_dim + n_extra_obs + 2:
        for k in keys:
            st[k] = st[k].sum(0)
    for k in ("SE_x0_x0", "SE_xpu_xpu", "SE_x_x", "SE_xr_xr"):
        st[k] = 0.5 * (st[k] + T_(st[k]))
    return st
