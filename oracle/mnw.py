"""Oracle: matrix-normal-Wishart regression node (torch CPU, LU route).  TEST INFRASTRUCTURE ONLY.

y = A x + noise, A ~ matrix normal (n x p), noise precision ~ Wishart (n x n).
Reference: transforms/MatrixNormalWishart.py.  State is a dict; functions are pure.
Messages are (invSigma, invSigmamu) / (mu, Sigma) tuples in vector format (trailing (dim, 1)).
"""
import math

import torch

from . import mvn as _mvn
from . import niw as _niw
from .mutils import precision_marginalizer

LOG2PI = math.log(2.0 * math.pi)
inv = torch.linalg.inv
T = lambda a: a.transpose(-2, -1)  # noqa: E731


def _sq(a):
    return a.squeeze(-1).squeeze(-1)


# ------------------------------------------------------------------ noise models
def gamma_new(event_shape, batch_shape, scale, alpha_init, beta_init, dtype=torch.float64):
    """DiagonalWishart / Gamma prior.  ref dists/DiagonalWishart.py:9-20, dists/Gamma.py:7-24 (nu=2, U=0.5).
    alpha_init / beta_init are the stored random draws of Gamma.py:20-21."""
    full = tuple(batch_shape) + tuple(event_shape)
    a0 = torch.tensor(2.0, dtype=dtype).expand(full)
    b0 = torch.tensor(scale ** 2 / 0.5, dtype=dtype).expand(full)
    return {"kind": "gamma", "alpha_0": a0, "beta_0": b0, "alpha": alpha_init, "beta": beta_init,
            "event_dim": len(event_shape)}


def _diag(v):
    return v.unsqueeze(-1) * torch.eye(v.shape[-1], dtype=v.dtype)


def _noise_expect(W):
    """expectations of the noise precision for either a Wishart or a diagonal-Gamma state"""
    if W.get("kind") == "gamma":
        mean, minv = W["alpha"] / W["beta"], W["beta"] / (W["alpha"] - 1)
        lgm = (W["alpha"].log() - W["beta"].log()).sum(-1)
        return {"EinvSigma": _diag(mean), "ESigma": _diag(minv), "invEinvSigma": _diag(1.0 / mean),
                "ElogdetinvSigma": lgm, "logdetEinvSigma": mean.log().sum(-1)}
    return _niw.wishart_expectations(W)


def _noise_update(W, arg, N, lr):
    if W.get("kind") == "gamma":  # ref transforms/MatrixNormalGamma.py:126, dists/DiagonalWishart.py:32-37, Gamma.py:34-46
        W = dict(W)
        d = arg.diagonal(dim1=-2, dim2=-1)
        W["alpha"] = (W["alpha_0"] + N.unsqueeze(-1) / 2.0) * lr + W["alpha"] * (1 - lr)
        W["beta"] = (W["beta_0"] + d / 2.0) * lr + W["beta"] * (1 - lr)
        return W
    return _niw.wishart_ss_update(W, arg, N, lr=lr, beta=None)


def _noise_kl(W, ed):
    if W.get("kind") == "gamma":  # ref dists/Gamma.py:102-104 (summed over the Gamma's event dims)
        a, b, a0, b0 = W["alpha"], W["beta"], W["alpha_0"], W["beta_0"]
        kl = (a - a0) * torch.digamma(a) - torch.lgamma(a) + torch.lgamma(a0) + a0 * (b.log() - b0.log()) + a * (b0 / b - 1)
        return kl.sum(tuple(range(-(ed - 1), 0)))
    return _niw.wishart_kl(W, ed)


def mnw_new(event_shape, batch_shape=(), mu_init=None, mu_0=0.0, scale=1.0, mask=None, X_mask=None, pad_X=False,
            fixed_precision=False, dtype=torch.float64):
    """ref transforms/MatrixNormalWishart.py:20-70.  mu_init = the stored random draw of :42 (already masked)."""
    es, bs = tuple(event_shape), tuple(batch_shape)
    n, p = es[-2], es[-1]
    m0 = torch.as_tensor(mu_0, dtype=dtype)
    if pad_X:
        p += 1
        es = es[:-1] + (p,)
        if m0.ndim != 0:
            m0 = torch.cat((m0, m0.new_zeros(m0.shape[:-1] + (1,))), -1)
    m0 = m0.expand(bs + es)
    invV_0 = torch.eye(p, dtype=dtype).expand(bs + es[:-2] + (p, p))
    st = {
        "n": n, "p": p, "pad_X": pad_X, "fixed_precision": fixed_precision, "event_dim": len(es), "batch_dim": len(bs),
        "mu_0": m0, "mu": m0 if mu_init is None else mu_init, "invV_0": invV_0, "invV": invV_0, "V": inv(invV_0),
        "logdetinvV": torch.logdet(invV_0), "logdetinvV_0": torch.logdet(invV_0),
        "W": _niw.wishart_new(es[:-2] + (n, n), bs, scale, dtype), "mask": mask, "X_mask": X_mask,
        "acc": None,
    }
    if X_mask is not None:
        if pad_X:
            st["X_mask"] = X_mask = torch.cat((X_mask, torch.ones(X_mask.shape[:-1] + (1,), dtype=torch.bool)), -1)
        st["mu_0"] = st["mu_0"] * X_mask
        st["V"] = st["V"] * X_mask * T(X_mask)
        st["invV"] = st["invV"] * X_mask * T(X_mask)
    if mask is not None:
        if pad_X:
            st["mask"] = mask = torch.cat((mask, torch.ones(mask.shape[:-1] + (1,), dtype=torch.bool)), -1)
        st["mu_0"] = st["mu_0"] * mask
    return st


def mnw_ss_update(st, SExx, SEyx, SEyy, N, lr=1.0, beta=None):
    """ref transforms/MatrixNormalWishart.py:82-141"""
    st = dict(st)
    if beta is not None:
        a = st["acc"] or (0.0, 0.0, 0.0, 0.0)
        st["acc"] = (beta * a[0] + SExx, beta * a[1] + SEyx, beta * a[2] + SEyy, beta * a[3] + N)
        SExx, SEyx, SEyy, N = st["acc"]
    m0, iV0, Xm, mask = st["mu_0"], st["invV_0"], st["X_mask"], st["mask"]
    if Xm is not None:
        SExx = SExx * Xm * T(Xm)
        SEyx = SEyx * Xm
        invV = iV0 + SExx
        mu = (m0 @ iV0 + SEyx) @ inv(invV)
        mu = mu * Xm
    else:
        invV = iV0 + SExx
        mu = T(torch.linalg.solve(invV, T(m0 @ iV0 + SEyx)))
    if mask is not None:  # constrained posterior mean, same mask for the whole batch (:111-120)
        V = inv(invV)
        U = inv(_noise_expect(st["W"])["EinvSigma"])
        Astar = V.unsqueeze(-3).unsqueeze(-2) * U.unsqueeze(-2).unsqueeze(-1)
        off = ~mask
        A = Astar[..., off, :, :][..., :, off]
        gamma = torch.zeros_like(mu)
        gamma[..., off] = torch.linalg.solve(A, mu[..., off])
        mu = (mu - U @ gamma @ V) * mask
    if not st["fixed_precision"]:
        arg = SEyy - mu @ invV @ T(mu) + m0 @ iV0 @ T(m0)
        st["W"] = _noise_update(st["W"], arg, N, lr)
    invV = lr * invV + (1.0 - lr) * st["invV"]
    st["invV"] = 0.5 * (invV + T(invV))
    st["mu"] = lr * mu + (1.0 - lr) * st["mu"]
    if mask is not None:
        st["mu"] = st["mu"] * mask
    st["V"] = inv(st["invV"])
    st["logdetinvV"] = torch.logdet(st["invV"])
    if Xm is not None:
        st["mu"] = st["mu"] * Xm
    return st


def _pad_stats(SExx, SEyx, SEx, SEy, N):
    """bias-column augmentation.  ref transforms/MatrixNormalWishart.py:159-170 / :191-202"""
    SExx = torch.cat((SExx, SEx), -1)
    col = torch.cat((SEx, N.reshape(N.shape + (1, 1))), -2)
    SExx = torch.cat((SExx, T(col)), -2)
    SEyx = torch.cat((SEyx, SEy.expand(SEyx.shape[:-1] + (1,))), -1)
    return SExx, SEyx


def mnw_moments_data(st, X, Y, p=None):
    """Sufficient statistics from data.  ref transforms/MatrixNormalWishart.py:174-202"""
    nd = st["event_dim"] + st["batch_dim"]
    sample_shape = X.shape[: X.ndim - nd]
    sd = tuple(range(len(sample_shape)))
    bshape = X.shape[len(sample_shape): X.ndim - 2]
    if p is None:
        w = 1.0
        cnt = 1
        for s in sample_shape:
            cnt *= s
        N = torch.tensor(float(cnt), dtype=X.dtype).expand(bshape)
    else:
        N = p.sum(sd)
        w = p.reshape(p.shape + (1,) * st["event_dim"])
    SExx = (X * T(X) * w).sum(sd)
    SEyy = (Y * T(Y) * w).sum(sd)
    SEyx = (Y * T(X) * w).sum(sd)
    if st["pad_X"]:
        SExx, SEyx = _pad_stats(SExx, SEyx, (X * w).sum(sd), (Y * w).sum(sd), N)
    return SExx, SEyx, SEyy, N


def mnw_moments_dists(st, EX, EXXT, EY, EYYT, p=None):
    """Sufficient statistics from input/output distributions.  ref transforms/MatrixNormalWishart.py:143-170"""
    nd = st["event_dim"] + st["batch_dim"]
    sample_shape = EX.shape[: EX.ndim - nd]
    sd = tuple(range(len(sample_shape)))
    bshape = EX.shape[len(sample_shape): EX.ndim - 2]
    if p is None:
        w = 1.0
        cnt = 1
        for s in sample_shape:
            cnt *= s
        N = torch.tensor(float(cnt), dtype=EX.dtype).expand(bshape)
    else:
        N = p.sum(sd)
        w = p.reshape(p.shape + (1,) * st["event_dim"])
    SExx = (EXXT * w).sum(sd)
    SEyy = (EYYT * w).sum(sd)
    SEyx = ((EY @ T(EX)) * w).sum(sd)
    if st["pad_X"]:
        SExx, SEyx = _pad_stats(SExx, SEyx, (EX * w).sum(sd), (EY * w).sum(sd), N)
    return SExx, SEyx, SEyy, N


def mng_new(event_shape, batch_shape=(), mu_init=None, alpha_init=None, beta_init=None, scale=1.0, mask=None,
            X_mask=None, pad_X=False, dtype=torch.float64):
    """MatrixNormalGamma state: an mnw state whose noise model is the diagonal Gamma one.
    ref transforms/MatrixNormalGamma.py:21-79 (NB :47: the initial mean is drawn without the prior mean)."""
    st = mnw_new(event_shape, batch_shape, mu_init=mu_init, scale=scale, mask=mask, X_mask=X_mask, pad_X=pad_X,
                 dtype=dtype)
    es = tuple(event_shape)
    if pad_X:
        es = es[:-1] + (es[-1] + 1,)
    st["W"] = gamma_new(es[:-1], batch_shape, scale, alpha_init, beta_init, dtype)
    return st


def mng_forward(st, Px, etax):
    """natural-parameter forward message of MatrixNormalGamma (no Res).  ref transforms/MatrixNormalGamma.py:315-335"""
    e = mnw_expectations(st)
    G, H = e["EinvUX"], e["EXTinvUX"]
    if st["pad_X"]:
        Jyx, Jxx, jy, jx = -G[..., :, :-1], H[..., :-1, :-1] + Px, G[..., :, -1:], etax - H[..., :-1, -1:]
    else:
        Jyx, Jxx, jy, jx = -G, H + Px, torch.zeros(e["EinvSigma"].shape[:-1] + (1,), dtype=Px.dtype), etax
    Pyy, nBiD = precision_marginalizer(e["EinvSigma"], Jyx, T(Jyx), Jxx)[0:2]
    return Pyy, jy + nBiD @ jx


def mnw_expectations(st):
    """ref transforms/MatrixNormalWishart.py:400-471"""
    we = _noise_expect(st["W"])
    R, mu, V, n, p = we["EinvSigma"], st["mu"], st["V"], st["n"], st["p"]
    return {
        "EinvSigma": R, "ESigma": we["ESigma"], "invEinvSigma": we["invEinvSigma"],
        "ElogdetinvSigma": we["ElogdetinvSigma"], "ElogdetinvU": we["ElogdetinvSigma"],
        "logdetEinvSigma": we["logdetEinvSigma"],
        "EinvUX": R @ mu, "EXTinvU": T(mu) @ R, "EXTinvUX": n * V + T(mu) @ R @ mu,
        "EXinvVXT": p * we["ESigma"] + mu @ st["invV"] @ T(mu),
        "EXmMUTinvUXmMU": n * V, "EXmMUinvVXmMUT": p * we["ESigma"],
        "mean": mu, "weights": mu[..., :-1] if st["pad_X"] else mu,
        "var": we["ESigma"].diagonal(dim1=-1, dim2=-2).unsqueeze(-1) * V.diagonal(dim1=-1, dim2=-2).unsqueeze(-2),
    }


def mnw_kl(st, event_dim=None):
    """ref transforms/MatrixNormalWishart.py:206-216"""
    n, p = st["n"], st["p"]
    ed = st["event_dim"] if event_dim is None else event_dim
    kl = n / 2.0 * st["logdetinvV"] - n / 2.0 * st["logdetinvV_0"] - n * p / 2.0
    if st["X_mask"] is not None:
        kl = kl + n / 2.0 * st["logdetinvV_0"] * st["X_mask"].sum((-1, -2))
    kl = kl + 0.5 * n * (st["invV_0"] * st["V"]).sum((-1, -2))
    d = st["mu"] - st["mu_0"]
    R = _noise_expect(st["W"])["EinvSigma"]
    kl = kl + 0.5 * (st["invV_0"] * (T(d) @ R @ d)).sum((-1, -2))
    for _ in range(ed - 2):
        kl = kl.sum(-1)
    if st["W"].get("kind") == "gamma":  # ref transforms/MatrixNormalGamma.py:216-231 (sums the extra event dims again)
        kl = kl + _noise_kl(st["W"], ed)
        for _ in range(ed - 2):
            kl = kl.sum(-1)
        return kl
    return kl + _niw.wishart_kl(st["W"], ed)


def mnw_elog_like(st, X, Y):
    """ref transforms/MatrixNormalWishart.py:219-232"""
    e = mnw_expectations(st)
    out = -0.5 * _sq(T(Y) @ e["EinvSigma"] @ Y)
    G, H = e["EinvUX"], e["EXTinvUX"]
    if st["pad_X"]:
        out = out + _sq(T(Y) @ (G[..., :, :-1] @ X + G[..., :, -1:]))
        out = out - 0.5 * _sq(T(X) @ H[..., :-1, :-1] @ X + 2 * H[..., -1:, :-1] @ X + H[..., -1:, -1:])
    else:
        out = out + _sq(T(Y) @ G @ X) - 0.5 * _sq(T(X) @ H @ X)
    out = out + 0.5 * e["ElogdetinvSigma"] - 0.5 * st["n"] * LOG2PI
    for _ in range(st["event_dim"] - 2):
        out = out.sum(-1)
    return out


def mnw_elog_like_dists(st, EX, EXXT, EY, EYYT):
    """ref transforms/MatrixNormalWishart.py:234-249"""
    e = mnw_expectations(st)
    G, H = e["EinvUX"], e["EXTinvUX"]
    out = -0.5 * (EYYT * e["EinvSigma"]).sum((-1, -2))
    if st["pad_X"]:
        out = out + _sq(T(EY) @ (G[..., :, :-1] @ EX + G[..., :, -1:]))
        out = out - 0.5 * (EXXT * H[..., :-1, :-1]).sum((-1, -2))
        out = out - _sq(H[..., -1:, :-1] @ EX) - 0.5 * H[..., -1, -1]
    else:
        out = out + _sq(T(EY) @ G @ EX) - 0.5 * (EXXT * H).sum((-1, -2))
    out = out + 0.5 * e["ElogdetinvSigma"] - 0.5 * st["n"] * LOG2PI
    for _ in range(st["event_dim"] - 2):
        out = out.sum(-1)
    return out


def mnw_elog_like_X(st, Y):
    """likelihood of X given observed Y as natural parameters.  ref transforms/MatrixNormalWishart.py:251-261"""
    e = mnw_expectations(st)
    H, Gt = e["EXTinvUX"], e["EXTinvU"]
    res = -0.5 * _sq(T(Y) @ e["EinvSigma"] @ Y) - 0.5 * st["n"] * LOG2PI + 0.5 * e["ElogdetinvSigma"]
    if st["pad_X"]:
        return H[..., :-1, :-1], Gt[..., :-1, :] @ Y - H[..., :-1, -1:], res - 0.5 * H[..., -1, -1]
    return H, Gt @ Y, res


def _res_nat(P, eta):
    mu = inv(P) @ eta
    return _mvn.residual(mu, eta, torch.logdet(P), P.shape[-1])


def mnw_predict(st, X):
    """ref transforms/MatrixNormalWishart.py:381-390"""
    e = mnw_expectations(st)
    G, H = e["EinvUX"], e["EXTinvUX"]
    if st["pad_X"]:
        eta = G[..., :, :-1] @ X + G[..., :, -1:]
        res = -0.5 * T(X) @ H[..., :-1, :-1] @ X - H[..., -1:, :-1] @ X - 0.5 * H[..., -1:, -1:]
    else:
        eta = G @ X
        res = -0.5 * T(X) @ H @ X
    res = _sq(res) + 0.5 * e["ElogdetinvSigma"] - 0.5 * st["n"] * LOG2PI
    return e["EinvSigma"], eta, res - _res_nat(e["EinvSigma"], eta)


def mnw_postdict(st, Y):
    """ref transforms/MatrixNormalWishart.py:392-395"""
    P, eta, res = mnw_elog_like_X(st, Y)
    return P, eta, res - _res_nat(P, eta)


def mnw_forward(st, Px, etax):
    """Message x -> y given natural parameters of p(x).  Returns (mu_y, Sigma_yy, Res).
    ref transforms/MatrixNormalWishart.py:303-328"""
    e = mnw_expectations(st)
    n, V, M = st["n"], st["V"], st["mu"]
    Sx = inv(Px)
    mux = Sx @ etax
    if not st["pad_X"]:
        S = inv(n * V + Px)
        eta = etax
        mu_y = M @ (S @ eta)
        Syy = M @ S @ T(M) + e["invEinvSigma"]
        res = -0.5 * _sq(T(mux) @ Px @ mux) + 0.5 * _sq(T(eta) @ S @ eta)
        res = res - 0.5 * torch.logdet(n * V @ Sx + torch.eye(st["p"], dtype=Px.dtype))
    else:
        S = inv(Px + n * V[..., :-1, :-1])
        eta = etax - n * V[..., :-1, -1:]
        mu_y = M[..., :-1] @ (S @ eta) + M[..., -1:]
        Syy = M[..., :-1] @ S @ T(M[..., :-1]) + e["invEinvSigma"]
        res = -0.5 * _sq(T(mux) @ Px @ mux) + 0.5 * _sq(T(eta) @ S @ eta) - 0.5 * n * V[..., -1, -1]
        res = res - 0.5 * torch.logdet(n * V[..., :-1, :-1] @ Sx + torch.eye(st["p"] - 1, dtype=Px.dtype))
    return mu_y, Syy, res


def _joint_blocks(st, Py, etay, sign_bias):
    e = mnw_expectations(st)
    G, H = e["EinvUX"], e["EXTinvUX"]
    Jyy = Py + e["EinvSigma"]
    if st["pad_X"]:
        return (Jyy, -G[..., :, :-1], H[..., :-1, :-1], etay + sign_bias * G[..., :, -1:], -H[..., :-1, -1:],
                H[..., -1, -1], e)
    px = st["p"]
    return Jyy, -G, H, etay, H.new_zeros(H.shape[:-1] + (1,)) if H.ndim > 2 else H.new_zeros(px, 1), H.new_zeros(()), e


def mnw_backward(st, Py, etay, Res=0.0):
    """Message y -> x.  Returns (invSigma_x, invSigmamu_x, Res).  ref transforms/MatrixNormalWishart.py:352-375"""
    Jyy, Jyx, Jxx, jy, jx, J11, e = _joint_blocks(st, Py, etay, +1.0)
    Pyy, nBiD, nCiA, Pxx = precision_marginalizer(Jyy, Jyx, T(Jyx), Jxx)
    eta_y = jy + nBiD @ jx
    eta_x = jx + nCiA @ jy
    n = Py.shape[-1]
    out = Res + _res_nat(Py, etay) + 0.5 * _sq(T(eta_y) @ inv(Pyy) @ eta_y) - 0.5 * torch.logdet(Pyy)
    out = out + 0.5 * n * LOG2PI + 0.5 * e["ElogdetinvSigma"] - 0.5 * J11
    return Pxx, eta_x, out - _res_nat(Pxx, eta_x)


def mnw_elog_like_X_given_pY(st, Py, etay):
    """ref transforms/MatrixNormalWishart.py:263-289 (bias enters with the opposite sign to backward)."""
    Jyy, Jyx, Jxx, jy, jx, J11, e = _joint_blocks(st, Py, etay, -1.0)
    Pyy, nBiD, nCiA, Pxx = precision_marginalizer(Jyy, Jyx, T(Jyx), Jxx)
    eta_y = jy + nBiD @ jx
    eta_x = jx + nCiA @ jy
    Sxx = inv(Pxx)
    n = Py.shape[-1]
    out = _res_nat(Py, etay) + 0.5 * _sq(T(eta_y) @ inv(Pyy) @ eta_y) - 0.5 * torch.logdet(Pyy)
    out = out + 0.5 * n * LOG2PI + 0.5 * e["ElogdetinvSigma"] - 0.5 * J11
    return Pxx, eta_x, Sxx @ eta_x, Sxx, out - _res_nat(Pxx, eta_x)
