"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch, fp64, dense broadcasting exactly as the reference forms it) of
the Polya-Gamma logistic-regression gate and its ARD coefficient posterior:
    dists/MVN_ard.py:23-113, transforms/MultiNomialLogisticRegression.py:5-300.
Functional style: states are dicts, nothing is mutated in place.  Pinned to the reference by tests/golden/dmix.npz
(tools/gen_golden.py, group "dmix").  Never imported by pyvbmp_amd/."""
import math

import torch

LOG2 = math.log(2.0)


def T_(x):
    return x.transpose(-2, -1)


# ------------------------------------------------------------------------------------------ MVN_ard
def ard_from(mu, invSigma, invSigmamu, Sigma, logdetinvSigma, alpha, beta, scale=1.0):
    """state captured from a fixture (the constructor draws from the global RNG, ref dists/MVN_ard.py:32,38)"""
    full = tuple(mu.shape)
    return {"mu": mu, "invSigma": invSigma, "invSigmamu": invSigmamu, "Sigma": Sigma, "logdetinvSigma": logdetinvSigma,
            "alpha": alpha, "beta": beta, "dim": mu.shape[-2],
            "alpha_0": torch.full(full, 0.5, dtype=mu.dtype), "beta_0": torch.full(full, 0.5 * scale ** 2, dtype=mu.dtype)}


def _gamma_mean(st):
    return st["alpha"] / st["beta"]


def ard_ss_update(st, SExx, SEx, iters=2, lr=1.0):
    """ref dists/MVN_ard.py:48-72 (beta=None).  NB :58: the first mean uses the PREVIOUS natural mean."""
    st = dict(st)
    eye = torch.eye(st["dim"], dtype=SExx.dtype)
    invSigmamu = SEx
    invSigma = SExx + _gamma_mean(st) * eye + 1e-6 * eye
    Sigma = torch.linalg.inv(invSigma)
    mu = Sigma @ st["invSigmamu"]
    for _ in range(iters):
        EXXT = Sigma.diagonal(dim1=-1, dim2=-2).unsqueeze(-1) + mu ** 2
        # Gamma.ss_update(SElogx = 1/2, SEx = EXXT/2), ref dists/Gamma.py:34-46
        st["alpha"] = (st["alpha_0"] + 0.5) * lr + st["alpha"] * (1 - lr)
        st["beta"] = (st["beta_0"] + 0.5 * EXXT) * lr + st["beta"] * (1 - lr)
        invSigma = SExx + _gamma_mean(st) * eye
        Sigma = torch.linalg.inv(invSigma)
        mu = Sigma @ invSigmamu
    st["invSigma"] = (1 - lr) * st["invSigma"] + lr * invSigma
    st["invSigmamu"] = (1 - lr) * st["invSigmamu"] + lr * invSigmamu
    st["Sigma"] = torch.linalg.inv(st["invSigma"])
    st["mu"] = st["Sigma"] @ st["invSigmamu"]
    st["logdetinvSigma"] = torch.logdet(st["invSigma"])
    return st


def ard_exxt(st):
    return st["Sigma"] + st["mu"] @ T_(st["mu"])


def ard_kl(st):
    """ref dists/MVN_ard.py:74-79 with Gamma.KLqprior (dists/Gamma.py:103-105); event = (n, p, 1)"""
    a, b, a0, b0 = st["alpha"], st["beta"], st["alpha_0"], st["beta_0"]
    am = a / b
    KL = 0.5 * (st["mu"].pow(2) * am).sum((-3, -2, -1))
    KL = KL - 0.5 * (a.log() - b.log()).sum((-3, -2, -1)) + 0.5 * st["logdetinvSigma"].sum(-1)
    KL = KL + (st["Sigma"].diagonal(dim1=-1, dim2=-2) * am.squeeze(-1)).sum((-2, -1))
    kg = (a - a0) * a.digamma() - a.lgamma() + a0.lgamma() + a0 * (b.log() - b0.log()) + a * (b0 / b - 1)
    return KL + kg.sum((-3, -2, -1))


# ------------------------------------------------------------------------------------------ the gate
def stick(Y):
    N = Y.sum(-1, True) - (Y.cumsum(-1) - Y)
    return N[..., :-1], (Y - N / 2.0)[..., :-1]


def pad(X):
    return torch.cat((X, torch.ones(X.shape[:-1] + (1,), dtype=X.dtype)), -1)


def _ew(pgb, pgc):
    return pgb / 2.0 / pgc * (pgc / 2.0).tanh()


def mnlr_raw_update(st, X, Y, iters=2, p=None, lr=1.0, pad_X=True):
    """ref transforms/MultiNomialLogisticRegression.py:43-80 (one sample axis, no batch)"""
    pgb, YmN = stick(Y)
    EX = pad(X) if pad_X else X
    EX = EX.reshape(EX.shape[:-1] + (1,) + EX.shape[-1:] + (1,))  # (S, 1, p, 1)
    EXXT = EX * T_(EX)
    w = 1.0 if p is None else p.reshape(p.shape + (1, 1, 1))
    SEyx = (YmN.reshape(YmN.shape + (1, 1)) * EX * w).sum(0)
    for _ in range(iters):
        pgc = (ard_exxt(st) * EXXT).sum(-1).sum(-1).sqrt()
        Ew = _ew(pgb, pgc).unsqueeze(-1).unsqueeze(-1)
        SExx = (Ew * EXXT * w).sum(0)
        st = ard_ss_update(st, SExx, SEyx, lr=lr)
    return st


def _padded_moments(mu, Sigma, pad_X):
    EX, EXXT = mu, Sigma + mu @ T_(mu)
    if pad_X:
        EXXT = torch.cat((EXXT, EX), -1)
        EX = torch.cat((EX, torch.ones(EX.shape[:-2] + (1, 1), dtype=EX.dtype)), -2)
        EXXT = torch.cat((EXXT, T_(EX)), -2)
    return EX, EXXT


def mnlr_update(st, mu, Sigma, pY, iters=2, p=None, lr=1.0, pad_X=True):
    """ref :82-118; input distribution given by its mean (S,p0,1) and covariance (S,p0,p0)"""
    pgb, YmN = stick(pY)
    EX, EXXT = _padded_moments(mu, Sigma, pad_X)
    EX, EXXT = EX.unsqueeze(-3), EXXT.unsqueeze(-3)
    w = 1.0 if p is None else p.reshape(p.shape + (1, 1, 1))
    SEyx = (YmN.reshape(YmN.shape + (1, 1)) * EX * w).sum(0)
    for _ in range(iters):
        pgc = (ard_exxt(st) * EXXT).sum(-1).sum(-1).sqrt()
        Ew = _ew(pgb, pgc).unsqueeze(-1).unsqueeze(-1)
        SExx = (Ew * EXXT * w).sum(0)
        st = ard_ss_update(st, SExx, SEyx, lr=lr)
    return st


def mnlr_elog_like(st, X, Y, pad_X=True):
    """ref :176-192"""
    pgb, YmN = stick(Y)
    X = pad(X) if pad_X else X
    X = X.unsqueeze(-2)
    SEyxb = (YmN.unsqueeze(-1) * X * st["mu"].squeeze(-1)).sum(-1)
    Xc = X.unsqueeze(-1)
    pgc = (Xc * (ard_exxt(st) @ Xc)).sum(-2).squeeze(-1).sqrt()
    return SEyxb.sum(-1) - (pgb * (0.5 * pgc).cosh().log()).sum(-1) - pgb.sum(-1) * LOG2


def mnlr_elog_like_dist(st, mu, Sigma, Y, pad_X=True):
    """ref :157-174"""
    EX, EXXT = _padded_moments(mu, Sigma, pad_X)
    pgb, YmN = stick(Y)
    EX, EXXT = EX.unsqueeze(-3), EXXT.unsqueeze(-3)
    SEyxb = (YmN.unsqueeze(-1) * EX.squeeze(-1) * st["mu"].squeeze(-1)).sum(-1)
    pgc = (EXXT * ard_exxt(st)).sum(-1).sum(-1).sqrt()
    return SEyxb.sum(-1) - (pgb * (0.5 * pgc).cosh().log()).sum(-1) - pgb.sum(-1) * LOG2


def _targets(n_classes, sample_ndim, dtype):
    Yt = torch.eye(n_classes, dtype=dtype)
    for _ in range(sample_ndim):
        Yt = Yt.unsqueeze(-2)
    return Yt


def mnlr_log_predict(st, X, pad_X=True):
    """ref :229-235"""
    n1 = st["mu"].shape[-3] + 1
    return mnlr_elog_like(st, X, _targets(n1, X.ndim - 1, X.dtype), pad_X).movedim(0, -1)


def mnlr_log_forward(st, mu, Sigma, pad_X=True):
    n1 = st["mu"].shape[-3] + 1
    return mnlr_elog_like_dist(st, mu, Sigma, _targets(n1, mu.ndim - 2, mu.dtype), pad_X).movedim(0, -1)


def mnlr_log_predict_1(st, X, pad_X=True):
    """ref :268-280"""
    X = pad(X) if pad_X else X
    lnpsb = X @ T_(st["mu"].squeeze(-1))
    Xc = X.unsqueeze(-1).unsqueeze(-3)
    pgc = (Xc * (ard_exxt(st) @ Xc)).sum(-2).squeeze(-1).sqrt()
    lnN = -(0.5 * pgc).cosh().log() - LOG2
    ln0 = -0.5 * lnpsb.sum(-1, True) + lnN.sum(-1, True)
    return torch.cat((lnpsb - 0.5 * lnpsb.cumsum(-1) + lnN.cumsum(-1), ln0), -1)


def mnlr_log_predict_2(st, X, pad_X=True):
    """ref :241-266"""
    X = pad(X) if pad_X else X
    X = X.unsqueeze(-2)
    psi_bar = (X * st["mu"].squeeze(-1)).sum(-1)
    Xc = X.unsqueeze(-1)
    pgc = (Xc * (ard_exxt(st) @ Xc)).sum(-2).squeeze(-1).sqrt()
    Ew = 0.5 / pgc * (0.5 * pgc).tanh()
    psi_var = (Xc * (st["Sigma"] @ Xc)).sum(-1).sum(-1)
    n1p = 0.5 + psi_bar / psi_var
    n1m = n1p - 1.0
    n2 = Ew + 1.0 / psi_var
    ln = 0.5 * n1p.pow(2) / n2 - 0.5 * n2.log() - 0.5 * psi_bar.pow(2) / psi_var - 0.5 * psi_var.log() - LOG2 \
        + (0.5 * pgc).cosh().log()
    lnm = ln + 0.5 * (n1m.pow(2) - n1p.pow(2)) / n2
    out = torch.zeros(ln.shape[:-1] + (ln.shape[-1] + 1,), dtype=ln.dtype)
    out[..., 1:] = lnm.cumsum(-1)
    out[..., :-1] = out[..., :-1] + ln
    return out


def mnlr_weights(st):
    mu = st["mu"][..., :-1, 0]
    return 2 * mu - mu.cumsum(-2)


def _res_nat(P, eta):
    """MultivariateNormal_vector_format.Res of a natural-parameter message (ref dists/MultivariateNormal_vector_format.py:118-119)"""
    mu = torch.linalg.inv(P) @ eta
    d = P.shape[-1]
    return -0.5 * (mu * eta).sum(-1).sum(-1) + 0.5 * torch.logdet(P) - 0.5 * d * math.log(2 * math.pi)


def mnlr_elog_like_X(st, like_P, like_eta, pY, iters=2, pad_X=True):
    """ref :201-227 (including the empty slice of the pad_X residual, :222)"""
    pgb, YmN = stick(pY)
    BBT, bm = ard_exxt(st), st["mu"]
    pgc = BBT.sum(-1).sum(-1).sqrt()
    Ew = _ew(pgb, pgc)
    v = lambda t: t.reshape(t.shape + (1, 1))  # noqa: E731
    for _ in range(iters):
        if pad_X:
            eta = like_eta + (v(YmN) * bm[..., :-1, -1:] - v(Ew) * BBT[..., :-1, -1:]).sum(-3)
            P = like_P + (v(Ew) * BBT[..., :-1, :-1]).sum(-3)
            Sigma = torch.linalg.inv(P)
            mu = Sigma @ eta
            pgc = ((BBT[..., :-1, :-1] * (Sigma + mu @ T_(mu)).unsqueeze(-3)).sum(-1).sum(-1)
                   + 2 * (BBT[..., -1:, :-1] @ mu.unsqueeze(-3)).squeeze(-1).squeeze(-1) + BBT[..., -1, -1]).sqrt()
        else:
            eta = like_eta + (v(YmN) * bm).sum(-3)
            P = like_P + (v(Ew) * BBT).sum(-3)
            Sigma = torch.linalg.inv(P)
            mu = Sigma @ eta
            pgc = ((BBT * (Sigma + mu @ T_(mu)).unsqueeze(-3)).sum(-1).sum(-1)).sqrt()
        Ew = _ew(pgb, pgc)
    if pad_X:
        Res = -pgb.sum(-1) * LOG2 + (YmN * ((bm[..., -1:, :-1] * mu.unsqueeze(-3)).sum(-1).sum(-1) + bm[..., -1, -1])).sum(-1)
    else:
        Res = -pgb.sum(-1) * LOG2 + (YmN * ((bm * mu.unsqueeze(-3)).sum(-1).sum(-1))).sum(-1)
    Res = Res - (pgb * (0.5 * pgc).cosh().log()).sum(-1) + _res_nat(like_P, like_eta)
    return P, eta, Sigma, mu, Res
