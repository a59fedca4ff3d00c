"""Oracle: Wishart and Normal-inverse-Wishart conjugate updates (torch CPU, LU route).

State is a plain dict of tensors; every function is pure (returns a new dict).
Reference: dists/Wishart.py, dists/NormalInverseWishart.py.
"""
import math

import torch

LOG2 = math.log(2.0)
LOG2PI = math.log(2.0 * math.pi)


def _bc(v, k):
    """append k singleton axes"""
    return v.reshape(v.shape + (1,) * k)


# ----------------------------------------------------------------------------------- Wishart
def wishart_new(event_shape, batch_shape=(), scale=1.0, dtype=torch.float64):
    """Prior/posterior right after construction.  ref dists/Wishart.py:9-26."""
    D = event_shape[-1]
    invU_0 = (scale ** 2 * torch.eye(D, dtype=dtype)).expand(tuple(batch_shape) + tuple(event_shape))
    nu_0 = torch.tensor(D + 2.0, dtype=dtype).expand(tuple(batch_shape) + tuple(event_shape[:-2]))
    return {
        "dim": D, "event_dim": len(event_shape), "batch_dim": len(batch_shape),
        "invU_0": invU_0, "nu_0": nu_0, "logdet_invU_0": torch.logdet(invU_0),
        "invU": invU_0, "U": torch.linalg.inv(invU_0), "nu": nu_0, "logdet_invU": torch.logdet(invU_0),
        "acc_SExx": 0.0, "acc_N": 0.0,
    }


def wishart_ss_update(st, SExx, N, lr=1.0, beta=None):
    """ref dists/Wishart.py:43-56: optional forgetting accumulator, lr blend, inverse + logdet."""
    st = dict(st)
    if beta is not None:
        st["acc_SExx"] = SExx + beta * st["acc_SExx"]
        st["acc_N"] = N + beta * st["acc_N"]
        SExx, N = st["acc_SExx"], st["acc_N"]
    st["invU"] = lr * (st["invU_0"] + SExx) + (1.0 - lr) * st["invU"]
    st["nu"] = lr * (st["nu_0"] + N) + (1.0 - lr) * st["nu"]
    st["U"] = torch.linalg.inv(st["invU"])
    st["logdet_invU"] = torch.logdet(st["invU"])
    return st


def _mv_lgamma(x, D):
    """sum_i lgamma(x - i/2), i<D.  ref dists/Wishart.py:37-38"""
    return torch.lgamma(x.unsqueeze(-1) - torch.arange(D, dtype=x.dtype) / 2.0).sum(-1)


def _mv_digamma(x, D):
    """ref dists/Wishart.py:40-41"""
    return torch.digamma(x.unsqueeze(-1) - torch.arange(D, dtype=x.dtype) / 2.0).sum(-1)


def wishart_expectations(st):
    """ref dists/Wishart.py:67-97."""
    D = st["dim"]
    nu2 = _bc(st["nu"], 2)
    out = {
        "mean": st["U"] * nu2,
        "meaninv": st["invU"] / (nu2 - D - 1),
        "ESigma": st["invU"] / (nu2 - D - 1),
        "EinvSigma": st["U"] * nu2,
        "invEinvSigma": st["invU"] / nu2,
        "ElogdetinvSigma": D * LOG2 - st["logdet_invU"] + _mv_digamma(st["nu"] / 2.0, D),
        "logdetEinvSigma": -st["logdet_invU"] + st["nu"].log(),
        "logZ": _mv_lgamma(st["nu"] / 2.0, D) + 0.5 * st["nu"] * D * LOG2 - 0.5 * st["nu"] * st["logdet_invU"],
    }
    out["KLqprior"] = wishart_kl(st)
    return out


def wishart_kl(st, event_dim=None):
    """KL(q || prior) summed over the event dims beyond the trailing (D, D).  ref dists/Wishart.py:88-94."""
    D = st["dim"]
    ed = st["event_dim"] if event_dim is None else event_dim
    nu, nu_0 = st["nu"], st["nu_0"]
    tr = (st["invU_0"] * st["U"]).sum((-1, -2))
    kl = nu_0 / 2.0 * (st["logdet_invU"] - st["logdet_invU_0"]) + nu / 2.0 * tr - nu * D / 2.0
    kl = kl + _mv_lgamma(nu_0 / 2.0, D) - _mv_lgamma(nu / 2.0, D) + (nu - nu_0) / 2.0 * _mv_digamma(nu / 2.0, D)
    for _ in range(ed - 2):
        kl = kl.sum(-1)
    return kl


# --------------------------------------------------------------------------------------- NIW
def niw_new(event_shape, batch_shape=(), scale=1.0, mu_init=None, lambda_mu_0=1.0, mu_0=0.0, nu_0=None, invU_0=None,
            fixed_precision=False, dtype=torch.float64):
    """ref dists/NormalInverseWishart.py:6-37.  The random initial mean (:22) must be supplied (mu_init)."""
    D = event_shape[-1]
    bs, es = tuple(batch_shape), tuple(event_shape)
    lam0 = torch.as_tensor(lambda_mu_0, dtype=dtype).expand(bs + (len(es) - 1) * (1,))
    m0 = torch.as_tensor(mu_0, dtype=dtype).expand(bs + es)
    w = wishart_new(es + (D,), bs, scale, dtype)
    if invU_0 is not None and nu_0 is not None:  # :25-33 (shape mismatches fall back to the default)
        if w["invU_0"].shape == invU_0.shape:
            w["invU_0"] = invU_0
        if w["nu_0"].shape == nu_0.shape:
            w["nu_0"] = nu_0
    return {
        "dim": D, "event_dim": len(es), "batch_dim": len(bs), "fixed_precision": fixed_precision,
        "lambda_mu_0": lam0, "mu_0": m0, "lambda_mu": lam0, "mu": m0 if mu_init is None else mu_init,
        "W": w, "acc_SExx": 0.0, "acc_SEx": 0.0, "acc_N": 0.0,
    }


def niw_ss_update(st, SExx, SEx, N, lr=1.0, beta=0.0):
    """ref dists/NormalInverseWishart.py:49-68 (the headline op, config 2)."""
    st = dict(st)
    if beta is not None:
        st["acc_SExx"] = beta * st["acc_SExx"] + SExx
        st["acc_SEx"] = beta * st["acc_SEx"] + SEx
        st["acc_N"] = beta * st["acc_N"] + N
        SExx, SEx, N = st["acc_SExx"], st["acc_SEx"], st["acc_N"]
    lam0, m0 = st["lambda_mu_0"], st["mu_0"]
    lam = lam0 + N
    mu = (lam0.unsqueeze(-1) * m0 + SEx) / lam.unsqueeze(-1)
    arg = SExx + _bc(lam0, 2) * m0.unsqueeze(-1) * m0.unsqueeze(-2) - _bc(lam, 2) * mu.unsqueeze(-1) * mu.unsqueeze(-2)
    st["lambda_mu"] = lr * lam + (1 - lr) * st["lambda_mu"]
    st["mu"] = lr * mu + (1 - lr) * st["mu"]
    if not st["fixed_precision"]:
        st["W"] = wishart_ss_update(st["W"], arg, N, lr)
    return st


def niw_raw_moments(X, p, batch_shape, event_shape):
    """Weighted sufficient statistics.  ref dists/NormalInverseWishart.py:70-84."""
    bd, ed = len(batch_shape), len(event_shape)
    sample_shape = X.shape[: X.ndim - ed - bd]
    sdims = tuple(range(len(sample_shape)))
    if p is None:
        SEx = X.sum(sdims)
        SExx = (X.unsqueeze(-1) * X.unsqueeze(-2)).sum(sdims)
        n = 1
        for s in sample_shape:
            n *= s
        N = torch.tensor(float(n), dtype=X.dtype).expand(tuple(batch_shape) + tuple(event_shape[:-1]))
    else:
        N = p.sum(sdims)
        N = N.reshape(N.shape + (1,) * (ed - 1))
        pv = p.reshape(p.shape + (1,) * ed)
        SExx = (X.unsqueeze(-1) * X.unsqueeze(-2) * pv.unsqueeze(-1)).sum(sdims)
        SEx = (X * pv).sum(sdims)
    return SExx, SEx, N


def niw_raw_update(st, X, p, batch_shape, event_shape, lr=1.0, beta=None):
    SExx, SEx, N = niw_raw_moments(X, p, batch_shape, event_shape)
    return niw_ss_update(st, SExx, SEx, N, lr, beta)


def niw_expectations(st):
    """ref dists/NormalInverseWishart.py:107-132."""
    D, mu, lam = st["dim"], st["mu"], st["lambda_mu"]
    we = wishart_expectations(st["W"])
    P = we["EinvSigma"]
    Pmu = (P * mu.unsqueeze(-2)).sum(-1)
    return {
        "mean": mu, "EX": mu,
        "EXXT": mu.unsqueeze(-1) * mu.unsqueeze(-2) + we["ESigma"] / _bc(lam, 2),
        "ESigma": we["ESigma"], "ElogdetinvSigma": we["ElogdetinvSigma"], "EinvSigma": P,
        "EinvSigmamu": Pmu, "EinvUX": Pmu,
        "EXTinvUX": (mu.unsqueeze(-1) * P * mu.unsqueeze(-2)).sum((-1, -2)) + D / lam,
    }


def niw_elog_like(st, X, event_dim=None):
    """Expected log-likelihood; sums over extra event dims.  ref dists/NormalInverseWishart.py:91-97."""
    D = st["dim"]
    ed = st["event_dim"] if event_dim is None else event_dim
    e = niw_expectations(st)
    out = -0.5 * ((X.unsqueeze(-1) * e["EinvSigma"]).sum(-2) * X).sum(-1) + (X * e["EinvSigmamu"]).sum(-1) - 0.5 * e["EXTinvUX"]
    out = out + 0.5 * e["ElogdetinvSigma"] - 0.5 * D * LOG2PI
    for _ in range(ed - 1):
        out = out.sum(-1)
    return out


def niw_kl(st, event_dim=None):
    """ref dists/NormalInverseWishart.py:99-105."""
    D = st["dim"]
    ed = st["event_dim"] if event_dim is None else event_dim
    lam0, lam = st["lambda_mu_0"], st["lambda_mu"]
    d = st["mu"] - st["mu_0"]
    Wmean = wishart_expectations(st["W"])["mean"]
    kl = 0.5 * (lam0 / lam - 1 + (lam / lam0).log()) * D
    kl = kl + 0.5 * lam0 * (d.unsqueeze(-1) * d.unsqueeze(-2) * Wmean).sum((-1, -2))
    for _ in range(ed - 1):
        kl = kl.sum(-1)
    return kl + wishart_kl(st["W"], ed + 1)
