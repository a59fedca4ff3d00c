"""Oracle: multivariate normal parameter conversions and updates (torch CPU, LU route).
TEST INFRASTRUCTURE ONLY.

`vf=True` selects the "vector format" (event = (D,1)); otherwise event = (D,).
Reference: dists/MultivariateNormal.py, dists/MultivariateNormal_vector_format.py.
"""
import math

import torch

LOG2PI = math.log(2.0 * math.pi)


def _col(v, vf):
    return v if vf else v.unsqueeze(-1)


def _uncol(v, vf):
    return v if vf else v.squeeze(-1)


def moments_from_natural(invSigma, invSigmamu, vf=False):
    """(mu, Sigma).  ref MultivariateNormal.py:35-43 / ..._vector_format.py:79-87"""
    Sigma = torch.linalg.inv(invSigma)
    return _uncol(Sigma @ _col(invSigmamu, vf), vf), Sigma


def natural_from_moments(mu, Sigma, vf=False):
    """(invSigmamu, invSigma).  ref MultivariateNormal.py:45-53 / ..._vector_format.py:89-97

    Reference quirk kept for parity: the plain format's EinvSigmamu (:50-53) multiplies the mean by
    EinvSigma().inverse(), i.e. it returns Sigma @ mu, not invSigma @ mu.  The vector format (:94-97) is
    the textbook invSigma @ mu."""
    P = torch.linalg.inv(Sigma)
    if not vf:
        return (torch.linalg.inv(P) * mu.unsqueeze(-2)).sum(-1), P
    return P @ mu, P


def logdet_precision(invSigma=None, Sigma=None):
    """ref MultivariateNormal.py:55-59 / ..._vector_format.py:104-107"""
    return torch.logdet(invSigma) if invSigma is not None else -torch.logdet(Sigma)


def second_moment(mu, Sigma, vf=False):
    """E[x x^T].  ref MultivariateNormal.py:64-65 / ..._vector_format.py:112-113"""
    c = _col(mu, vf)
    return Sigma + c @ c.transpose(-2, -1)


def trace_second_moment(mu, Sigma, vf=False):
    """EXTX.  NB the two formats differ in the reference: plain sums every entry of EXXT
    (MultivariateNormal.py:67-68); vector format sums every entry of Sigma and adds mu^T mu (:115-116)."""
    if vf:
        return Sigma.sum((-1, -2)) + (mu.transpose(-2, -1) @ mu).squeeze(-1).squeeze(-1)
    return second_moment(mu, Sigma, False).sum((-1, -2))


def residual(mu, invSigmamu, logdetinvSigma, D):
    """vector format Res().  ref ..._vector_format.py:118-119"""
    return -0.5 * (mu * invSigmamu).sum((-1, -2)) + 0.5 * logdetinvSigma - 0.5 * D * LOG2PI


def ss_update(SExx, SEx, n, vf=False):
    """moment update.  ref MultivariateNormal.py:70-74 / ..._vector_format.py:121-126"""
    if vf:
        n = n.unsqueeze(-1).unsqueeze(-1)
        mu = SEx / n
        return mu, SExx / n - mu @ mu.transpose(-2, -1)
    mu = SEx / n.unsqueeze(-1)
    return mu, SExx / n.unsqueeze(-1).unsqueeze(-1) - mu.unsqueeze(-1) * mu.unsqueeze(-2)


def raw_update(X, p, batch_shape, vf=False):
    """ref MultivariateNormal.py:76-100 / ..._vector_format.py:128-153.  Returns (mu, Sigma)."""
    ed = 2 if vf else 1
    target = ed + len(batch_shape)
    c = _col(X, vf)
    if p is None:
        SEx, SExx = X, c @ c.transpose(-2, -1)
        sample_shape = X.shape[: X.ndim - target]
        cnt = 1
        for s in sample_shape:
            cnt *= s
        n = torch.tensor(float(cnt), dtype=X.dtype).expand(tuple(batch_shape))
        while SEx.ndim > target:
            SExx, SEx = SExx.sum(0), SEx.sum(0)
        return ss_update(SExx, SEx, n, vf)
    pe = p.reshape(p.shape + (1,) * ed)
    SEx = X * pe
    SExx = (c @ c.transpose(-2, -1)) * (pe if vf else pe.unsqueeze(-1))
    while SEx.ndim > target:
        SExx, SEx, pe = SExx.sum(0), SEx.sum(0), pe.sum(0)
    n = pe.reshape(pe.shape[: pe.ndim - ed])
    return ss_update(SExx, SEx, n, vf)


def elog_like(X, mu, invSigma, logdetinvSigma, vf=False):
    """ref MultivariateNormal.py:103-112 / ..._vector_format.py:156-165 (event_dim at its default)."""
    d = _col(X - mu, vf)
    D = d.shape[-2]
    q = (d.transpose(-2, -1) @ invSigma @ d).squeeze(-1).squeeze(-1)
    return -0.5 * q - 0.5 * D * LOG2PI + 0.5 * logdetinvSigma
