"""Oracle: Dirichlet weights + mixture E-step / M-step glue (torch CPU).  TEST INFRASTRUCTURE ONLY.

Reference: dists/Dirichlet.py, dists/Mixture.py, models/GaussianMixtureModel.py.
"""
import torch

from . import niw as _niw


def dirichlet_loggeomean(alpha, event_dim=1):
    """E[log pi].  ref dists/Dirichlet.py:52-53"""
    dims = tuple(range(-event_dim, 0))
    return torch.digamma(alpha) - torch.digamma(alpha.sum(dims, keepdim=True))


def dirichlet_ss_update(alpha_0, alpha, NA, lr=1.0):
    """ref dists/Dirichlet.py:22-28 (beta=None)"""
    return lr * (NA + alpha_0) + (1 - lr) * alpha


def _lg0(x):
    y = torch.lgamma(x)
    return torch.where(torch.isinf(y) & (y > 0), torch.zeros_like(y), y)


def _dg0(x):
    y = torch.digamma(x)
    return torch.where(torch.isinf(y) & (y < 0), torch.zeros_like(y), y)


def dirichlet_kl(alpha_0, alpha, event_dim=1, batch_dim=0):
    """ref dists/Dirichlet.py:73-83"""
    dims = tuple(range(-event_dim, 0))
    a_sum, a0_sum = alpha.sum(dims), alpha_0.sum(dims)
    kl = torch.lgamma(a_sum) - _lg0(alpha).sum(dims) - torch.lgamma(a0_sum) + _lg0(alpha_0).sum(dims)
    kl = kl + ((alpha - alpha_0) * (_dg0(alpha) - torch.digamma(a_sum).reshape(a_sum.shape + (1,) * event_dim))).sum(dims)
    while kl.ndim > batch_dim:
        kl = kl.sum(-1)
    return kl


def logsumexp_last(x, ndims):
    """max-shifted log-sum-exp over the last ndims axes.  ref dists/Mixture.py:102-121"""
    dims = tuple(range(-ndims, 0))
    m = x.amax(dims, keepdim=True)
    return (m + (x - m).exp().sum(dims, keepdim=True).log()).reshape(x.shape[: x.ndim - ndims])


def mixture_estep(niw_state, alpha, X, mix_event_dim=1, mix_batch_dim=0, dist_event_dim=1):
    """Responsibilities.  ref dists/Mixture.py:38-45, :68-70.

    X: sample + mix_batch + dist_event  ->  viewed as sample + mix_batch + (1,)*mix_event_dim + dist_event.
    Returns p, NA, logZ.
    """
    Xv = X.reshape(X.shape[: X.ndim - dist_event_dim] + (1,) * mix_event_dim + X.shape[X.ndim - dist_event_dim:])
    log_p = _niw.niw_elog_like(niw_state, Xv, dist_event_dim) + dirichlet_loggeomean(alpha, mix_event_dim)
    lse = logsumexp_last(log_p, mix_event_dim)
    p = (log_p - lse.reshape(lse.shape + (1,) * mix_event_dim)).exp()
    sample_dim = p.ndim - mix_batch_dim - mix_event_dim
    sd = tuple(range(sample_dim))
    return p, p.sum(sd), lse.sum(sd)


def mixture_kl(niw_state, alpha_0, alpha, mix_event_dim=1, mix_batch_dim=0, dist_event_dim=1):
    """ref dists/Mixture.py:71-72"""
    dims = tuple(range(-mix_event_dim, 0))
    return _niw.niw_kl(niw_state, dist_event_dim).sum(dims) + dirichlet_kl(alpha_0, alpha, mix_event_dim, mix_batch_dim)


def mixture_iteration(niw_state, alpha_0, alpha, X, lr=1.0, mix_event_shape=None, mix_batch_shape=(),
                      dist_event_shape=None):
    """One Mixture.update pass: E-step, ELBO, M-step.  ref dists/Mixture.py:54-66."""
    med, mbd, ded = len(mix_event_shape), len(mix_batch_shape), len(dist_event_shape)
    p, NA, logZ = mixture_estep(niw_state, alpha, X, med, mbd, ded)
    elbo = logZ - mixture_kl(niw_state, alpha_0, alpha, med, mbd, ded)
    alpha = dirichlet_ss_update(alpha_0, alpha, NA, lr)
    Xv = X.reshape(X.shape[: X.ndim - ded] + (1,) * med + X.shape[X.ndim - ded:])
    niw_state = _niw.niw_raw_update(niw_state, Xv, p, tuple(mix_batch_shape) + tuple(mix_event_shape), dist_event_shape, lr, None)
    return niw_state, alpha, {"p": p, "NA": NA, "logZ": logZ, "ELBO": elbo}
