"""Small log-space helpers (surface of the reference's utils/torch_functions.py:2-22).

The reference hand-rolls the max-shifted log-sum-exp; `torch.logsumexp` already is that, fused, so the two "stable"
helpers are thin names over it.  The multivariate (di)gamma helpers share one half-integer ladder.
"""
import math

import torch


def stable_logsumexp(x, dims, keepdim=False):
    """log sum exp over `dims` without overflow"""
    return torch.logsumexp(x, dims, keepdim=keepdim)


def stable_softmax(x, dims):
    """LOG of the softmax over `dims` (the reference's function of this name returns log-probabilities, :8-9)"""
    return x - stable_logsumexp(x, dims, keepdim=True)


def logmatmulexp(x, y):
    """log(exp(x) @ exp(y)) with the row maxima of x and the column maxima of y taken out first"""
    row_max, col_max = x.amax(-1, keepdim=True), y.amax(-2, keepdim=True)
    return torch.log(torch.exp(x - row_max) @ torch.exp(y - col_max)) + row_max + col_max


def _ladder(nu, dim, fn):
    """sum_{i<dim} fn(nu - i/2)"""
    steps = 0.5 * torch.arange(dim, device=nu.device, dtype=nu.dtype)
    return fn(nu.unsqueeze(-1) - steps).sum(-1)


def log_mvgamma(nu, dim):
    """log of the multivariate gamma function Gamma_dim(nu)"""
    return _ladder(nu, dim, torch.lgamma) + 0.25 * dim * (dim - 1) * math.log(math.pi)


mvgammaln = log_mvgamma


def mvdigamma(nu, dim):
    """derivative of log_mvgamma with respect to nu"""
    return _ladder(nu, dim, torch.digamma)
