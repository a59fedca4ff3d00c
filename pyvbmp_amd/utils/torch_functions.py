"""Small log-space helpers (surface of the reference's utils/torch_functions.py:2-22)."""
import math

import torch


def stable_logsumexp(x, dims, keepdim=False):
    return torch.logsumexp(x, dims, keepdim=keepdim)


def stable_softmax(x, dims):
    return x - torch.logsumexp(x, dims, keepdim=True)


def logmatmulexp(x, y):
    xs = x.amax(-1, keepdim=True)
    ys = y.amax(-2, keepdim=True)
    return torch.matmul((x - xs).exp(), (y - ys).exp()).log() + xs + ys


def log_mvgamma(nu, dim):
    ar = torch.arange(dim, device=nu.device, dtype=nu.dtype) / 2.0
    return (nu.unsqueeze(-1) - ar).lgamma().sum(-1) + dim * (dim - 1) / 4.0 * math.log(math.pi)


mvgammaln = log_mvgamma


def mvdigamma(nu, dim):
    ar = torch.arange(dim, device=nu.device, dtype=nu.dtype) / 2.0
    return (nu.unsqueeze(-1) - ar).digamma().sum(-1)
