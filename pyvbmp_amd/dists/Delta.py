"""Observed tensor dressed as a distribution so that transforms can ask it for expectations
(surface of the reference's dists/Delta.py:6-51).  A point mass: every expectation is the function of the value
itself, so the expectation methods are generated from one table of value -> statistic maps."""
import torch

# name of the reference's method -> statistic of the observed value x (vector format: trailing (dim, 1))
_STATISTICS = {
    "mean": lambda x: x,
    "EX": lambda x: x,
    "EXXT": lambda x: x @ x.transpose(-1, -2),
    "EXTX": lambda x: x.transpose(-1, -2) @ x,
    "EXX": lambda x: x ** 2,
    "ElogX": torch.log,
}


class Delta():
    def __init__(self, X):
        self.X = X

    @property
    def shape(self):
        return self.X.shape

    def _rewrap(self, op, *args):
        return Delta(getattr(self.X, op)(*args))

    def unsqueeze(self, dim):
        return self._rewrap("unsqueeze", dim)

    def squeeze(self, dim):
        return self._rewrap("squeeze", dim)

    def sum(self, dim, keepdim=False):
        return self.X.sum(dim, keepdim=keepdim)

    def cumsum(self, dim):
        return self.X.cumsum(dim)

    def ESigma(self):
        """no covariance: None lets the moment kernels skip the covariance pass for observed data"""
        return None

    def EXTAX(self, A):
        return self.X.transpose(-1, -2) @ A @ self.X

    def E(self, f):
        return f(self.X)


def _install(name, stat):
    def method(self):
        return stat(self.X)
    method.__name__ = name
    setattr(Delta, name, method)


for _name, _stat in _STATISTICS.items():
    _install(_name, _stat)
