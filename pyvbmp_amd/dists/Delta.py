"""Observed tensor dressed as a distribution so that transforms can ask it for expectations
(surface of the reference's dists/Delta.py:6-51)."""
import torch


class Delta():
    def __init__(self, X):
        self.X = X

    @property
    def shape(self):
        return self.X.shape

    def unsqueeze(self, dim):
        return Delta(self.X.unsqueeze(dim))

    def squeeze(self, dim):
        return Delta(self.X.squeeze(dim))

    def sum(self, dim, keepdim=False):
        return self.X.sum(dim, keepdim=keepdim)

    def cumsum(self, dim):
        return self.X.cumsum(dim)

    def mean(self):
        return self.X

    def EX(self):
        return self.X

    def ESigma(self):
        """zero covariance (lets the moment kernels treat data and Gaussians uniformly)"""
        return None

    def EXXT(self):
        return self.X @ self.X.transpose(-1, -2)

    def EXTX(self):
        return self.X.transpose(-1, -2) @ self.X

    def EXTAX(self, A):
        return self.X.transpose(-1, -2) @ A @ self.X

    def EXX(self):
        return self.X ** 2

    def ElogX(self):
        return torch.log(self.X)

    def E(self, f):
        return f(self.X)
