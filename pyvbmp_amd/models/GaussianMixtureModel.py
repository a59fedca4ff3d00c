"""Gaussian mixture = Mixture over a batch of NormalInverseWishart components
(surface of the reference's models/GaussianMixtureModel.py:6-16)."""
import torch

from ..dists.Mixture import Mixture
from ..dists.NormalInverseWishart import NormalInverseWishart


class GaussianMixtureModel(Mixture):
    def __init__(self, nc, dim, isotropic=False, device=None, dtype=None):
        if isotropic:
            raise NotImplementedError("isotropic=True uses NormalGamma, which is outside the accelerated path")
        dist = NormalInverseWishart(event_shape=(dim,), batch_shape=(nc,), scale=1.0 / nc ** (1.0 / dim),
                                    device=device, dtype=dtype)
        super().__init__(dist, event_shape=(nc,))

    def initialize(self, data):
        idx = torch.randint(data.shape[0], self.event_shape, device=data.device)
        self.dist.mu = data[idx, :]
