"""Multivariate normal with lazily converted moment / natural parameters, event = last axis
(surface of the reference's dists/MultivariateNormal.py:3-115).

Every `.inverse()` / `.logdet()` of the reference is one K1 launch here; the quadratic form of Elog_like is K3a and the
weighted moments of raw_update are K4 (both in `_gaussian.GaussianNode`, shared with the vector-format class).  This class
keeps the reference's conversion semantics to the letter: a conversion stores only the attribute the reference stores."""
from .. import ops
from ._gaussian import GaussianNode


class MultivariateNormal(GaussianNode):
    _event_axes = 1

    def __init__(self, mu=None, Sigma=None, invSigmamu=None, invSigma=None):
        self.mu, self.Sigma, self.invSigmamu, self.invSigma = mu, Sigma, invSigmamu, invSigma
        self._adopt_shapes(mu if mu is not None else invSigmamu)

    def to_event(self, n):
        if n == 0:
            return self
        self._shift_event(n)

    def mean(self):
        if self.mu is None:
            self.mu = self._mv(ops.spd_inverse(self.invSigma), self.invSigmamu)
        return self.mu

    def ESigma(self):
        if self.Sigma is None:
            self.Sigma = ops.spd_inverse(self.invSigma)
        return self.Sigma

    def EinvSigma(self):
        if self.invSigma is None:
            self.invSigma = ops.spd_inverse(self.Sigma)
        return self.invSigma

    def EinvSigmamu(self):
        if self.invSigmamu is None:
            # NB: the reference multiplies the mean by EinvSigma().inverse() here (MultivariateNormal.py:50-53),
            # i.e. by the covariance.  Kept for parity; the vector-format class has the textbook product.
            self.invSigmamu = self._mv(ops.spd_inverse(self.EinvSigma()), self.mean())
        return self.invSigmamu

    def ElogdetinvSigma(self):
        if self.Sigma is None:
            return ops.spd_inv_logdet(self.invSigma)[1]
        return -ops.spd_inv_logdet(self.Sigma)[1]

    def EXTX(self):
        return self.EXXT().sum((-1, -2))

    def ss_update(self, SExx, SEx, n, lr=1.0):
        self._set_moments_from_statistics(SExx, SEx, n)
