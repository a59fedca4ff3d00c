"""Multivariate normal in "vector format" (event = (dim, 1)): the message type exchanged by the
linear-Gaussian transforms (surface of the reference's dists/MultivariateNormal_vector_format.py:3-168).

Attributes mu / Sigma / invSigmamu / invSigma / logdetinvSigma are plain tensors that callers read,
index-assign and reset to None, exactly as with the reference.  Conversions run K1 (one launch gives
the inverse AND the logdet, which is cached in `logdetinvSigma` so that Res() does not factor again).
Shared arithmetic (moments, raw_update, Elog_like): `_gaussian.GaussianNode`."""
from .. import ops
from ._gaussian import LOG2PI, GaussianNode


class MultivariateNormal_vector_format(GaussianNode):
    _event_axes = 2

    def __init__(self, mu=None, Sigma=None, invSigmamu=None, invSigma=None, logdetinvSigma=None):
        self.mu, self.Sigma, self.invSigmamu, self.invSigma = mu, Sigma, invSigmamu, invSigma
        self.logdetinvSigma = logdetinvSigma
        self._adopt_shapes(mu if mu is not None else invSigmamu)

    @property
    def shape(self):
        return self.batch_shape + self.event_shape

    def to_event(self, n):
        if n != 0:
            self._shift_event(n)
        return self

    def unsqueeze(self, dim):
        assert (dim + self.event_dim < 0)
        parts = [None if t is None else t.unsqueeze(dim) for t in (self.mu, self.Sigma, self.invSigmamu, self.invSigma)]
        return MultivariateNormal_vector_format(*parts).to_event(self.event_dim - 2)

    # ---------------------------------------------------------------- combining messages (natural parameters add)
    def _absorb(self, invSigma, invSigmamu):
        self.invSigma = self.EinvSigma() + invSigma
        self.invSigmamu = self.EinvSigmamu() + invSigmamu
        self.Sigma = self.mu = self.logdetinvSigma = None

    def combiner(self, other):
        self._absorb(other.EinvSigma(), other.EinvSigmamu())

    def nat_combiner(self, invSigma, invSigmamu):
        self._absorb(invSigma, invSigmamu)

    # ---------------------------------------------------------------- conversions: one factorisation serves two attributes
    def _factor(self, of_precision):
        inv, ld = ops.spd_inv_logdet(self.invSigma if of_precision else self.Sigma)
        if of_precision:
            self.Sigma = inv
        else:
            self.invSigma, ld = inv, -ld
        if self.logdetinvSigma is None:
            self.logdetinvSigma = ld

    def mean(self):
        if self.mu is None:
            self.mu = self.ESigma() @ self.invSigmamu
        return self.mu

    def ESigma(self):
        if self.Sigma is None:
            self._factor(True)
        return self.Sigma

    def EinvSigma(self):
        if self.invSigma is None:
            self._factor(False)
        return self.invSigma

    def EinvSigmamu(self):
        if self.invSigmamu is None:
            self.invSigmamu = self.EinvSigma() @ self.mean()
        return self.invSigmamu

    def ElogdetinvSigma(self):
        if self.logdetinvSigma is None:
            self._factor(self.invSigma is not None)
        return self.logdetinvSigma

    def EXTX(self):
        m = self.mean()
        return self.ESigma().sum((-1, -2)) + (m.transpose(-2, -1) @ m).squeeze(-1).squeeze(-1)

    def Res(self):
        return -0.5 * (self.mean() * self.EinvSigmamu()).sum((-1, -2)) + 0.5 * self.ElogdetinvSigma() - 0.5 * self.dim * LOG2PI

    def ss_update(self, SExx, SEx, n, lr=1.0):
        # the moment form (the reference defines ss_update twice; its later definition, :121-126, wins)
        self._set_moments_from_statistics(SExx, SEx, n)
