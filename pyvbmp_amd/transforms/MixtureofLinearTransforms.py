"""Mixture of linear transforms: `dim` MatrixNormalWishart (or MatrixNormalGamma) experts gated by a Dirichlet
(surface of the reference's transforms/MixtureofLinearTransforms.py:10-215; SURVEY.md 8(f) row 4, the part that
needs no new arithmetic -- the logistic-regression gate of dMixtureofLinearTransforms is not on this path).

Everything heavy is the MatrixNormalWishart path already built: the experts are ONE transform with batch_shape
(..., dim); the E-step is its joint quadratic form over z = [x; y] for every (sample, expert) followed by the
softmax over experts (K3a + an elementwise pass), and the M-step is
`W.raw_update / W.update` with the responsibilities as weights (K4 moments of z, K5b covariance sums, K1/K2a
update).
"""
import torch

from .. import ops

from ..dists.Dirichlet import Dirichlet
from ..dists.MultivariateNormal_vector_format import MultivariateNormal_vector_format
from .MatrixNormalGamma import MatrixNormalGamma
from .MatrixNormalWishart import MatrixNormalWishart


class MixtureofLinearTransforms():
    def __init__(self, n, p, dim, batch_shape=(), pad_X=True, type='Wishart', device=None, dtype=None):
        self.n, self.p, self.dim = n, p, dim  # dim = number of experts
        self.event_dim = 1
        self.event_shape = (dim,)
        self.batch_shape = tuple(batch_shape)
        self.batch_dim = len(self.batch_shape)
        self.padX = pad_X
        if type == 'Wishart':
            cls = MatrixNormalWishart
        elif type == 'Gamma':
            cls = MatrixNormalGamma
        else:
            raise ValueError('type must be either Wishart (default) or Gamma')
        self.W = cls(event_shape=(n, p), batch_shape=self.batch_shape + (dim,), scale=1.0 / dim ** (1.0 / n),
                     pad_X=pad_X, device=device, dtype=dtype)
        self.device, self.dtype = self.W.device, self.W.dtype
        self.pi = Dirichlet(event_shape=(dim,), batch_shape=self.batch_shape, device=self.device, dtype=self.dtype)
        self.KL_last = self.KLqprior()
        self.ELBO_last = torch.full((), -torch.inf, device=self.device, dtype=self.dtype)

    # ------------------------------------------------------------------ E-step
    def _normalise(self, log_p):
        """responsibilities and per-sample evidence from unnormalised log-probabilities (ref :36-43)"""
        logZ = torch.logsumexp(log_p, -1, keepdim=True)
        self.p = torch.exp(log_p - logZ)
        self.logZ = logZ.squeeze(-1)

    def update_assignments(self, X, Y):
        W = self.W
        n_samples = 1
        for v in X.shape[:-2]:
            n_samples *= v
        if self.batch_dim == 0 and hasattr(W, '_joint_quadratic') and W.event_dim == 2 and X.is_cuda and X.ndim >= 3 and Y.ndim == X.ndim \
                and tuple(X.shape[:-2]) == tuple(Y.shape[:-2]) \
                and ops.estep_sym_serves(n_samples, self.dim, X.shape[-2] + Y.shape[-2], X.dtype):  # (asked before Z, P, b, c are built)
            # unbatched mixture over dense samples: likelihood, softmax and evidence in ONE fused launch (K3, symmetric-packed
            # form) on the stacked vector z = [x; y]; None when the shape is outside that kernel form
            P, b, c = W._joint_quadratic()
            sample = tuple(X.shape[:-2])
            Z = torch.cat((X, Y), dim=-2).reshape(-1, X.shape[-2] + Y.shape[-2])
            hit = ops.mixture_estep(Z, P, b, c + self.pi.loggeomean(), want_lse=True)
            if hit is not None:
                self.p = hit[0].reshape(sample + (self.dim,))
                self.logZ = hit[3].reshape(sample)
                return
        # W.Elog_like is one K3a launch: the joint quadratic form of z = [x; y] for every (sample, expert)
        self._normalise(W.Elog_like(X.unsqueeze(-3), Y.unsqueeze(-3)) + self.pi.loggeomean())

    def update_assignments_given_pX_pY(self, pX, pY):
        self._normalise(self.W.Elog_like_given_pX_pY(pX.unsqueeze(-3), pY.unsqueeze(-3)) + self.pi.loggeomean())

    def Elog_like(self, X, Y):
        self.update_assignments(X, Y)
        ELL = self.logZ
        for i in range(self.event_dim - 1):
            ELL = ELL.sum(-1)
        return ELL

    def Elog_like_given_pX_pY(self, pX, pY):
        ELL = (self.W.Elog_like_given_pX_pY(pX.unsqueeze(-3), pY.unsqueeze(-3)) * self.p).sum(-1)
        for i in range(self.event_dim - 1):
            ELL = ELL.sum(-1)
        return ELL

    # ------------------------------------------------------------------ VB iterations
    def raw_update(self, X, Y, iters=1, lr=1.0, verbose=False):
        for i in range(iters):
            self.update_assignments(X, Y)
            ELBO = self.ELBO()
            self.pi.ss_update(self.p.sum(0), lr=lr)
            self.W.raw_update(X.unsqueeze(-3), Y.unsqueeze(-3), p=self.p, lr=lr)
            if verbose:
                print('MixLinearTransform: Percent Change in ELBO = ',
                      ((ELBO - self.ELBO_last) / self.ELBO_last.abs()).data * 100)
            self.ELBO_last = ELBO

    def update(self, pX, pY, iters=1, lr=1, verbose=False):
        for i in range(iters):
            self.update_assignments_given_pX_pY(pX, pY)
            ELBO = self.ELBO()
            self.pi.ss_update(self.p.sum(0), lr=lr)
            self.W.update(pX.unsqueeze(-3), pY.unsqueeze(-3), p=self.p, lr=lr)
            if verbose:
                print('MixLinearTransform: Percent Change in ELBO = ',
                      ((ELBO - self.ELBO_last) / self.ELBO_last.abs()).data * 100)
            self.ELBO_last = ELBO

    def predict(self, X):
        """moment-matched Gaussian over y and the gate probabilities given x (ref :91-109)"""
        pY, Res = self.W.predict(X.unsqueeze(-3))
        log_p = Res + self.pi.loggeomean()
        p = torch.softmax(log_p, -1)
        pw = p.unsqueeze(-1).unsqueeze(-1)
        m = pY.mean()
        mu = (m * pw).sum(-3)
        Sigma = ((pY.ESigma() + m @ m.transpose(-2, -1)) * pw).sum(-3) - mu @ mu.transpose(-2, -1)
        return MultivariateNormal_vector_format(mu=mu, Sigma=Sigma), p

    def forward(self, pX):
        pass

    def Elog_like_X(self, Y):
        pass

    def backward(self, pY):
        pass

    def KLqprior(self):
        return self.pi.KLqprior() + self.W.KLqprior().sum(-1)

    def ELBO(self):
        logZ = self.logZ.sum(0)
        while logZ.ndim > self.batch_dim:
            logZ = logZ.sum(0)
        return logZ - self.KLqprior()

    def assignment_pr(self):
        return self.p

    def assignment(self):
        return self.p.argmax(-1)

    def mean(self):
        return self.p

    # ------------------------------------------------------------------ responsibility-weighted expectations
    def event_average(self, A):
        """A: mixture batch + (dim,) + W.event_shape -> sample + W.event_shape, averaged with self.p (ref :141-149)"""
        p = self.p.reshape(tuple(self.p.shape) + (1,) * self.W.event_dim)
        out = A * p
        for i in range(self.event_dim):
            out = out.sum(-self.W.event_dim - 1)
        return out

    def average(self, A):
        out = self.p * A
        for i in range(self.event_dim):
            out = out.sum(-1)
        return out

    def weights(self):
        return self.W.mu[..., :-1] if self.padX else self.W.mu

    def bias(self):
        return self.W.mu[..., -1] if self.padX else None

    def means(self):
        return self.W.mu


# responsibility-weighted versions of the experts' expectations (ref :151-197): matrix-valued ones through
# event_average, per-expert scalars through average
def _weighted(name, arity):
    if arity == 0:
        def method(self):
            return self.event_average(getattr(self.W, name)())
    else:
        def method(self, A):
            return self.event_average(getattr(self.W, name)(A))
    method.__name__ = name
    return method


for _name in ("EinvUX", "EXTinvU", "EXTinvUX", "EXinvVXT", "EXmMUTinvUXmMU", "EXmMUinvVXmMUT", "EXTX", "EXXT", "EinvSigma",
              "ESigma"):
    setattr(MixtureofLinearTransforms, _name, _weighted(_name, 0))
for _name in ("EXTAX", "EXAXT"):
    setattr(MixtureofLinearTransforms, _name, _weighted(_name, 1))
MixtureofLinearTransforms.ElogdetinvU = lambda self: self.average(self.W.invU.ElogdetinvSigma())
MixtureofLinearTransforms.ElogdetinvSigma = lambda self: self.average(self.W.ElogdetinvSigma())
