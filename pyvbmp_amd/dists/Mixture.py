"""Mixture over the trailing batch dims of any distribution node (surface of the reference's
dists/Mixture.py:5-127).

E-step: when the component node offers `mixture_estep_params()` (NormalInverseWishart does) and the
mixture is a plain K-component one, responsibilities, NA and logZ come from ONE fused HIP kernel
(K3: quadratic form + log-sum-exp + normalisation + reductions; nothing of size (N,K,D,D) exists).
Otherwise the generic composition Elog_like (K3a) + log-sum-exp is used, exactly like the reference.
M-step: dist.raw_update (K4 moments + K2 update).
"""
import torch

from .. import ops
from .Dirichlet import Dirichlet


class Mixture():
    def __init__(self, dist, event_shape, prior_parms=None):
        event_shape = tuple(event_shape)
        assert tuple(dist.batch_shape[-len(event_shape):]) == event_shape
        self.event_shape = event_shape
        self.event_dim = len(event_shape)
        self.batch_shape = tuple(dist.batch_shape[:-len(event_shape)])
        self.batch_dim = len(self.batch_shape)
        self.device, self.dtype = dist.device, dist.dtype
        self.pi = Dirichlet(event_shape=event_shape, batch_shape=self.batch_shape, prior_parms=prior_parms,
                            device=self.device, dtype=self.dtype)
        self.dist = dist
        # set to a pyvbmp_amd.parallel.SuffStatReducer when the SAMPLE axis is sharded over ranks: the
        # statistics of one VB iteration (NA, logZ, N, SEx, SExx) then cross ranks in one all-reduce
        self.reducer = None
        self.logZ = torch.tensor(-torch.inf, device=self.device, dtype=self.dtype)
        self.ELBO_last = torch.tensor(-torch.inf, device=self.device, dtype=self.dtype)

    def to_event(self, n):
        if n == 0:
            return self
        self.event_dim = self.event_dim + n
        self.event_shape = self.batch_shape[-n:] + self.event_shape
        self.batch_shape = self.batch_shape[:-n]
        self.pi.to_event(n)
        self.dist.to_event(n)
        return self

    def _fusable(self, X):
        d = self.dist
        return (hasattr(d, "mixture_estep_params") and self.batch_dim == 0 and self.event_dim == 1
                and d.event_dim == 1 and d.batch_dim == 1 and X.shape[-1] == d.dim)

    def update_assignments(self, X):
        if self._fusable(X):
            # the component expectations AND E log pi in one launch (K13) when the mixing weights are a plain Dirichlet
            alpha = getattr(self.pi, "alpha", None)
            if alpha is not None and alpha.ndim == 1:
                P, b, c = self.dist.mixture_estep_params(alpha=alpha)
            else:
                P, b, c = self.dist.mixture_estep_params()
                c = c + self.pi.loggeomean()
            sample_shape = tuple(X.shape[:-1])
            p, NA, logZ = ops.mixture_estep(X.reshape(-1, X.shape[-1]), P, b, c)
            self.p = p.reshape(sample_shape + self.event_shape)
            self.NA = NA
            self.logZ = logZ
            return
        log_p = self.Elog_like(X)
        ev = tuple(range(-self.event_dim, 0))
        logZ = torch.logsumexp(log_p, ev)
        self.p = (log_p - logZ.reshape(tuple(logZ.shape) + self.event_dim * (1,))).exp()
        sd = tuple(range(self.p.ndim - self.batch_dim - self.event_dim))
        self.NA = self.p.sum(sd)
        self.logZ = logZ.sum(sd)

    def update_parms(self, X, lr=1.0):
        self.pi.ss_update(self.NA, lr=lr)
        self.update_dist(X, lr=lr)

    def raw_update(self, X, iters=1, lr=1.0, verbose=False):
        self.update(X, iters=iters, lr=lr, verbose=verbose)

    def _update_sharded(self, X, lr):
        """One VB iteration when each rank holds a slice of the samples (case 2)."""
        self.update_assignments(X)  # local responsibilities, local NA / logZ
        SExx, SEx, N = self.dist.raw_moments(self._view(X), self.p)
        self.NA, self.logZ, N, SEx, SExx = self.reducer.all_reduce([self.NA, self.logZ, N, SEx, SExx])
        ELBO = self.ELBO()
        self.pi.ss_update(self.NA, lr=lr)
        self.dist.ss_update(SExx, SEx, N, lr, None)
        return ELBO

    def update(self, X, iters=1, lr=1.0, verbose=False, graphed=False):
        """VB iterations (ref dists/Mixture.py:47-58).  graphed=True replays the iteration as one HIP graph
        (pyvbmp_amd.graph): for small, launch-bound problems; not with verbose (printing synchronises) nor with a
        multi-rank reducer (collectives stay outside graphs here)."""
        if graphed and not verbose and (self.reducer is None or self.reducer.world_size == 1):
            from .. import graph
            key = (X.data_ptr(), tuple(X.shape), X.dtype, float(lr))
            graph.run_iterations(self, lambda m: m.update(X, iters=1, lr=lr), iters, key)
            return
        for i in range(iters):
            if self.reducer is not None:
                ELBO = self._update_sharded(X, lr)
            else:
                self.update_assignments(X)
                ELBO = self.ELBO()
                self.update_parms(X, lr)
            if verbose:
                print('Percent Change in ELBO:   ', (ELBO - self.ELBO_last) / self.ELBO_last.abs() * 100.0)
            self.ELBO_last = ELBO

    def _view(self, X):
        k = X.ndim - self.dist.event_dim
        return X.reshape(tuple(X.shape[:k]) + self.event_dim * (1,) + tuple(self.dist.event_shape))

    def update_dist(self, X, lr):
        self.dist.raw_update(self._view(X), self.p, lr)

    def Elog_like(self, X):
        return self.dist.Elog_like(self._view(X)) + self.pi.loggeomean()

    def KLqprior(self):
        return self.dist.KLqprior().sum(tuple(range(-self.event_dim, 0))) + self.pi.KLqprior()

    def ELBO(self):
        return self.logZ - self.KLqprior()

    def assignment_pr(self):
        return self.p

    def assignment(self):
        return self.p.argmax(-1)

    def means(self):
        return self.dist.mean()

    def event_average_f(self, function_string, A=None, keepdim=False):
        f = getattr(self.dist, function_string)
        return self.event_average(f() if A is None else f(A), keepdim=keepdim)

    def average_f(self, function_string, A=None, keepdim=False):
        f = getattr(self.dist, function_string)
        return self.average(f() if A is None else f(A), keepdim=keepdim)

    def average(self, A, keepdim=False):
        return (A * self.p).sum(-1, keepdim)

    def event_average(self, A, keepdim=False):
        ded = self.dist.event_dim
        out = (A * self.p.reshape(tuple(self.p.shape) + (1,) * ded)).sum(-1 - ded, keepdim)
        for i in range(self.event_dim - 1):
            out = out.sum(-ded - 1, keepdim)
        return out

    def stable_logsumexp(self, x, dim=None, keepdim=False):
        return torch.logsumexp(x, dim, keepdim=keepdim)
