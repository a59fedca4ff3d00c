"""Mixture of linear dynamical systems: every series is assigned to one of `num_systems` LDS models
(surface of the reference's models/MixtureofLinearDynamicalSystems.py:5-47).

All systems live in ONE LinearDynamicalSystems with batch_shape (num_systems,), so the E-step of every
(series, system) pair is one K9 launch (series x systems are the kernel's independent recursions), the
responsibilities are a softmax over the per-series evidences `lds.logZ`, and the M-step is the LDS M-step with
the statistics weighted by the responsibilities (`lds.ss_update(p=...)`).  Series sharded over ranks: set
`lds.reducer`; NA then crosses the ranks with the LDS statistics in the same packed all-reduce.
"""
import torch

from ..dists.Dirichlet import Dirichlet
from .LinearDynamicalSystems import LinearDynamicalSystems


class MixtureofLinearDynamicalSystems():
    def __init__(self, num_systems, obs_shape, hidden_dim, control_dim, regression_dim, device=None, dtype=None):
        self.num_systems = num_systems
        self.lds = LinearDynamicalSystems(obs_shape, hidden_dim, control_dim, regression_dim,
                                          latent_noise='independent', batch_shape=(num_systems,), device=device,
                                          dtype=dtype)
        self.lds.expand_to_batch = True
        self.device, self.dtype = self.lds.device, self.lds.dtype
        self.pi = Dirichlet((num_systems,), device=self.device, dtype=self.dtype)

    def update(self, y, u, r, iters=1, lr=1, verbose=True):
        """ref :12-34 (the reference prints the ELBO change on every iteration; verbose=False skips the print and
        with it the device synchronisation)"""
        y, u, r = self.lds.reshape_inputs(y, u, r)
        ELBO = torch.full((), -torch.inf, device=self.device, dtype=self.dtype)
        for i in range(iters):
            ELBO_last = ELBO
            self.lds.update_latents(y, u, r)
            # responsibilities of every system for every series: softmax of evidence + expected log mixing weight
            log_p = self.lds.logZ + self.pi.loggeomean()
            self.logZ = torch.logsumexp(log_p, -1)  # sample shape
            self.p = torch.exp(log_p - self.logZ.unsqueeze(-1))
            self.NA = self.p.sum(0)
            red = self.lds.reducer
            if red is not None:
                # series sharded over ranks: weight locally, then ONE packed all-reduce of the LDS statistics, NA and
                # the evidence; the M-step that follows is replicated
                self.lds.weight_statistics(self.p)
                self.NA, lz = self.lds.reduce_statistics(extra=[self.NA, self.logZ.sum()])
                pw = None
            else:
                lz, pw = self.logZ.sum(), self.p
            ELBO = lz - self.KLqprior()
            self.pi.ss_update(self.NA, lr=lr)
            self.lds.ss_update(p=pw, lr=lr)  # also weights the statistics handed to obs_model.ss_update
            self.lds.obs_model.ss_update(self.lds.SE_xr_xr, self.lds.SE_y_xr, self.lds.SE_y_y, self.lds.T, lr)
            if verbose:
                print('Percent Change in ELBO = %f' % (((ELBO - ELBO_last) / ELBO_last.abs()).item() * 100))
        self.ELBO_last = ELBO

    def KLqprior(self):
        return self.pi.KLqprior() + self.lds.KLqprior().sum(-1)

    def ELBO(self):
        """the reference's method (:39-40) evaluates `self.KL_last - self.logZ` and returns nothing; this returns the
        bound that its update loop reports"""
        return self.logZ.sum() - self.KLqprior()

    def assignment_pr(self):
        return self.p

    def assignment(self):
        return self.p.argmax(-1)
