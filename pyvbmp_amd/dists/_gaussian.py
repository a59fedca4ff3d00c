"""Shared machinery of the two Gaussian nodes (`MultivariateNormal`: event = last axis; `MultivariateNormal_vector_format`:
event = (dim, 1), the message type of the linear-Gaussian transforms).

What the two have in common lives here once: shape bookkeeping, the moment <-> statistic arithmetic written against two
layout hooks (`_mv`: matrix times event vector, `_outer`: event vector times its transpose), the K4 weighted moments of
`raw_update` and the K3a quadratic form of `Elog_like`.  What differs -- which conversions cache what, the reference's
quirks -- stays in the two public classes."""
import math

import torch

from .. import ops

LOG2PI = math.log(2.0 * math.pi)


class GaussianNode():
    _event_axes = 1  # 1: (dim,)   2: (dim, 1)

    def _adopt_shapes(self, like):
        """shape attributes from whichever of mu / invSigmamu was given; False (after the reference's printed complaint)
        when neither was"""
        if like is None:
            print(type(self).__name__ + ': needs mu or invSigmamu to take its shapes from')
            return False
        k = self._event_axes
        self.dim = like.shape[-k]
        self.event_shape = tuple(like.shape[-k:])
        self.batch_shape = tuple(like.shape[:-k])
        self.batch_dim, self.event_dim = len(self.batch_shape), k
        self.device, self.dtype = like.device, like.dtype
        return True

    def _shift_event(self, n):
        self.event_dim, self.batch_dim = self.event_dim + n, self.batch_dim - n
        self.event_shape = self.batch_shape[-n:] + self.event_shape
        self.batch_shape = self.batch_shape[:-n]

    # layout hooks ------------------------------------------------------------------------------------------------
    def _mv(self, M, v):
        return M @ v if self._event_axes == 2 else (M @ v.unsqueeze(-1)).squeeze(-1)

    def _outer(self, v):
        return v @ v.transpose(-2, -1) if self._event_axes == 2 else v.unsqueeze(-1) * v.unsqueeze(-2)

    def _rows(self, X):
        """event vectors as the trailing axis (what K3a / K4 take)"""
        return X.squeeze(-1) if self._event_axes == 2 else X

    # shared arithmetic -------------------------------------------------------------------------------------------
    def EX(self):
        return self.mean()

    def EXXT(self):
        return self.ESigma() + self._outer(self.mean())

    def _set_moments_from_statistics(self, SExx, SEx, n):
        """mu = SEx / n, Sigma = SExx / n - mu mu'; the natural parameters are forgotten"""
        n_vec = n.reshape(tuple(n.shape) + (1,) * self._event_axes)          # broadcasts over the event vector
        n_mat = n.reshape(tuple(n.shape) + (1, 1))                            # ... over the (dim, dim) matrices
        self.mu = SEx / n_vec
        self.Sigma = SExx / n_mat - self._outer(self.mu)
        self.invSigma = None
        self.invSigmamu = None

    def raw_update(self, X, p=None, lr=1.0):
        """data (+ weights) -> weighted moments (K4) -> moment parameters"""
        sample_axes = X.ndim - self.event_dim - self.batch_dim
        n, SEx, SExx = ops.weighted_moments(self._rows(X), p, sample_axes, self.batch_shape)
        self.ss_update(SExx, SEx if self._event_axes == 1 else SEx.unsqueeze(-1), n, lr)

    def Elog_like(self, X):
        """log N(x; mu, Sigma) per (sample, batch element): one K3a launch on the centred data"""
        P = self.EinvSigma()
        no_shift = torch.zeros(self.batch_shape + (self.dim,), device=X.device, dtype=X.dtype)
        const = 0.5 * self.ElogdetinvSigma() - 0.5 * self.dim * LOG2PI
        out = ops.quadform_loglike(self._rows(X - self.mu), P, no_shift, const.expand(self.batch_shape))
        for _ in range(self.event_dim - 2):
            out = out.sum(-1)
        return out

    def KLqprior(self):
        return torch.tensor(0.0, device=self.device, dtype=self.dtype)
