"""Autoregressive HMMs: an HMM whose emission node is a MatrixNormalWishart regression per state (surface of the
reference's models/ARHMM.py:13-91).  ARHMM_prXRY is the role model of DynamicMarkovBlanketDiscovery: latent input x
(a Gaussian message), observed regressors r and observed outputs y."""
import torch

from .._common import shared_matvec, shared_weighted_matvec, shared_weighted_sum
from ..dists.Delta import Delta
from ..dists.MultivariateNormal_vector_format import MultivariateNormal_vector_format
from ..transforms.MatrixNormalWishart import MatrixNormalWishart
from .HMM import HMM


def _weight(p, k):
    return p.reshape(tuple(p.shape) + (1,) * k)


def _emission(dim, n, p, batch_shape, pad_X, X_mask, mask, device, dtype):
    """one MatrixNormalWishart regression y = A x per hidden state: the states are its last batch axis"""
    return MatrixNormalWishart(event_shape=(n, p), batch_shape=tuple(batch_shape) + (dim,), pad_X=pad_X, X_mask=X_mask,
                               mask=mask, device=device, dtype=dtype)


def _role_average(p, P, eta, Res, sum_axis=None):
    """natural-parameter message to x averaged over the state posterior p (None = leave per-state).
    sum_axis (a sample axis of p, negative, counted on p): the caller wants the messages SUMMED over that axis as well (the
    observables of a DynamicMarkovBlanketDiscovery, ref models/DynamicMarkovBlanketDiscovery.py:98-104) with the axis kept.  A
    precision shared by all samples is linear in the weights, so the weights are summed first: sum_obs sum_r p P_r =
    sum_r (sum_obs p) P_r -- n_obs times less GEMM work and output (519 -> 43 MB at the flocking sizes)."""
    if p is None:
        assert sum_axis is None
        return P, eta, Res
    w = _weight(p, 2)
    shared = P.dim() == 3 and p.dim() > 1
    if sum_axis is None:
        # a precision shared by all samples (it depends on the parameters only) is averaged as one GEMM over the states
        Pbar = shared_weighted_sum(P, p) if shared else (P * w).sum(-3)
        return Pbar, (eta * w).sum(-3), (Res * p).sum(-1)
    Pbar = shared_weighted_sum(P, p.sum(sum_axis, True)) if shared else (P * w).sum(-3).sum(sum_axis - 1, True)
    return Pbar, (eta * w).sum(-3).sum(sum_axis - 1, True), (Res * p).sum(-1).sum(sum_axis + 1, True)


class ARHMM(HMM):
    def __init__(self, dim, n, p, batch_shape=(), pad_X=True, X_mask=None, mask=None, transition_mask=None,
                 device=None, dtype=None):
        super().__init__(_emission(dim, n, p, batch_shape, pad_X, X_mask, mask, device, dtype),
                         transition_mask=transition_mask)

    def obs_logits(self, XY, t=None):
        if t is not None:
            return self.obs_dist.Elog_like(XY[0][t], XY[1][t])
        return self.obs_dist.Elog_like(XY[0], XY[1])

    def update_obs_parms(self, XY, lr=1.0, beta=None):
        self.obs_dist.raw_update(XY[0], XY[1], p=self.p, lr=lr, beta=beta)


class ARHMM_prXY(HMM):
    def __init__(self, dim, n, p, batch_shape=(), X_mask=None, mask=None, pad_X=True, transition_mask=None,
                 device=None, dtype=None):
        super().__init__(_emission(dim, n, p, batch_shape, pad_X, X_mask, mask, device, dtype),
                         transition_mask=transition_mask)

    def obs_logits(self, XY):
        return self.obs_dist.Elog_like_given_pX_pY(XY[0], XY[1])

    def update_obs_parms(self, XY, lr=1.0, beta=None):
        self.obs_dist.update(XY[0], XY[1], self.p, lr=lr, beta=beta)

    def Elog_like_X_given_pY(self, pY):
        px, Res = self.obs_dist.Elog_like_X_given_pY(pY)
        return _role_average(self.p, px.invSigma, px.invSigmamu, Res)


class ARHMM_prXRY(HMM):
    """x: Gaussian message (vector format), r and y observed."""

    def __init__(self, dim, n, p1, p2, batch_shape=(), mask=None, X_mask=None, transition_mask=None, pad_X=False,
                 device=None, dtype=None):
        self.p1 = p1
        self.p2 = p2
        super().__init__(_emission(dim, n, p1 + p2, batch_shape, pad_X, X_mask, mask, device, dtype),
                         transition_mask=transition_mask)

    def _joint_input(self, XRY):
        """[x; r] as one Gaussian: covariance block-diag(Sigma_x, 0), mean [mu_x; r] (ref :59-66)"""
        px, R = XRY[0], XRY[1]
        Sx = px.ESigma()
        lead = torch.broadcast_shapes(Sx.shape[:-2], R.shape[:-2])
        # The covariance does not depend on r: it is assembled ONCE per distinct latent message and handed on as a broadcast
        # (stride-0) view over the remaining sample axes -- the observables of a DynamicMarkovBlanketDiscovery all see the same
        # message, and materialising its 52 x 52 covariance for each of them was 519 MB written and re-read per iteration at
        # the flocking sizes.  Consumers that sum over samples reduce their weights over such axes first (MatrixNormalWishart._moments).
        keep = tuple(i for i, st in enumerate(Sx.stride()[:-2]) if not (st == 0 and Sx.shape[i] > 1))
        Sc = Sx[tuple(slice(None) if i in keep else slice(0, 1) for i in range(Sx.ndim - 2))]
        if self.p2 == 0:
            Sigma_c = Sc
        else:
            Sigma_c = torch.zeros(tuple(Sc.shape[:-2]) + (self.p1 + self.p2,) * 2, device=Sx.device, dtype=Sx.dtype)
            Sigma_c[..., :self.p1, :self.p1] = Sc
        Sigma = Sigma_c.expand(tuple(lead) + (self.p1 + self.p2,) * 2)
        mu = torch.cat((px.mean().expand(tuple(lead) + (self.p1, 1)), R.expand(tuple(lead) + (self.p2, 1))), dim=-2)
        return MultivariateNormal_vector_format(mu=mu, Sigma=Sigma)

    def Elog_like(self, XRY):
        return (self.obs_logits(XRY) * self.p).sum(-1)

    def obs_logits(self, XRY):
        return self.obs_dist.Elog_like_given_pX_pY(self._joint_input(XRY), Delta(XRY[2]))

    def update_obs_parms(self, XRY, lr=1.0, beta=None):
        self.obs_dist.update(self._joint_input(XRY), Delta(XRY[2]), p=self.p, lr=lr, beta=beta)

    def Elog_like_X(self, YR, sum_axis=None):
        """likelihood of x as natural parameters, averaged over the role posterior (ref :79-91); sum_axis: see _role_average"""
        B, p1, R = self.obs_dist, self.p1, YR[1]
        if self.p is not None and not B.pad_X and B.batch_dim == 1 and B.event_dim == 2 and self.p.dim() > 1 and YR[0].is_cuda:
            # Role-averaged message without the per-role intermediates: the precision and the linear term are linear in the role
            # weights, so sum_r p_r (G_r' y) is ONE GEMM on the features p (x) y (shared_weighted_matvec) and sum_r p_r H_r one GEMM on
            # p (summed over `sum_axis` first) -- the per-(sample, role) message (samples x 25 x 52) is never formed
            H, Gt = B.EXTinvUX(), B.EXTinvU()  # (roles, p, p), (roles, p, n)
            k = -1 - B.event_dim                # the role axis of the samples
            Y = YR[0].squeeze(k)
            Rs = R.squeeze(k)
            eta = shared_weighted_matvec(Gt[..., :p1, :], Y, self.p)
            Res = B._residual_y(YR[0])
            if self.p2 > 0:
                eta = eta - shared_weighted_matvec(H[..., :p1, p1:], Rs, self.p)
                Res = Res - 0.5 * (H[..., p1:, p1:] * (R * R.transpose(-2, -1))).sum((-1, -2))
                Res = Res + (shared_matvec(Gt[..., p1:, :], YR[0]) * R).sum((-1, -2))
            Res = (Res * self.p).sum(-1)
            Hxx = H[..., :p1, :p1]
            if sum_axis is None:
                return shared_weighted_sum(Hxx, self.p), eta, Res
            return (shared_weighted_sum(Hxx, self.p.sum(sum_axis, True)), eta.sum(sum_axis - 1, True), Res.sum(sum_axis + 1, True))
        P_xr, eta_xr, Res = self.obs_dist.Elog_like_X(YR[0])
        P = P_xr[..., :p1, :p1]
        eta = eta_xr[..., :p1, :] - shared_matvec(P_xr[..., :p1, p1:], R)
        Res = Res - 0.5 * (P_xr[..., p1:, p1:] * (R * R.transpose(-2, -1))).sum((-1, -2))
        Res = Res + (eta_xr[..., p1:, :] * R).sum((-1, -2))
        return _role_average(self.p, P, eta, Res, sum_axis)
