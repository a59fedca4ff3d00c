"""Oracle: discrete HMM forward-backward in log space (torch CPU).  TEST INFRASTRUCTURE ONLY.

Reference: models/HMM.py:72-105 (forward_backward_logits) with utils/torch_functions.py:2-4 (the max-shifted
log-sum-exp).  Pinned to the reference by tests/golden/hmm.npz (tests/test_oracle_golden.py).
"""
import torch


def _lse(x, dims, keepdim=False):
    """ref utils/torch_functions.py:2-4"""
    return torch.logsumexp(x, dims, keepdim=keepdim)


def forward_backward(obs_logits, trans, init, ptemp=1.0):
    """obs_logits (T,)+lead+batch+(K,): observation log-likelihoods; trans batch+(K,K) = E log transition (row from,
    column to); init batch+(K,) = E log initial.  Returns (p, SEzz, SEz0, logZ): smoothed state probabilities with
    temperature ptemp (:100-101), pair posteriors summed over time incl. the initial step (:86-98), logZ (:79-81)."""
    T = obs_logits.shape[0]
    alpha = []
    prev = init
    for t in range(T):  # :76-78  alpha_t[j] = lse_i(alpha_{t-1}[i] + trans[i,j]) + obs_t[j]
        prev = _lse(prev.unsqueeze(-1) + trans + obs_logits[t].unsqueeze(-2), -2)
        alpha.append(prev)
    logZ = _lse(alpha[-1], -1, keepdim=True)
    msg = [a - logZ for a in alpha]  # :80: every time step is shifted by the final normaliser
    pair_sum = torch.zeros(tuple(msg[0].shape) + (msg[0].shape[-1],), dtype=obs_logits.dtype)

    def pair_logits(filt, nxt):
        # :85-86 / :93-94: condition the pair (i -> j) on the smoothed message of the later step
        joint = filt.unsqueeze(-1) + trans
        return joint - _lse(joint, -2, keepdim=True) + nxt.unsqueeze(-2)
    for t in range(T - 2, -1, -1):
        xi = pair_logits(msg[t], msg[t + 1])
        msg[t] = _lse(xi, -1)
        pair_sum = pair_sum + (xi - _lse(xi, (-1, -2), keepdim=True)).exp()
    xi = pair_logits(init, msg[0])
    z0 = _lse(xi, -1)
    SEz0 = (z0 - _lse(z0, -1, keepdim=True)).exp()
    pair_sum = pair_sum + (xi - _lse(xi, (-1, -2), keepdim=True)).exp()
    sm = torch.stack(msg)
    p = ((sm - sm.amax(-1, keepdim=True)) / ptemp).exp()
    return p / p.sum(-1, keepdim=True), pair_sum, SEz0, logZ.squeeze(-1)
