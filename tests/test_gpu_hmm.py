"""GPU parity of the K11 HMM forward-backward kernel against a plain log-space restatement (the algorithm of the
reference's models/HMM.py:72-105, written here with torch CPU ops) on seeded inputs incl. forbidden transitions."""
import pytest
import torch

from tests.helpers import TOL32, assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"


def hmm_reference(logits, trans, init, ptemp):
    lse = torch.logsumexp
    T = logits.shape[0]
    fw = [None] * T
    fw[0] = lse(init.unsqueeze(-1) + trans + logits[0].unsqueeze(-2), -2)
    for t in range(1, T):
        fw[t] = lse(fw[t - 1].unsqueeze(-1) + trans + logits[t].unsqueeze(-2), -2)
    logZ = lse(fw[-1], -1, True)
    fw = [f - logZ for f in fw]
    SEzz = torch.zeros(fw[0].shape + fw[0].shape[-1:], dtype=logits.dtype)
    for t in range(T - 2, -1, -1):
        temp = fw[t].unsqueeze(-1) + trans
        xi = (temp - lse(temp, -2, True)) + fw[t + 1].unsqueeze(-2)
        fw[t] = lse(xi, -1)
        SEzz = SEzz + (xi - lse(xi, (-1, -2), True)).exp()
    temp = init.unsqueeze(-1) + trans
    xi = (temp - lse(temp, -2, True)) + fw[0].unsqueeze(-2)
    z0 = lse(xi, -1)
    SEz0 = (z0 - lse(z0, -1, True)).exp()
    SEzz = SEzz + (xi - lse(xi, (-1, -2), True)).exp()
    p = torch.stack(fw)
    p = ((p - p.max(-1, keepdim=True)[0]) / ptemp).exp()
    return p / p.sum(-1, keepdim=True), SEzz, SEz0, logZ.squeeze(-1)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("K,T,lead,batch,ptemp", [(4, 30, (5, 3), (), 1.0), (25, 20, (7,), (), 6.0), (3, 17, (4, 2), (2,), 1.0),
                                                  (2, 9, (1,), (), 1.0), (9, 12, (70,), (), 2.0)])
def test_hmm_forward_backward_vs_restatement(K, T, lead, batch, ptemp, dtype):
    from pyvbmp_amd import ops
    g = torch.Generator().manual_seed(K * 7 + T)
    logits = (2.0 * torch.randn((T,) + lead + (K,), generator=g, dtype=torch.float64)).to(dtype)
    A = torch.rand(batch + (K, K), generator=g, dtype=torch.float64) + 0.1
    mask = torch.rand(K, K, generator=g) > 0.25
    mask |= torch.eye(K, dtype=torch.bool)
    trans = torch.where(mask, A.log(), torch.full_like(A, -float("inf")))
    trans = trans - torch.logsumexp(trans, -1, keepdim=True)
    init = torch.log_softmax(torch.randn(batch + (K,), generator=g, dtype=torch.float64), -1)
    p, SEzz, SEz0, logZ = ops.hmm_forward_backward(logits.to(DEV), trans.to(dtype).to(DEV), init.to(dtype).to(DEV), batch, ptemp)
    rp, rzz, rz0, rlz = hmm_reference(logits.double(), trans, init, ptemp)
    tol = 1e-10 if dtype == torch.float64 else TOL32 * 5
    assert_close(p, rp, tol, what="p")
    assert_close(SEzz, rzz, tol, what="SEzz")
    assert_close(SEz0, rz0, tol, what="SEz0")
    assert_close(logZ, rlz, tol, what="logZ")


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("K,T,lead,scale,keep", [(4, 40, (9,), 400.0, 0.3), (25, 25, (6,), 250.0, 0.15), (8, 60, (33,), 1500.0, 0.5),
                                                 (6, 30, (5,), 60.0, 0.3)])
def test_hmm_extreme_logits_take_the_log_space_step(K, T, lead, scale, keep, dtype):
    """observation logits whose spread (hundreds to thousands) underflows a scaled probability-space recursion, over a
    sparse transition graph: the kernel must notice and fall back to the literal log-space step, chain by chain"""
    from pyvbmp_amd import ops
    g = torch.Generator().manual_seed(K * 11 + T)
    logits = (scale * torch.randn((T,) + lead + (K,), generator=g, dtype=torch.float64)).to(dtype)
    A = torch.rand(K, K, generator=g, dtype=torch.float64) + 0.05
    mask = torch.rand(K, K, generator=g) < keep
    mask |= torch.eye(K, dtype=torch.bool).roll(1, 0)  # a cycle keeps every state reachable
    trans = torch.where(mask, A.log(), torch.full_like(A, -float("inf")))
    trans = trans - torch.logsumexp(trans, -1, keepdim=True)
    init = torch.log_softmax(torch.randn(K, generator=g, dtype=torch.float64), -1)
    p, SEzz, SEz0, logZ = ops.hmm_forward_backward(logits.to(DEV), trans.to(dtype).to(DEV), init.to(dtype).to(DEV), (), 1.0)
    rp, rzz, rz0, rlz = hmm_reference(logits.double(), trans.to(dtype).double(), init.to(dtype).double(), 1.0)
    # fp32: the running log-likelihood reaches ~1e5 here, where one fp32 ulp is 8e-3 -- the message differences that
    # decide an ambiguous state carry that absolute error whatever the recursion
    tol = 1e-10 if dtype == torch.float64 else 5e-3
    assert torch.isfinite(logZ).all()
    assert_close(p, rp, tol, what="p")
    assert_close(SEzz, rzz, tol, what="SEzz")
    assert_close(SEz0, rz0, tol, what="SEz0")
    assert_close(logZ, rlz, tol, what="logZ")


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_hmm_revived_state_needs_the_log_space_step(dtype):
    """state 2 is reachable only through state 1, whose filtered weight is exp(-1000) of the leader's (zero in any
    scaled probability-space recursion); the next observation then favours state 2 by 3000 nats.  The posterior must
    follow the path 1 -> 2, which only a log-space step at that time can see."""
    from pyvbmp_amd import ops
    ninf = -float("inf")
    trans = torch.tensor([[0.0, ninf, ninf], [ninf, -0.7, -0.7], [ninf, ninf, 0.0]], dtype=torch.float64)
    init = torch.log(torch.tensor([0.5, 0.5, 1e-300], dtype=torch.float64))
    logits = torch.zeros(6, 2, 3, dtype=torch.float64)
    logits[0, :, 0] = 1000.0   # state 0 leads state 1 by 1000 nats after the first step
    logits[2, :, 2] = 3000.0   # ... and state 2 is what the third observation wants
    logits[4, 1, 1] = 5.0
    p, SEzz, SEz0, logZ = ops.hmm_forward_backward(logits.to(dtype).to(DEV), trans.to(dtype).to(DEV), init.to(dtype).to(DEV), (), 1.0)
    rp, rzz, rz0, rlz = hmm_reference(logits, trans, init.to(dtype).double(), 1.0)
    assert rp[3, 0, 2] > 0.99  # the reference ends up in state 2
    tol = 1e-10 if dtype == torch.float64 else TOL32 * 5
    assert_close(p, rp, tol, what="p")
    assert_close(SEzz, rzz, tol, what="SEzz")
    assert_close(SEz0, rz0, tol, what="SEz0")
    assert_close(logZ, rlz, tol, what="logZ")
