"""GPU parity of the K11 HMM forward-backward kernel against outputs of the reference (tests/golden/hmm.npz) and, on larger
seeded inputs incl. forbidden transitions and extreme logits, against the CPU oracle (oracle/hmm.py)."""
import pytest
import torch

from tests.helpers import TOL32, assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"


def hmm_reference(logits, trans, init, ptemp):
    """the CPU oracle (oracle/hmm.py, pinned to the reference's models/HMM.py:72-105 by tests/golden/hmm.npz)"""
    from oracle import hmm as ohmm
    return ohmm.forward_backward(logits, trans, init, ptemp)


HMM_CASES = ["hmm_k25_roles", "hmm_k25_roles_ptemp", "hmm_k4", "hmm_k3_b2", "hmm_k9_T1", "hmm_k2_T2"]


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("case", HMM_CASES)
def test_hmm_forward_backward_golden(golden, case, dtype):
    """K11 against outputs of the reference's HMM.forward_backward_logits (K = 25 with the masked role transitions of the
    flocking DMBD, ptemp != 1, batched transition matrices, T = 1 and T = 2)"""
    from pyvbmp_amd import ops
    c = golden("hmm")[case]
    batch = tuple(int(v) for v in c["batch_shape"])
    p, SEzz, SEz0, logZ = ops.hmm_forward_backward(c["logits"].to(DEV, dtype), c["trans"].to(DEV, dtype), c["init"].to(DEV, dtype),
                                                   batch, float(c["ptemp"]))
    tol = 1e-10 if dtype == torch.float64 else TOL32
    assert_close(p, c["p"], tol, what="p")
    assert_close(SEzz, c["SEzz"], tol, what="SEzz")
    assert_close(SEz0, c["SEz0"], tol, what="SEz0")
    assert_close(logZ, c["logZ"], tol, what="logZ")


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_hmm_class_golden(golden, dtype):
    """the product's HMM class (its Dirichlet nodes feed K11) on the K = 25 role-chain golden"""
    from pyvbmp_amd.dists import NormalInverseWishart
    from pyvbmp_amd.models.HMM import HMM
    c = golden("hmm")["hmm_k25_roles"]
    K = int(c["K"])
    h = HMM(NormalInverseWishart((2,), (K,), device=DEV, dtype=dtype), transition_mask=c["mask"].to(DEV), ptemp=float(c["ptemp"]))
    # loggeomean is what the kernel consumes: check the class against it at its own Dirichlet state, then replay
    h.transition.loggeomean = lambda: c["trans"].to(DEV, dtype)
    h.initial.loggeomean = lambda: c["init"].to(DEV, dtype)
    p, SEzz, SEz0, logZ = h.forward_backward_logits(c["logits"].to(DEV, dtype))
    tol = 1e-10 if dtype == torch.float64 else TOL32
    assert_close(p, c["p"], tol, what="p")
    assert_close(SEzz, c["SEzz"], tol, what="SEzz")
    assert_close(logZ, c["logZ"], tol, what="logZ")


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("K,T,lead,batch,ptemp", [(4, 30, (5, 3), (), 1.0), (25, 20, (7,), (), 6.0), (3, 17, (4, 2), (2,), 1.0),
                                                  (2, 9, (1,), (), 1.0), (9, 12, (70,), (), 2.0)])
def test_hmm_forward_backward_vs_oracle(K, T, lead, batch, ptemp, dtype):
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


def test_hmm_beyond_the_kernel_limit_raises():
    """more than VBMP_HMM_MAX_K = 64 states: a VbmpHipError naming the limit (no host-loop fallback)"""
    from pyvbmp_amd import _lib
    from pyvbmp_amd.dists import NormalInverseWishart
    from pyvbmp_amd.models import HMM
    K = 65
    m = HMM(NormalInverseWishart((2,), (K,), device=DEV, dtype=torch.float64))
    with pytest.raises(_lib.VbmpHipError, match="VBMP_HMM_MAX_K"):
        m.forward_backward_logits(torch.zeros(3, K, device=DEV, dtype=torch.float64))
    # extreme logits and a sharp chain: the extended-range step is the only step there is now
    from pyvbmp_amd import ops
    lg = torch.zeros(40, 3, 25, dtype=torch.float64, device=DEV)
    lg[..., 7] = 5000.0
    lg[20:, :, 7] = -5000.0
    lg[20:, :, 3] = 4000.0
    tr = torch.log_softmax(torch.randn(25, 25, dtype=torch.float64, generator=torch.Generator().manual_seed(1)), -1).to(DEV)
    ini = torch.log_softmax(torch.zeros(25, dtype=torch.float64), -1).to(DEV)
    p, SEzz, SEz0, logZ = ops.hmm_forward_backward(lg, tr, ini, ())
    from oracle import hmm as ohmm
    po, So, S0o, lzo = ohmm.forward_backward(lg.cpu(), tr.cpu(), ini.cpu())
    assert_close(p, po, 1e-10, what="p")
    assert_close(SEzz, So, 1e-10, what="SEzz")
    assert_close(logZ, lzo, 1e-10, what="logZ")
