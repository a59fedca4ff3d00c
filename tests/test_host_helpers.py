"""Host-side helpers that replace broadcasting products by one library GEMM (CPU tensors: plain torch, no kernels)."""
import torch

from pyvbmp_amd._common import shared_matvec, shared_weighted_sum


def test_shared_matvec_matches_broadcast_matmul():
    g = torch.Generator().manual_seed(0)
    G = torch.randn(5, 7, 3, generator=g, dtype=torch.float64)
    Y = torch.randn(4, 6, 2, 1, 3, 1, generator=g, dtype=torch.float64)
    out = shared_matvec(G, Y)
    assert out.shape == (4, 6, 2, 5, 7, 1)
    assert torch.allclose(out, G @ Y, rtol=1e-13, atol=1e-13)
    G2 = torch.randn(2, 5, 7, 3, generator=g, dtype=torch.float64)  # two batch axes
    Y2 = torch.randn(9, 1, 1, 3, 1, generator=g, dtype=torch.float64)
    assert torch.allclose(shared_matvec(G2, Y2), G2 @ Y2, rtol=1e-13, atol=1e-13)
    Gs = G[..., :, 1:]  # a non-contiguous slice, as the callers pass
    Ys = Y[..., 1:, :]
    assert torch.allclose(shared_matvec(Gs, Ys), Gs @ Ys, rtol=1e-13, atol=1e-13)


def test_shared_matvec_leaves_other_shapes_to_matmul():
    g = torch.Generator().manual_seed(1)
    G = torch.randn(5, 7, 3, generator=g, dtype=torch.float64)
    for Y in (torch.randn(4, 5, 3, 1, generator=g, dtype=torch.float64),   # per-batch vectors
              torch.randn(3, 1, generator=g, dtype=torch.float64),         # no sample axes
              torch.randn(4, 1, 3, 2, generator=g, dtype=torch.float64)):  # matrices, not vectors
        assert torch.equal(shared_matvec(G, Y), G @ Y)
    M = torch.randn(7, 3, generator=g, dtype=torch.float64)               # no batch axis on G
    Y = torch.randn(4, 3, 1, generator=g, dtype=torch.float64)
    assert torch.allclose(shared_matvec(M, Y), M @ Y, rtol=1e-13, atol=1e-13)  # one row-major GEMM on the CPU / for few rows


def test_shared_weighted_sum_matches_broadcast_sum():
    g = torch.Generator().manual_seed(2)
    P = torch.randn(5, 4, 6, generator=g, dtype=torch.float64)
    w = torch.rand(3, 2, 5, generator=g, dtype=torch.float64)
    ref = (P * w.reshape(3, 2, 5, 1, 1)).sum(-3)
    out = shared_weighted_sum(P, w)
    assert out.shape == (3, 2, 4, 6)
    assert torch.allclose(out, ref, rtol=1e-13, atol=1e-13)


def test_rows_matmul_is_matmul_off_the_device():
    from pyvbmp_amd._common import rows_matmul
    g = torch.Generator().manual_seed(4)
    X = torch.randn(5, 7, 3, generator=g, dtype=torch.float64)
    W = torch.randn(3, 4, generator=g, dtype=torch.float64)
    assert torch.equal(rows_matmul(X, W), X @ W)
