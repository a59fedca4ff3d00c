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


def test_k12_routing_threshold():
    """where shared_matvec / rows_matmul hand a tall-skinny product to K12 instead of the library GEMM (measured on MI355X,
    tools/exp/rows_time.py: K12 wins 5-6x for k, n <= 9 at >= 1e5 rows and still at 16 x 16; rocBLAS wins from 32 x 32 in
    fp32): at least 16384 rows and k * n <= 256"""
    from pyvbmp_amd._common import _k12_pays
    assert _k12_pays(4_096_000, 6, 6) and _k12_pays(16384, 16, 16) and _k12_pays(100_000, 1, 64) and _k12_pays(100_000, 9, 9)
    assert not _k12_pays(16383, 6, 6)            # too few rows: launch-bound either way, the library call is fine
    assert not _k12_pays(4_096_000, 32, 32)      # the library GEMM is faster (fp32 64 x 64: 39 against 249 us)
    assert not _k12_pays(100_000, 16, 17)
