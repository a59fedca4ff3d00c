"""N > 1 path on CPU: two gloo ranks shard the SAMPLE axis of a GMM, compute their local statistics (with
the CPU oracle, this is a host-logic test), push them through pyvbmp_amd.parallel.SuffStatReducer (one flat
all-reduce per VB iteration) and must land on the single-process result.  Also covers the batch-axis
shard helper used by bench.py (independent posteriors, no collective)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import assert_close


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_bounds_partition():
    from pyvbmp_amd.parallel import shard_bounds
    for n in (0, 1, 7, 8, 1_000_000, 1_000_003):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1


def _gmm_worker(rank, world, port, X, mu_init, alpha_init, iters, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import mixture as omix
        from oracle import niw as oniw
        from pyvbmp_amd.parallel import SuffStatReducer, shard_bounds
        K, D = mu_init.shape
        lo, hi = shard_bounds(X.shape[0], rank, world)
        Xl = X[lo:hi]
        red = SuffStatReducer()
        st = oniw.niw_new((D,), (K,), mu_init=mu_init.clone())
        alpha_0 = torch.full((K,), 0.5, dtype=torch.float64)
        alpha = alpha_init.clone()
        for _ in range(iters):
            p, NA, logZ = omix.mixture_estep(st, alpha, Xl)
            SExx, SEx, N = oniw.niw_raw_moments(Xl.unsqueeze(-2), p, (K,), (D,))
            NA, logZ, N, SEx, SExx = red.all_reduce([NA, logZ, N, SEx, SExx])  # ONE collective
            alpha = omix.dirichlet_ss_update(alpha_0, alpha, NA)
            st = oniw.niw_ss_update(st, SExx, SEx, N, lr=1.0, beta=None)
        assert red.calls == iters
        if rank == 0:
            torch.save({"mu": st["mu"], "invU": st["W"]["invU"], "U": st["W"]["U"], "alpha": alpha, "logZ": logZ,
                        "NA": NA}, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_gmm_sample_sharding_two_ranks(tmp_path):
    from oracle import mixture as omix
    from oracle import niw as oniw
    K, D, N, iters = 3, 4, 600, 3
    g = torch.Generator().manual_seed(11)
    centers = 4.0 * torch.randn(K, D, generator=g, dtype=torch.float64)
    X = centers[torch.randint(K, (N,), generator=g)] + torch.randn(N, D, generator=g, dtype=torch.float64)
    mu_init = centers + 0.3 * torch.randn(K, D, generator=g, dtype=torch.float64)
    alpha_init = 0.5 + torch.rand(K, generator=g, dtype=torch.float64)
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_gmm_worker, args=(2, _free_port(), X, mu_init, alpha_init, iters, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    # single-process reference
    st = oniw.niw_new((D,), (K,), mu_init=mu_init.clone())
    alpha_0 = torch.full((K,), 0.5, dtype=torch.float64)
    alpha = alpha_init.clone()
    for _ in range(iters):
        st, alpha, o = omix.mixture_iteration(st, alpha_0, alpha, X, 1.0, (K,), (), (D,))
    assert_close(got["mu"], st["mu"], 1e-10)
    assert_close(got["invU"], st["W"]["invU"], 1e-10)
    assert_close(got["U"], st["W"]["U"], 1e-10)
    assert_close(got["alpha"], alpha, 1e-10)
    assert_close(got["logZ"], o["logZ"], 1e-10)
    assert_close(got["NA"], o["NA"], 1e-10)


def test_reducer_single_process_is_identity():
    from pyvbmp_amd.parallel import SuffStatReducer
    r = SuffStatReducer()
    a, b = torch.randn(3, 2), torch.randn(())
    x, y = r.all_reduce([a, b])
    assert torch.equal(x, a) and torch.equal(y, b) and r.calls == 0
