"""One rank of the multi-process sharding check (started by tests/test_gpu_multirank.py, one process per rank; RANK /
WORLD_SIZE / MASTER_* in the environment).  Every rank runs the SHARDED product path -- Mixture._update_sharded,
LinearDynamicalSystems.reduce_statistics, the DMBD sharded iteration, MixtureofLinearDynamicalSystems -- on its slice with
the real SuffStatReducer over torch.distributed (backend nccl = RCCL with one GPU per rank; gloo on device tensors when
the ranks have to share a GPU) and compares with the single-rank run of the same model on the full data at 1e-10 (sums
arrive in a different order).  Exit code 0 = all comparisons passed on this rank."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from tests.helpers import assert_close, load_golden  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("VBMP_TEST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = rank % ndev
    torch.cuda.set_device(dev_index)
    DEV = torch.device("cuda", dev_index)
    dist.init_process_group(backend, device_id=DEV if backend == "nccl" else None)
    from pyvbmp_amd.parallel import SuffStatReducer, shard_bounds
    TOL = 1e-10

    # ---- GMM: samples sharded, ONE packed all-reduce per iteration (Mixture._update_sharded)
    from pyvbmp_amd.models import GaussianMixtureModel
    K, D, N = 4, 16, 6000
    g = torch.Generator().manual_seed(5)
    centers = 3.0 * torch.randn(K, D, generator=g, dtype=torch.float64)
    X = (centers[torch.randint(K, (N,), generator=g)] + torch.randn(N, D, generator=g, dtype=torch.float64)).to(DEV)

    def gmm():
        m = GaussianMixtureModel(K, D, device=DEV, dtype=torch.float64)
        m.dist.mu = (centers + 0.2).to(DEV)
        m.pi.alpha = torch.full((K,), 0.7, dtype=torch.float64, device=DEV)
        return m
    ref = gmm()
    ref.update(X, iters=3, lr=1.0)
    m = gmm()
    m.reducer = SuffStatReducer()
    lo, hi = shard_bounds(N, rank, world)
    m.update(X[lo:hi], iters=3, lr=1.0)
    assert m.reducer.calls == 3, m.reducer.calls
    assert m.reducer.world_size == world
    for f in ("mu", "lambda_mu"):
        assert_close(getattr(m.dist, f), getattr(ref.dist, f), TOL, what="gmm " + f)
    assert_close(m.dist.invU.invU, ref.dist.invU.invU, TOL, what="gmm invU")
    assert_close(m.dist.invU.U, ref.dist.invU.U, TOL, what="gmm U")
    assert_close(m.pi.alpha, ref.pi.alpha, TOL, what="gmm alpha")
    assert_close(m.p, ref.p[lo:hi], TOL, what="gmm p (own slice)")

    # ---- LDS: series sharded, ONE packed all-reduce per iteration (LinearDynamicalSystems.reduce_statistics)
    from pyvbmp_amd.models import LinearDynamicalSystems
    from tools.synth import lorenz
    y = lorenz(60, 12, torch.Generator().manual_seed(3), device="cpu").to(DEV)
    torch.manual_seed(11)
    proto = LinearDynamicalSystems((6,), 6, device=DEV, dtype=torch.float64)
    init = (proto.x0.mu.clone(), proto.A.mu.clone(), proto.A.invU.gamma.alpha.clone(), proto.A.invU.gamma.beta.clone(),
            proto.obs_model.mu.clone())

    def lds():
        q = LinearDynamicalSystems((6,), 6, device=DEV, dtype=torch.float64)
        q.x0.mu, q.A.mu, q.A.invU.gamma.alpha, q.A.invU.gamma.beta, q.obs_model.mu = (t.clone() for t in init)
        q.set_latent_parms()
        return q
    ref = lds()
    ref.update(y, iters=3)
    q = lds()
    q.reducer = SuffStatReducer()
    lo, hi = shard_bounds(y.shape[1], rank, world)
    q.update(y[:, lo:hi], iters=3)
    assert q.reducer.calls == 3, q.reducer.calls
    assert_close(q.A.mu, ref.A.mu, TOL, what="lds A mu")
    assert_close(q.obs_model.mu, ref.obs_model.mu, TOL, what="lds obs mu")
    assert_close(q.x0.mu, ref.x0.mu, TOL, what="lds x0 mu")
    assert_close(q.ELBO_last, ref.ELBO_last, TOL, what="lds ELBO")

    # ---- DMBD: series sharded, two data-dependent exchanges per iteration, each ONE packed all-reduce
    from pyvbmp_amd.models import DynamicMarkovBlanketDiscovery
    c = load_golden("dmbd")["dmbd_lorenz_like"]
    yd = torch.cat((c["y"], c["y"].flip(1) * 0.9), 1).to(DEV)  # 6 series

    def dmbd():
        d = DynamicMarkovBlanketDiscovery(obs_shape=(4, 2), role_dims=(1, 2, 1), hidden_dims=(2, 2, 2), device=DEV,
                                          dtype=torch.float64)
        d.A.mu = c["init_A_mu"].to(DEV)
        d.A.invU.gamma.alpha = c["init_A_alpha"].to(DEV)
        d.A.invU.gamma.beta = c["init_A_beta"].to(DEV)
        d.B.mu = c["init_B_mu"].to(DEV)
        d.obs_model.transition.alpha = c["init_trans_alpha"].to(DEV)
        d.obs_model.initial.alpha = c["init_init_alpha"].to(DEV)
        d.set_latent_parms()
        return d
    ref = dmbd()
    ref.update(yd, None, None, iters=2)
    d = dmbd()
    d.reducer = SuffStatReducer()
    lo, hi = shard_bounds(yd.shape[1], rank, world)
    d.update(yd[:, lo:hi], None, None, iters=2)
    assert d.reducer.calls == 4, d.reducer.calls
    tol = 1e-9  # two alternating E-steps per iteration on top of the reordered sums
    assert_close(d.A.mu, ref.A.mu, tol, what="dmbd A mu")
    assert_close(d.B.mu, ref.B.mu, tol, what="dmbd B mu")
    assert_close(d.obs_model.transition.alpha, ref.obs_model.transition.alpha, tol, what="dmbd trans alpha")
    assert_close(d.ELBO_last, ref.ELBO_last, tol, what="dmbd ELBO")

    # ---- DMBD at BASELINE configs[4]'s real hyper-parameters (examples/Flocking_example.py:38 of the reference: hidden 52,
    # 25 masked roles, regression_dim = -1): the block-form smoother at h = 52, the K = 25 role chain and the 644-entry
    # masked solves, sharded over the series.  Inputs and initial state are the golden fixture's; its 2 series are
    # duplicated (flipped copies, so that the ranks' slices differ) to 4.
    cf = load_golden("dmbd_flock")["dmbd_flocking"]
    yf = torch.cat((cf["y"], cf["y"].flip(1) * 0.95), 1).to(DEV)  # (T, 4 series, 12 birds, 4)

    def flock():
        d = DynamicMarkovBlanketDiscovery(obs_shape=(12, 4), role_dims=(1, 2, 2), hidden_dims=(4, 4, 4), regression_dim=-1,
                                          control_dim=0, number_of_objects=6, unique_obs=False, device=DEV, dtype=torch.float64)
        d.x0.mu = cf["init_x0_mu"].to(DEV)
        d.A.mu = cf["init_A_mu"].to(DEV)
        d.A.invU.gamma.alpha = cf["init_A_alpha"].to(DEV)
        d.A.invU.gamma.beta = cf["init_A_beta"].to(DEV)
        d.B.mu = cf["init_B_mu"].to(DEV)
        d.obs_model.transition.alpha = cf["init_trans_alpha"].to(DEV)
        d.obs_model.initial.alpha = cf["init_init_alpha"].to(DEV)
        d.set_latent_parms()
        return d
    ref = flock()
    assert ref.hidden_dim == 52 and ref.obs_model.transition_mask.shape == (25, 25)
    ref.update(yf, None, None, iters=2, latent_iters=1, lr=1.0)
    d = flock()
    d.reducer = SuffStatReducer()
    lo, hi = shard_bounds(yf.shape[1], rank, world)
    d.update(yf[:, lo:hi], None, None, iters=2, latent_iters=1, lr=1.0)
    assert d.reducer.calls == 4, d.reducer.calls
    assert_close(d.A.mu, ref.A.mu, tol, what="flocking dmbd A mu")
    assert_close(d.A.invU.gamma.beta, ref.A.invU.gamma.beta, tol, what="flocking dmbd A beta")
    assert_close(d.B.mu, ref.B.mu, tol, what="flocking dmbd B mu")
    assert_close(d.B.invU.invU, ref.B.invU.invU, tol, what="flocking dmbd B invU")
    assert_close(d.x0.mu, ref.x0.mu, tol, what="flocking dmbd x0 mu")
    assert_close(d.obs_model.transition.alpha, ref.obs_model.transition.alpha, tol, what="flocking dmbd trans alpha")
    assert_close(d.ELBO_last, ref.ELBO_last, tol, what="flocking dmbd ELBO")
    assert_close(d.px.mu, ref.px.mu[:, lo:hi], tol, what="flocking dmbd px mu (own slice)")
    assert_close(d.obs_model.p, ref.obs_model.p[:, lo:hi], tol, what="flocking dmbd role posteriors (own slice)")

    # ---- batch-sharded NIW (independent posteriors): no collective, the slices tile the unsharded result
    from pyvbmp_amd.dists import NormalInverseWishart
    B, Dn = 1003, 16
    gg = torch.Generator().manual_seed(9)
    A = torch.randn(B, Dn, 32, generator=gg, dtype=torch.float64)
    SExx, SEx, Nn = (A @ A.transpose(-2, -1)).to(DEV), A.sum(-1).to(DEV), torch.full((B,), 32.0, dtype=torch.float64, device=DEV)
    full = NormalInverseWishart((Dn,), (B,), device=DEV, dtype=torch.float64)
    full.ss_update(SExx, SEx, Nn, lr=1.0, beta=None)
    lo, hi = shard_bounds(B, rank, world)
    part = NormalInverseWishart((Dn,), (hi - lo,), device=DEV, dtype=torch.float64)
    part.ss_update(SExx[lo:hi], SEx[lo:hi], Nn[lo:hi], lr=1.0, beta=None)
    assert torch.equal(part.invU.U, full.invU.U[lo:hi]) and torch.equal(part.mu, full.mu[lo:hi])

    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}/{world} on cuda:{dev_index} ({backend}): sharded == single-rank", flush=True)


if __name__ == "__main__":
    main()
