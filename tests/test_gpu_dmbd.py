"""GPU parity of DynamicMarkovBlanketDiscovery (role HMM + masked LDS on the HIP kernels) against golden
fixtures captured from the reference."""
import pytest
import torch

from tests.helpers import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"
CASES = ["dmbd_lorenz_like", "dmbd_latent2", "dmbd_two_objects"]


def _n_iters(c):
    return max(int(k[2]) for k in c if k.startswith("it") and k[2].isdigit())


@pytest.mark.parametrize("case", CASES)
def test_dmbd_golden(golden, case):
    from pyvbmp_amd.models import DynamicMarkovBlanketDiscovery
    c = golden("dmbd")[case]
    m = DynamicMarkovBlanketDiscovery(obs_shape=(int(c["n_obs"]), int(c["obs_dim"])),
                                      role_dims=tuple(int(v) for v in c["role_dims"]),
                                      hidden_dims=tuple(int(v) for v in c["hidden_dims"]), batch_shape=(), regression_dim=0,
                                      control_dim=0, number_of_objects=int(c["number_of_objects"]), device=DEV,
                                      dtype=torch.float64)
    # the masks are deterministic functions of the dims
    assert torch.equal(m.A.mask.cpu(), c["A_mask"])
    assert torch.equal(m.B.X_mask.cpu(), c["B_X_mask"])
    assert torch.equal(m.obs_model.transition_mask.cpu(), c["role_mask"])
    assert_close(m.B.invU.invU_0, c["init_B_invU_0"])
    assert_close(m.B.invU.logdet_invU_0, c["init_B_logdet_invU_0"])
    # replay the reference's random initial state
    m.x0.mu = c["init_x0_mu"].to(DEV)
    m.A.mu = c["init_A_mu"].to(DEV)
    m.A.invU.gamma.alpha = c["init_A_alpha"].to(DEV)
    m.A.invU.gamma.beta = c["init_A_beta"].to(DEV)
    m.B.mu = c["init_B_mu"].to(DEV)
    m.obs_model.transition.alpha = c["init_trans_alpha"].to(DEV)
    m.obs_model.initial.alpha = c["init_init_alpha"].to(DEV)
    m.set_latent_parms()
    y = c["y"].to(DEV)
    li = int(c["latent_iters"])
    for it in range(1, _n_iters(c) + 1):
        pre = f"it{it}_"
        tol = 1e-10 if it == 1 else 1e-9  # later iterations start from the earlier ones' rounding differences
        m.update(y, None, None, iters=1, latent_iters=li, lr=1.0)
        assert_close(m.obs_model.p, c[pre + "p"], tol, what=pre + "p")
        assert_close(m.NA, c[pre + "NA"], tol, what=pre + "NA")
        assert_close(m.SEzz, c[pre + "SEzz"], tol, what=pre + "SEzz")
        assert_close(m.SEz0, c[pre + "SEz0"], tol, what=pre + "SEz0")
        assert_close(m.px.mu, c[pre + "px_mu"], tol, what=pre + "px mu")
        assert_close(m.px.Sigma, c[pre + "px_Sigma"], tol, what=pre + "px Sigma")
        assert_close(m.logZ, c[pre + "logZ"], tol, what=pre + "logZ")
        assert_close(m.ELBO_last, c[pre + "ELBO"], tol, what=pre + "ELBO")
        assert_close(m.A.mu, c[pre + "A_mu"], tol, what=pre + "A mu")
        assert_close(m.A.invU.gamma.beta, c[pre + "A_beta"], tol)
        assert_close(m.B.mu, c[pre + "B_mu"], tol, what=pre + "B mu")
        assert_close(m.B.invU.invU, c[pre + "B_invU_invU"], tol)
        assert_close(m.obs_model.transition.alpha, c[pre + "trans_alpha"], tol)
        assert_close(m.x0.mu, c[pre + "x0_mu"], tol)
    assert_close(m.assignment_pr(), c["assignment_pr"], 1e-9)
    assert_close(m.particular_assignment_pr(), c["particular_assignment_pr"], 1e-9)


def test_dmbd_flocking_hyperparameters_golden(golden):
    """BASELINE configs[4] at the reference example's exact hyper-parameters (examples/Flocking_example.py:38:
    role_dims=(1,2,2), hidden_dims=(4,4,4), number_of_objects=6, regression_dim=-1 -> hidden 52, 25 roles) against two VB
    iterations of the imported reference on boids data (T=20, 2 runs, 12 birds).  This drives the block-per-series
    smoother (h = 52), the K = 25 masked role chain, the masked transition / emission updates."""
    from pyvbmp_amd.models import DynamicMarkovBlanketDiscovery
    c = golden("dmbd_flock")["dmbd_flocking"]
    m = DynamicMarkovBlanketDiscovery(obs_shape=(12, 4), role_dims=(1, 2, 2), hidden_dims=(4, 4, 4), regression_dim=-1,
                                      control_dim=0, number_of_objects=6, unique_obs=False, device=DEV, dtype=torch.float64)
    assert m.hidden_dim == 52 and m.obs_model.transition_mask.shape == (25, 25)
    assert torch.equal(m.A.mask.cpu(), c["A_mask"])
    assert torch.equal(m.B.X_mask.cpu(), c["B_X_mask"])
    assert torch.equal(m.obs_model.transition_mask.cpu(), c["role_mask"])
    assert_close(m.B.invU.invU_0, c["init_B_invU_0"])
    m.x0.mu = c["init_x0_mu"].to(DEV)
    m.A.mu = c["init_A_mu"].to(DEV)
    m.A.invU.gamma.alpha = c["init_A_alpha"].to(DEV)
    m.A.invU.gamma.beta = c["init_A_beta"].to(DEV)
    m.B.mu = c["init_B_mu"].to(DEV)
    m.obs_model.transition.alpha = c["init_trans_alpha"].to(DEV)
    m.obs_model.initial.alpha = c["init_init_alpha"].to(DEV)
    m.set_latent_parms()
    y = c["y"].to(DEV)
    for it in (1, 2):
        pre = f"it{it}_"
        tol = 1e-10 if it == 1 else 1e-9  # the second iteration starts from the first one's rounding differences
        m.update(y, None, None, iters=1, latent_iters=1, lr=1.0)
        assert_close(m.obs_model.p, c[pre + "p"], tol, what=pre + "p")
        assert_close(m.NA, c[pre + "NA"], tol, what=pre + "NA")
        assert_close(m.SEzz, c[pre + "SEzz"], tol, what=pre + "SEzz")
        assert_close(m.SEz0, c[pre + "SEz0"], tol, what=pre + "SEz0")
        assert_close(m.logZ, c[pre + "logZ"], tol, what=pre + "logZ")
        assert_close(m.ELBO_last, c[pre + "ELBO"], tol, what=pre + "ELBO")
        assert_close(m.px.mu, c[pre + "px_mu"], tol, what=pre + "px mu")
        assert_close(m.A.mu, c[pre + "A_mu"], tol, what=pre + "A mu")
        assert_close(m.A.invU.gamma.beta, c[pre + "A_beta"], tol, what=pre + "A beta")
        if float(c[pre + "B_mu"].abs().max()) == 0.0:
            assert float(m.B.mu.abs().max()) == 0.0
        else:
            assert_close(m.B.mu, c[pre + "B_mu"], tol, what=pre + "B mu")
        assert_close(m.B.invU.invU, c[pre + "B_invU_invU"], tol, what=pre + "B invU")
        assert_close(m.x0.mu, c[pre + "x0_mu"], tol, what=pre + "x0 mu")
        assert_close(m.obs_model.transition.alpha, c[pre + "trans_alpha"], tol, what=pre + "trans alpha")


@pytest.mark.parametrize("which", ["lorenz_like", "flocking"])
def test_dmbd_graphed_update_matches_eager(golden, which):
    """DynamicMarkovBlanketDiscovery.update(..., graphed=True): the VB iteration replayed as ONE HIP graph (pyvbmp_amd.graph)
    against the eager loop -- same state after 5 iterations (atomics reorder the sums: 1e-8), at the Lorenz-like and at the
    flocking hyper-parameters (hidden 52, 25 roles: block-form smoother, masked solves and the role chain inside the capture)."""
    from pyvbmp_amd.models import DynamicMarkovBlanketDiscovery
    if which == "flocking":
        c = golden("dmbd_flock")["dmbd_flocking"]
        kw = dict(obs_shape=(12, 4), role_dims=(1, 2, 2), hidden_dims=(4, 4, 4), regression_dim=-1, control_dim=0,
                  number_of_objects=6, unique_obs=False)
    else:
        c = golden("dmbd")["dmbd_lorenz_like"]
        kw = dict(obs_shape=(int(c["n_obs"]), int(c["obs_dim"])), role_dims=tuple(int(v) for v in c["role_dims"]),
                  hidden_dims=tuple(int(v) for v in c["hidden_dims"]), regression_dim=0, control_dim=0,
                  number_of_objects=int(c["number_of_objects"]))
    y = c["y"].to(DEV)
    out = []
    for graphed in (False, True):
        m = DynamicMarkovBlanketDiscovery(device=DEV, dtype=torch.float64, **kw)
        m.x0.mu = c["init_x0_mu"].to(DEV)
        m.A.mu = c["init_A_mu"].to(DEV)
        m.A.invU.gamma.alpha = c["init_A_alpha"].to(DEV)
        m.A.invU.gamma.beta = c["init_A_beta"].to(DEV)
        m.B.mu = c["init_B_mu"].to(DEV)
        m.obs_model.transition.alpha = c["init_trans_alpha"].to(DEV)
        m.obs_model.initial.alpha = c["init_init_alpha"].to(DEV)
        m.set_latent_parms()
        m.update(y, None, None, iters=3, latent_iters=1, lr=1.0, graphed=graphed)
        m.update(y, None, None, iters=2, latent_iters=1, lr=1.0, graphed=graphed)  # a second call reuses the cached graph
        out.append(m)
    a, b = out
    assert b.iters == a.iters == 5 and b.ELBO_save.shape == a.ELBO_save.shape
    tol = 1e-8
    assert_close(b.ELBO_save[1:], a.ELBO_save[1:], tol, what="ELBO trace")
    assert_close(b.px.mu, a.px.mu, tol, what="px mu")
    assert_close(b.obs_model.p, a.obs_model.p, tol, what="role posteriors")
    assert_close(b.A.mu, a.A.mu, tol, what="A mu")
    assert_close(b.B.invU.invU, a.B.invU.invU, tol, what="B invU")
    assert_close(b.obs_model.transition.alpha, a.obs_model.transition.alpha, tol, what="transition alpha")


class _PairReducer:
    """two 'ranks' (threads on one GPU) with a barrier standing in for the all-reduce"""

    def __init__(self):
        import threading
        self.bar = threading.Barrier(2)
        self.slots = [None, None]

    def view(self, rank):
        outer = self

        class V:
            calls = 0

            def all_reduce(self, tensors):
                V.calls += 1
                outer.slots[rank] = [t.clone() for t in tensors]
                outer.bar.wait()
                out = [a + b for a, b in zip(*outer.slots)]
                outer.bar.wait()
                return out
        return V()


def _run_pair(make, data_slices, step):
    import threading
    shared = _PairReducer()
    models, errs = [make(), make()], []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            models[rank].reducer = shared.view(rank)
            step(models[rank], data_slices[rank])
        except Exception as e:  # pragma: no cover
            errs.append(e)
            shared.bar.abort()
    ts = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    return models


def test_lds_series_sharded_matches_single():
    """LDS with the series split over two ranks: one packed all-reduce per iteration, same posterior as unsharded"""
    from pyvbmp_amd.models import LinearDynamicalSystems
    from tests.test_gpu_lds import lorenz
    g = torch.Generator().manual_seed(3)
    y = lorenz(80, 24, g).to(DEV)
    ref = LinearDynamicalSystems((6,), 6, device=DEV, dtype=torch.float64)
    init = (ref.x0.mu.clone(), ref.A.mu.clone(), ref.A.invU.gamma.alpha.clone(), ref.A.invU.gamma.beta.clone(),
            ref.obs_model.mu.clone())

    def make():
        m = LinearDynamicalSystems((6,), 6, device=DEV, dtype=torch.float64)
        m.x0.mu, m.A.mu, m.A.invU.gamma.alpha, m.A.invU.gamma.beta, m.obs_model.mu = (t.clone() for t in init)
        m.set_latent_parms()
        return m
    ref.update(y, iters=3)
    models = _run_pair(make, [y[:, :12], y[:, 12:]], lambda m, d: m.update(d, iters=3))
    for m in models:
        assert_close(m.A.mu, ref.A.mu, 1e-9)
        assert_close(m.obs_model.mu, ref.obs_model.mu, 1e-9)
        assert_close(m.x0.invU.invU, ref.x0.invU.invU, 1e-9)
        assert_close(m.ELBO_last, ref.ELBO_last, 1e-9)


def test_dmbd_series_sharded_matches_single(golden):
    """BASELINE config 5 logic on one GPU: DMBD with the series split over two ranks (two packed all-reduces per VB
    iteration) reproduces the unsharded run"""
    from pyvbmp_amd.models import DynamicMarkovBlanketDiscovery
    c = golden("dmbd")["dmbd_lorenz_like"]
    y = torch.cat((c["y"], c["y"].flip(1) * 0.9), 1).to(DEV)  # 6 series

    def make():
        m = DynamicMarkovBlanketDiscovery(obs_shape=(4, 2), role_dims=(1, 2, 1), hidden_dims=(2, 2, 2), device=DEV,
                                          dtype=torch.float64)
        m.A.mu = c["init_A_mu"].to(DEV)
        m.A.invU.gamma.alpha = c["init_A_alpha"].to(DEV)
        m.A.invU.gamma.beta = c["init_A_beta"].to(DEV)
        m.B.mu = c["init_B_mu"].to(DEV)
        m.obs_model.transition.alpha = c["init_trans_alpha"].to(DEV)
        m.obs_model.initial.alpha = c["init_init_alpha"].to(DEV)
        m.set_latent_parms()
        return m
    ref = make()
    ref.update(y, None, None, iters=2)
    models = _run_pair(make, [y[:, :3], y[:, 3:]], lambda m, d: m.update(d, None, None, iters=2))
    for m in models:
        assert m.reducer.calls == 4  # two exchange steps per iteration
        assert_close(m.A.mu, ref.A.mu, 1e-8)
        assert_close(m.B.mu, ref.B.mu, 1e-8)
        assert_close(m.obs_model.transition.alpha, ref.obs_model.transition.alpha, 1e-8)
        assert_close(m.ELBO_last, ref.ELBO_last, 1e-8)


def test_mixture_of_lds_series_sharded_matches_single(golden):
    """MixtureofLinearDynamicalSystems with the series split over two ranks: the weighted LDS statistics, NA and the
    evidence cross the ranks in ONE packed all-reduce per iteration; same posterior as the unsharded run"""
    from pyvbmp_amd.models import MixtureofLinearDynamicalSystems
    c = golden("mixlds")["mix3_h3_o5"]
    y = c["y"].to(DEV)  # (T, 6 series, 5)

    def make():
        m = MixtureofLinearDynamicalSystems(3, (5,), 3, 0, 0, device=DEV, dtype=torch.float64)
        m.lds.x0.mu = c["init_x0_mu"].to(DEV)
        m.lds.A.mu = c["init_A_mu"].to(DEV)
        m.lds.A.invU.gamma.alpha = c["init_A_alpha"].to(DEV)
        m.lds.A.invU.gamma.beta = c["init_A_beta"].to(DEV)
        m.lds.obs_model.mu = c["init_obs_mu"].to(DEV)
        m.lds.set_latent_parms()
        m.pi.alpha = c["init_pi_alpha"].to(DEV)
        return m

    def step(m, d):
        m.lds.reducer = m.reducer  # _run_pair hands the reducer to the model; the statistics live in its LDS
        m.update(d, None, None, iters=3, verbose=False)
    ref = make()
    ref.update(y, None, None, iters=3, verbose=False)
    assert_close(ref.p, c["it3_p"], 1e-9)
    models = _run_pair(make, [y[:, :3], y[:, 3:]], step)
    for m in models:
        assert m.reducer.calls == 3
        assert_close(m.lds.A.mu, ref.lds.A.mu, 1e-9)
        assert_close(m.lds.obs_model.mu, ref.lds.obs_model.mu, 1e-9)
        assert_close(m.pi.alpha, ref.pi.alpha, 1e-9)
        assert_close(m.ELBO_last, ref.ELBO_last, 1e-9)
    assert_close(torch.cat((models[0].p, models[1].p), 0), ref.p, 1e-9)
