"""GPU parity of LinearDynamicalSystems (K9 persistent filter/smoother + K1/K2/K2a M-step) against golden
fixtures captured from the reference (latent_noise='shared'), and against the CPU oracle on a
Lorenz-like workload (BASELINE config 4 shape at reduced size)."""
import pytest
import torch

from tests.helpers import assert_close
from tests.test_oracle_lds import LDS_CASES, n_iters

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _make(c):
    from pyvbmp_amd.models import LinearDynamicalSystems
    h = int(c["hidden"])
    obs_shape = tuple(int(v) for v in c["obs_shape"])
    batch = tuple(int(v) for v in c["batch_shape"])
    m = LinearDynamicalSystems(obs_shape, h, control_dim=int(c["control"]), regression_dim=int(c["regression"]),
                               latent_noise='shared', batch_shape=batch, device=DEV, dtype=torch.float64)
    m.x0.mu = c["init_x0_mu"].to(DEV)
    m.A.mu = c["init_A_mu"].to(DEV)
    m.obs_model.mu = c["init_obs_mu"].to(DEV)
    m.set_latent_parms()
    m.expand_to_batch = len(batch) > 0
    return m


@pytest.fixture
def smoother_form():
    """K9 has two forms (row-per-lane for few series, lane-per-series for many); the debug switch of the library
    forces one so that both meet the goldens whatever the series count."""
    from pyvbmp_amd import _lib
    lib = _lib.load()
    lib.vbmp_debug_set_flags.argtypes = [__import__("ctypes").c_int]
    lib.vbmp_debug_set_flags.restype = None

    def force(form):
        lib.vbmp_debug_set_flags({"rows": 0x20, "lanes": 0x10, "auto": 0}[form])
    yield force
    lib.vbmp_debug_set_flags(0)


@pytest.mark.parametrize("form", ["rows", "lanes"])
@pytest.mark.parametrize("case", LDS_CASES)
def test_lds_golden(golden, case, form, smoother_form):
    smoother_form(form)
    c = golden("lds")[case]
    m = _make(c)
    lr = float(c["lr"])
    dev = lambda k: c[k].to(DEV) if k in c else None  # noqa: E731
    y, u, r = m.reshape_inputs(dev("y"), dev("u"), dev("r"))
    tol = 1e-10
    for it in range(1, n_iters(c) + 1):
        pre = f"it{it}_"
        m.update_latents(y, u, r)
        for f in ("mu", "Sigma", "invSigma", "invSigmamu"):
            assert_close(getattr(m.px, f), c[pre + "px_" + f], tol, what=pre + f)
        for f in ("SE_x_x", "SE_x0_x0", "SE_x0", "SE_y_xr", "SE_y_y", "SE_xpu_xpu", "SE_x_xpu", "SE_xr_xr", "T", "N",
                  "logZ"):
            assert_close(getattr(m, f), c[pre + f], tol, what=pre + f)
        assert_close(m.ELBO(), c[pre + "ELBO"], tol, what=pre + "ELBO")
        m.ss_update(p=None, lr=lr)
        m.obs_model.ss_update(m.SE_xr_xr, m.SE_y_xr, m.SE_y_y, m.T, lr)
        assert_close(m.x0.mu, c[pre + "x0_mu"], tol)
        assert_close(m.x0.invU.invU, c[pre + "x0_invU"], tol)
        assert_close(m.A.mu, c[pre + "A_mu"], tol)
        assert_close(m.A.invV, c[pre + "A_invV"], tol)
        assert_close(m.A.invU.invU, c[pre + "A_invU_invU"], tol)
        assert_close(m.obs_model.mu, c[pre + "obs_mu"], tol)
        assert_close(m.obs_model.invU.U, c[pre + "obs_invU_U"], tol)
        for f in ("invQ", "ATQA_x_x", "invATQA_x_x", "logdetATQA_x_x", "ATQA_x_u", "ATQA_u_u", "QA_xp_x", "QA_xp_u"):
            assert_close(getattr(m, f), c[pre + f], tol, what=pre + f)
    assert_close(m.KLqprior(), c["KLqprior"], tol)


def lorenz(T, S, gen, dt=0.01, stride=5):
    """Euler-integrated Lorenz-63 trajectories (our own generator, tools/synth.py; same kind of data as simulations/Lorenz.py)."""
    from tools.synth import lorenz as _lorenz
    return _lorenz(T, S, gen, dt, stride, device="cpu")


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 1e-4)])
def test_lds_estep_vs_oracle_lorenz(dtype, tol):
    """config-4 shaped E-step (hidden 6, obs 6) at T=200, 96 series: px.* and logZ against the oracle."""
    from oracle import lds as olds
    from oracle import mnw as omnw
    from oracle import niw as oniw
    from pyvbmp_amd.models import LinearDynamicalSystems
    T, S, h = 200, 96, 6
    g = torch.Generator().manual_seed(4)
    y = lorenz(T, S, g)
    m = LinearDynamicalSystems((6,), h, latent_noise='shared', device=DEV, dtype=dtype)
    x0 = oniw.niw_new((h,), (), mu_init=m.x0.mu.cpu().double())
    A = omnw.mnw_new((h, h + 1), (), mu_init=m.A.mu.cpu().double())
    obs = omnw.mnw_new((6, h + 1), (), mu_init=m.obs_model.mu.cpu().double())
    yy, uu, rr = m.reshape_inputs(y.to(dtype).to(DEV))
    m.update_latents(yy, uu, rr)
    yo, uo, ro = olds.reshape_inputs(y, None, None, (6,), 1, 1)
    sm = olds.smoother(olds.latent_parms(A, h), x0, h, yo, uo, ro, obs, 0)
    for f in ("mu", "Sigma", "invSigma", "invSigmamu"):
        assert_close(getattr(m.px, f), sm[f], tol, what=f)
    st = olds.latent_stats(sm, yo, uo, ro, (6,), 1, 1, (), 0)
    assert_close(m.logZ, st["logZ"], tol, what="logZ")
    assert_close(m.SE_x_xpu, st["SE_x_xpu"], tol, what="SE_x_xpu")
    assert_close(m.SE_xpu_xpu, st["SE_xpu_xpu"], tol, what="SE_xpu_xpu")


def test_lds_elbo_increases_full_vb():
    """a few full VB iterations on Lorenz data run end to end and the ELBO goes up"""
    from pyvbmp_amd.models import LinearDynamicalSystems
    g = torch.Generator().manual_seed(1)
    y = lorenz(120, 64, g).to(DEV)
    m = LinearDynamicalSystems((6,), 6, latent_noise='shared', device=DEV, dtype=torch.float64)
    elbos = []
    for _ in range(6):
        m.update(y, iters=1, lr=1.0)
        elbos.append(float(m.ELBO_last))
    assert all(torch.isfinite(torch.tensor(elbos)))
    assert elbos[-1] > elbos[1]


@pytest.mark.parametrize("case", ["lds_h6_o6", "lds_h3_o5_ctrl_reg", "lds_h4_o5_batch2", "lds_h2_o32"])
def test_lds_composed_smoother_matches_golden(golden, case, monkeypatch):
    """the hidden_dim > 8 route (host loop of K1 launches + GEMMs) replayed on the golden cases by forcing it"""
    from pyvbmp_amd import _lib
    monkeypatch.setattr(_lib, "LDS_MAX_H", 0)
    monkeypatch.setattr(_lib, "LDS_MAX_H_BLOCK", 0)
    c = golden("lds")[case]
    m = _make(c)
    dev = lambda k: c[k].to(DEV) if k in c else None  # noqa: E731
    y, u, r = m.reshape_inputs(dev("y"), dev("u"), dev("r"))
    m.update_latents(y, u, r)
    for f in ("mu", "Sigma", "invSigma", "invSigmamu"):
        assert_close(getattr(m.px, f), c["it1_px_" + f], 1e-10, what=f)
    for f in ("SE_x_x", "SE_x0_x0", "SE_x0", "SE_xpu_xpu", "SE_x_xpu", "SE_xr_xr", "logZ"):
        assert_close(getattr(m, f), c["it1_" + f], 1e-10, what=f)


def test_lds_hidden_12_runs():
    """hidden_dim 12 (> 8): composed route end to end, ELBO finite and increasing"""
    from pyvbmp_amd.models import LinearDynamicalSystems
    g = torch.Generator().manual_seed(2)
    y = lorenz(60, 16, g).to(DEV)
    m = LinearDynamicalSystems((6,), 12, latent_noise='shared', device=DEV, dtype=torch.float64)
    e = []
    for _ in range(4):
        m.update(y, iters=1)
        e.append(float(m.ELBO_last))
    assert all(torch.isfinite(torch.tensor(e))) and e[-1] > e[0]


@pytest.mark.parametrize("noise", ["shared", "independent"])
def test_lds_graphed_update_matches_eager(noise):
    """hipGraph replay of the LDS VB iteration (pyvbmp_amd.graph) against the eager loop on a short Lorenz-like set"""
    from pyvbmp_amd.models import LinearDynamicalSystems
    g = torch.Generator().manual_seed(11)
    y = lorenz(60, 9, g).to(DEV)
    out = []
    for graphed in (False, True):
        torch.manual_seed(2)
        m = LinearDynamicalSystems((6,), 4, latent_noise=noise, device=DEV, dtype=torch.float64)
        m.update(y, iters=6, lr=1.0, graphed=graphed)
        out.append(m)
    a, b = out
    assert_close(b.px.mu, a.px.mu, 1e-8, what="px.mu")
    assert_close(b.px.Sigma, a.px.Sigma, 1e-8, what="px.Sigma")
    assert_close(b.A.mu, a.A.mu, 1e-8, what="A.mu")
    assert_close(b.obs_model.mu, a.obs_model.mu, 1e-8, what="obs.mu")
    assert_close(b.ELBO().sum(), a.ELBO().sum(), 1e-8, what="ELBO")


@pytest.mark.parametrize("form", ["rows", "lanes"])
@pytest.mark.parametrize("h", [1, 2, 3, 5, 7, 8])
def test_lds_smoother_every_hidden_dim_both_forms(h, form, smoother_form):
    """K9 for every hidden dimension it is instantiated for, in both device forms, with a series count that leaves the
    last wave of the row-per-lane form (4 series per wave) and of the lane-per-series form partly empty: px.* and the
    evidence against the CPU oracle."""
    from oracle import lds as olds
    from oracle import mnw as omnw
    from oracle import niw as oniw
    from pyvbmp_amd.models import LinearDynamicalSystems
    smoother_form(form)
    T, S = 40, 13
    g = torch.Generator().manual_seed(100 + h)
    y = lorenz(T, S, g)
    m = LinearDynamicalSystems((6,), h, latent_noise='shared', device=DEV, dtype=torch.float64)
    x0 = oniw.niw_new((h,), (), mu_init=m.x0.mu.cpu())
    A = omnw.mnw_new((h, h + 1), (), mu_init=m.A.mu.cpu())
    obs = omnw.mnw_new((6, h + 1), (), mu_init=m.obs_model.mu.cpu())
    yy, uu, rr = m.reshape_inputs(y.to(DEV))
    m.update_latents(yy, uu, rr)
    yo, uo, ro = olds.reshape_inputs(y, None, None, (6,), 1, 1)
    sm = olds.smoother(olds.latent_parms(A, h), x0, h, yo, uo, ro, obs, 0)
    for f in ("mu", "Sigma", "invSigma", "invSigmamu"):
        assert_close(getattr(m.px, f), sm[f], 1e-10, what=f)
    st = olds.latent_stats(sm, yo, uo, ro, (6,), 1, 1, (), 0)
    assert_close(m.logZ, st["logZ"], 1e-10, what="logZ")
    assert_close(m.SE_x_xpu, st["SE_x_xpu"], 1e-10, what="SE_x_xpu")


@pytest.mark.parametrize("case", LDS_CASES)
def test_lds_block_form_matches_golden(golden, case, smoother_flags):
    """K9's block-per-series form (LDS-resident matrices, meant for 8 < hidden <= 64) forced onto the golden cases"""
    smoother_flags(0x200)
    c = golden("lds")[case]
    m = _make(c)
    dev = lambda k: c[k].to(DEV) if k in c else None  # noqa: E731
    y, u, r = m.reshape_inputs(dev("y"), dev("u"), dev("r"))
    m.update_latents(y, u, r)
    for f in ("mu", "Sigma", "invSigma", "invSigmamu"):
        assert_close(getattr(m.px, f), c["it1_px_" + f], 1e-10, what=f)
    for f in ("SE_x_x", "SE_x0_x0", "SE_x0", "SE_xpu_xpu", "SE_x_xpu", "SE_xr_xr", "logZ"):
        assert_close(getattr(m, f), c["it1_" + f], 1e-10, what=f)


@pytest.mark.parametrize("h,dtype,tol", [(9, torch.float64, 1e-10), (12, torch.float64, 1e-10), (14, torch.float64, 1e-10), (21, torch.float64, 1e-10),
                                         (33, torch.float64, 1e-10), (14, torch.float32, 1e-4),
                                         (52, torch.float64, 1e-10), (60, torch.float64, 1e-10), (61, torch.float64, 1e-10), (12, torch.float32, 1e-4),
                                         (52, torch.float32, 1e-4), (64, torch.float32, 1e-4)])
def test_lds_block_form_vs_oracle_and_composed(h, dtype, tol, monkeypatch):
    """hidden dimensions beyond the register forms (52 = the latent of the flocking DMBD): the block-per-series kernel
    against the CPU oracle (fp64, the reference's algorithm) at the north-star tolerance, and the composed recursion
    (host loop of K1 launches + GEMMs, the route beyond the kernel's sizes) against the same oracle"""
    from oracle import lds as olds
    from oracle import mnw as omnw
    from oracle import niw as oniw
    from pyvbmp_amd import _lib
    from pyvbmp_amd.models import LinearDynamicalSystems
    assert _lib.lds_block_fits(h, 8 if dtype == torch.float64 else 4)
    g = torch.Generator().manual_seed(h)
    y = lorenz(30, 5, g).to(dtype)
    torch.manual_seed(h)
    m = LinearDynamicalSystems((6,), h, latent_noise='shared', device=DEV, dtype=dtype)
    x0 = oniw.niw_new((h,), (), mu_init=m.x0.mu.cpu().double())
    A = omnw.mnw_new((h, h + 1), (), mu_init=m.A.mu.cpu().double())
    obs = omnw.mnw_new((6, h + 1), (), mu_init=m.obs_model.mu.cpu().double())
    yo, uo, ro = olds.reshape_inputs(y.double(), None, None, (6,), 1, 1)
    sm = olds.smoother(olds.latent_parms(A, h), x0, h, yo, uo, ro, obs, 0)
    st = olds.latent_stats(sm, yo, uo, ro, (6,), 1, 1, (), 0)
    yy, uu, rr = m.reshape_inputs(y.to(DEV))
    launched = []
    _lib.launch_hooks = (lambda name: launched.append(name), lambda name: None)
    try:
        m.update_latents(yy, uu, rr)
    finally:
        _lib.launch_hooks = None
    assert launched.count("vbmp_lds_smoother") == 1  # the block form ran
    for f in ("mu", "Sigma", "invSigma", "invSigmamu"):
        assert_close(getattr(m.px, f), sm[f], tol, what=f"block {f}")
    for f in ("SE_x_x", "SE_x_xpu", "logZ"):
        assert_close(getattr(m, f), st[f], tol, what=f"block {f}")
    monkeypatch.setattr(_lib, "LDS_MAX_H_BLOCK", 0)
    m.px = None
    m.update_latents(yy, uu, rr)
    for f in ("mu", "Sigma", "invSigma", "invSigmamu"):
        assert_close(getattr(m.px, f), sm[f], tol, what=f"composed {f}")
    for f in ("SE_x_x", "SE_x_xpu", "logZ"):
        assert_close(getattr(m, f), st[f], tol, what=f"composed {f}")


@pytest.mark.parametrize("form", ["rows", "lanes", "block"])
@pytest.mark.parametrize("T", [1, 2, 3])
def test_lds_smoother_short_series(T, form, smoother_form, smoother_flags):
    """one, two and three time steps: the sweeps' peeled first / last steps and the smoothed covariance that every
    device form carries over to the following step (the first pending one is the filtered posterior of the last step)"""
    from oracle import lds as olds
    from oracle import mnw as omnw
    from oracle import niw as oniw
    from pyvbmp_amd.models import LinearDynamicalSystems
    if form == "block":
        smoother_flags(0x200)
    else:
        smoother_form(form)
    h, S = 3, 6
    g = torch.Generator().manual_seed(7 + T)
    y = lorenz(max(T, 4), S, g)[:T]
    m = LinearDynamicalSystems((6,), h, latent_noise='shared', device=DEV, dtype=torch.float64)
    x0 = oniw.niw_new((h,), (), mu_init=m.x0.mu.cpu())
    A = omnw.mnw_new((h, h + 1), (), mu_init=m.A.mu.cpu())
    obs = omnw.mnw_new((6, h + 1), (), mu_init=m.obs_model.mu.cpu())
    yy, uu, rr = m.reshape_inputs(y.to(DEV))
    m.update_latents(yy, uu, rr)
    yo, uo, ro = olds.reshape_inputs(y, None, None, (6,), 1, 1)
    sm = olds.smoother(olds.latent_parms(A, h), x0, h, yo, uo, ro, obs, 0)
    for f in ("mu", "Sigma", "invSigma", "invSigmamu"):
        assert_close(getattr(m.px, f), sm[f], 1e-10, what=f)
    st = olds.latent_stats(sm, yo, uo, ro, (6,), 1, 1, (), 0)
    for f in ("logZ", "SE_x_x", "SE_x_xpu", "SE_x0_x0"):
        assert_close(getattr(m, f), st[f], 1e-10, what=f)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 1e-4)])
@pytest.mark.parametrize("T", [1, 2])
def test_lds_single_steps_meet_the_north_star_tolerance(T, dtype, tol):
    """ONE forward step (T = 1: predict + update from the prior) and one forward + one backward step (T = 2) of the K9
    recursion (ref models/LinearDynamicalSystems.py:268-330) against the oracle at the north-star tolerance, 1e-10 in
    fp64 and 1e-4 in fp32 -- the looser bounds of the long recursions elsewhere in this file are the recursion's own
    error growth (tools/exp/lds_error_growth.py), not the kernel's per-step accuracy"""
    from oracle import lds as olds
    from oracle import mnw as omnw
    from oracle import niw as oniw
    from pyvbmp_amd.models import LinearDynamicalSystems
    S, h = 64, 6
    g = torch.Generator().manual_seed(40 + T)
    y = lorenz(T + 3, S, g)[3:]
    torch.manual_seed(12)
    m = LinearDynamicalSystems((6,), h, latent_noise='shared', device=DEV, dtype=dtype)
    x0 = oniw.niw_new((h,), (), mu_init=m.x0.mu.cpu().double())
    A = omnw.mnw_new((h, h + 1), (), mu_init=m.A.mu.cpu().double())
    obs = omnw.mnw_new((6, h + 1), (), mu_init=m.obs_model.mu.cpu().double())
    yy, uu, rr = m.reshape_inputs(y.to(dtype).to(DEV))
    m.update_latents(yy, uu, rr)
    yo, uo, ro = olds.reshape_inputs(y.to(dtype).double(), None, None, (6,), 1, 1)
    sm = olds.smoother(olds.latent_parms(A, h), x0, h, yo, uo, ro, obs, 0)
    for f in ("mu", "Sigma", "invSigma", "invSigmamu"):
        assert_close(getattr(m.px, f), sm[f], tol, what=f"{f} T={T}")
    st = olds.latent_stats(sm, yo, uo, ro, (6,), 1, 1, (), 0)
    assert_close(m.logZ, st["logZ"], tol, what="logZ")


@pytest.mark.parametrize("mode", ["exact", "default"])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("h,T,S", [(6, 300, 13), (3, 120, 5), (8, 90, 4), (6, 3, 4), (6, 40, 9)])
def test_lds_fixed_point_shortcut_against_the_full_recursion(h, T, S, dtype, mode, smoother_flags):
    """K9 stops its matrix recursions once they have converged (time-independent likelihood precision) and runs only the
    mean recursion from there.  Flag 0x800 stops only at a BITWISE repeat: every output must then be bit for bit what the
    full recursion (flag 0x8000) produces.  The default also stops when the precision has moved by <= 4 ulp for 8 steps in
    a row (a last-bit limit cycle of the full recursion): outputs agree to rounding (1e-13 fp64 / 2e-6 fp32, normwise)."""
    from pyvbmp_amd.models import LinearDynamicalSystems
    g = torch.Generator().manual_seed(h * 1000 + T)
    y = lorenz(T, S, g).to(dtype).to(DEV)
    outs = []
    for flag in (0x8000, 0x800 if mode == "exact" else 0):
        smoother_flags(flag)
        torch.manual_seed(3)
        m = LinearDynamicalSystems((6,), h, latent_noise='shared', device=DEV, dtype=dtype)
        m.update_latents(*m.reshape_inputs(y))
        outs.append({f: getattr(m.px, f).clone() for f in ("mu", "Sigma", "invSigma", "invSigmamu")} |
                    {f: getattr(m, f).clone() for f in ("logZ", "SE_x_x", "SE_x_xpu", "SE_xpu_xpu", "SE_x0_x0", "SE_x0")})
    full, short = outs
    tol = 1e-13 if dtype == torch.float64 else 2e-6
    for k in full:
        if mode == "exact":
            assert torch.equal(full[k], short[k]), f"{k}: max abs diff {float((full[k] - short[k]).abs().max()):.3e}"
        else:
            assert_close(short[k], full[k], tol, what=k)
    if mode == "default" and T >= 90:  # the shortcut was taken: the smoothed covariance is constant in the middle
        mid = short["Sigma"][T // 2 - 2:T // 2 + 2]
        assert torch.equal(mid[0], mid[1]) and torch.equal(mid[1], mid[2])


@pytest.mark.parametrize("form", ["rows", "lanes"])
@pytest.mark.parametrize("T", [3, 60, 400])
def test_lds_cross_covariances_dense_and_work_buffer_modes(T, form, smoother_form):
    """forward_backward_loop returns every Sigma_t_tp1[t] like the reference (:361-381); update_latents asks the kernel only for
    slot T-1, the time sums and sum_t logZ (vbmp_lds_args.flags = VBMP_LDS_CROSS_WORK | VBMP_LDS_LOGZ_SUM).  Dense mode: all slots against the oracle; work
    buffer mode: slot T-1 and every other output identical to the dense run."""
    from oracle import lds as olds
    from oracle import mnw as omnw
    from oracle import niw as oniw
    from pyvbmp_amd.models import LinearDynamicalSystems
    smoother_form(form)
    h, S = 6, 9
    y = lorenz(T, S, torch.Generator().manual_seed(T))
    m = LinearDynamicalSystems((6,), h, latent_noise='shared', device=DEV, dtype=torch.float64)
    x0 = oniw.niw_new((h,), (), mu_init=m.x0.mu.cpu())
    A = omnw.mnw_new((h, h + 1), (), mu_init=m.A.mu.cpu())
    obs = omnw.mnw_new((6, h + 1), (), mu_init=m.obs_model.mu.cpu())
    yy, uu, rr = m.reshape_inputs(y.to(DEV))
    m.px = None
    from pyvbmp_amd.dists import MultivariateNormal_vector_format
    m.px = MultivariateNormal_vector_format(mu=torch.zeros(tuple(yy.shape[:-2]) + (h, 1), device=DEV, dtype=torch.float64))
    cross, S00, m0, logZ, _ = m.forward_backward_loop(yy, uu, rr)
    dense = {f: getattr(m.px, f).clone() for f in ("mu", "Sigma", "invSigma", "invSigmamu")}
    sums = tuple(t.clone() for t in m._time_sums)
    yo, uo, ro = olds.reshape_inputs(y, None, None, (6,), 1, 1)
    sm = olds.smoother(olds.latent_parms(A, h), x0, h, yo, uo, ro, obs, 0)
    assert_close(cross, sm["Sigma_t_tp1"], 1e-10, what="Sigma_t_tp1")
    cross2, S002, m02, logZ2, _ = m.forward_backward_loop(yy, uu, rr, sums_only=True)
    assert torch.equal(cross2[-1], cross[-1]) and torch.equal(S002, S00) and torch.equal(m02, m0)
    assert logZ2.shape[0] == 1
    assert_close(logZ2[0], logZ.sum(0), 1e-13, what="sum_t logZ")
    assert_close(logZ, torch.as_tensor(sm["logZ"]).reshape(logZ.shape), 1e-10, what="logZ") if "logZ" in sm else None
    for f, v in dense.items():
        assert torch.equal(getattr(m.px, f), v), f
    for a, b in zip(m._time_sums, sums):
        assert torch.equal(a, b)


@pytest.mark.parametrize("mode", ["exact", "default"])
def test_lds_fixed_point_shortcut_with_time_varying_controls(mode, smoother_flags):
    """control and regression inputs that change every step leave the matrix recursion time-independent: the shortcut runs with
    per-step control terms in its request ring (the instance whose controls are NOT hoisted).  Against the oracle at 1e-10, and
    against the full recursion (bitwise under flag 0x800)."""
    from oracle import lds as olds
    from oracle import mnw as omnw
    from oracle import niw as oniw
    from pyvbmp_amd.models import LinearDynamicalSystems
    T, S, h = 150, 7, 4
    g = torch.Generator().manual_seed(77)
    y = lorenz(T, S, g)[..., :5].contiguous()
    u = torch.randn(T, S, 2, generator=g, dtype=torch.float64)
    r = torch.randn(T, S, 1, generator=g, dtype=torch.float64)
    outs = []
    for flag in (0x8000, 0x800 if mode == "exact" else 0):
        smoother_flags(flag)
        torch.manual_seed(5)
        m = LinearDynamicalSystems((5,), h, control_dim=2, regression_dim=1, latent_noise='shared', device=DEV, dtype=torch.float64)
        m.update_latents(*m.reshape_inputs(y.to(DEV), u.to(DEV), r.to(DEV)))
        outs.append({f: getattr(m.px, f).clone() for f in ("mu", "Sigma", "invSigma", "invSigmamu")} |
                    {f: getattr(m, f).clone() for f in ("logZ", "SE_x_x", "SE_x_xpu", "SE_xpu_xpu", "SE_x0_x0", "SE_x0")})
    full, short = outs
    for k in full:
        if mode == "exact":
            assert torch.equal(full[k], short[k]), f"{k}: max abs diff {float((full[k] - short[k]).abs().max()):.3e}"
        else:
            assert_close(short[k], full[k], 1e-13, what=k)
    mid = short["Sigma"][T // 2 - 2:T // 2 + 2]
    if mode == "default":
        assert torch.equal(mid[0], mid[1]) and torch.equal(mid[1], mid[2])  # the shortcut was taken
    x0 = oniw.niw_new((h,), (), mu_init=m.x0.mu.cpu())
    A = omnw.mnw_new((h, h + 3), (), mu_init=m.A.mu.cpu())
    obs = omnw.mnw_new((5, h + 2), (), mu_init=m.obs_model.mu.cpu())
    yo, uo, ro = olds.reshape_inputs(y, u, r, (5,), 3, 2)
    sm = olds.smoother(olds.latent_parms(A, h), x0, h, yo, uo, ro, obs, 0)
    for f in ("mu", "Sigma", "invSigma", "invSigmamu"):
        assert_close(short[f], sm[f], 1e-10, what=f)
    st = olds.latent_stats(sm, yo, uo, ro, (5,), 3, 2, (), 0)
    assert_close(short["logZ"], st["logZ"], 1e-10, what="logZ")
    assert_close(short["SE_x_xpu"], st["SE_x_xpu"], 1e-10, what="SE_x_xpu")


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 1e-4)])
def test_lds_fixed_point_shortcut_on_a_slowly_converging_system(dtype, tol, smoother_flags):
    """near-unit-root transition and a weakly informative observation model: the Riccati recursion creeps for hundreds of steps.
    The shortcut must not freeze a sequence that still drifts: every output against the fp64 oracle at the north-star tolerance,
    and against the full recursion."""
    from oracle import lds as olds
    from oracle import mnw as omnw
    from oracle import niw as oniw
    from pyvbmp_amd.models import LinearDynamicalSystems
    T, S, h = 700, 6, 4
    g = torch.Generator().manual_seed(11)
    y = 0.3 * lorenz(T, S, g)[..., :3].contiguous()
    torch.manual_seed(2)
    m = LinearDynamicalSystems((3,), h, latent_noise='shared', device=DEV, dtype=dtype)
    A0 = torch.cat((0.9995 * torch.eye(h, dtype=dtype), torch.zeros(h, 1, dtype=dtype)), -1)
    m.A.mu = A0.to(DEV).reshape(m.A.mu.shape)
    m.obs_model.mu = (0.02 * torch.randn(m.obs_model.mu.shape, generator=g, dtype=torch.float64)).to(dtype).to(DEV)
    qs = 1e-4  # process-noise scale: E[invQ] = nu U grows by 1 / qs
    W = m.A.invU
    W.invU, W.U, W.logdet_invU = W.invU * qs, W.U / qs, W.logdet_invU + h * float(torch.log(torch.tensor(qs)))
    m.set_latent_parms()
    outs = []
    for flag in (0x8000, 0):
        smoother_flags(flag)
        m.update_latents(*m.reshape_inputs(y.to(dtype).to(DEV)))
        outs.append({f: getattr(m.px, f).clone() for f in ("mu", "Sigma", "invSigma", "invSigmamu")} |
                    {f: getattr(m, f).clone() for f in ("logZ", "SE_x_x", "SE_x_xpu")})
    full, short = outs
    x0 = oniw.niw_new((h,), (), mu_init=m.x0.mu.cpu().double())
    A = omnw.mnw_new((h, h + 1), (), mu_init=m.A.mu.cpu().double())
    obs = omnw.mnw_new((3, h + 1), (), mu_init=m.obs_model.mu.cpu().double())
    A["W"] = dict(A["W"])
    A["W"]["invU"], A["W"]["U"] = A["W"]["invU"] * qs, A["W"]["U"] / qs
    A["W"]["logdet_invU"] = A["W"]["logdet_invU"] + h * float(torch.log(torch.tensor(qs)))
    yo, uo, ro = olds.reshape_inputs(y, None, None, (3,), 1, 1)
    sm = olds.smoother(olds.latent_parms(A, h), x0, h, yo, uo, ro, obs, 0)
    st = olds.latent_stats(sm, yo, uo, ro, (3,), 1, 1, (), 0)
    from tests.helpers import relerr
    errs = {f: (relerr(short[f], sm[f]), relerr(full[f], sm[f])) for f in ("mu", "Sigma", "invSigma", "invSigmamu")}
    errs["logZ"] = (relerr(short["logZ"], st["logZ"]), relerr(full["logZ"], st["logZ"]))
    errs["SE_x_xpu"] = (relerr(short["SE_x_xpu"], st["SE_x_xpu"]), relerr(full["SE_x_xpu"], st["SE_x_xpu"]))
    print("normwise errors against the fp64 oracle (shortcut, full recursion):", {k: (f"{a:.1e}", f"{b:.1e}") for k, (a, b) in errs.items()})
    for k, (a, b) in errs.items():
        # the shortcut may not be further from the oracle than the north-star tolerance, or than the full recursion in the same
        # precision is (an ill-conditioned system in fp32)
        assert a <= max(tol, 2.0 * b), f"{k}: shortcut {a:.2e}, full recursion {b:.2e}"
    # how slowly: the filtered covariance of the full recursion is still moving after 100 steps
    d = (full["Sigma"][T // 2 + 1] - full["Sigma"][T // 2]).abs().max() / full["Sigma"][T // 2].abs().max()
    d100 = (full["Sigma"][101] - full["Sigma"][100]).abs().max() / full["Sigma"][100].abs().max()
    assert float(d100) > (1e-13 if dtype == torch.float64 else 1e-6), f"not a slow system: {float(d100):.2e} at t = 100, {float(d):.2e} at T / 2"


@pytest.mark.parametrize("mode", ["exact", "default"])
def test_lds_fixed_point_shortcut_with_a_batch_of_systems(mode, smoother_flags):
    """two systems with different parameters share every wave (series s belongs to system s % 2): each 16-lane row freezes on its
    own recursion.  Long horizon so that the shortcut is taken; against the oracle and the full recursion."""
    from oracle import lds as olds
    from oracle import mnw as omnw
    from oracle import niw as oniw
    from pyvbmp_amd.models import LinearDynamicalSystems
    T, S, h, NB = 260, 5, 4, 2
    g = torch.Generator().manual_seed(41)
    y = lorenz(T, S, g)[..., :5].contiguous()
    outs = []
    for flag in (0x8000, 0x800 if mode == "exact" else 0):
        smoother_flags(flag)
        torch.manual_seed(9)
        m = LinearDynamicalSystems((5,), h, latent_noise='shared', batch_shape=(NB,), device=DEV, dtype=torch.float64)
        m.expand_to_batch = True
        m.update_latents(*m.reshape_inputs(y.to(DEV)))
        outs.append({f: getattr(m.px, f).clone() for f in ("mu", "Sigma", "invSigma", "invSigmamu")} |
                    {f: getattr(m, f).clone() for f in ("logZ", "SE_x_x", "SE_x_xpu", "SE_y_xr")})
    full, short = outs
    for k in full:
        if mode == "exact":
            assert torch.equal(full[k], short[k]), f"{k}: max abs diff {float((full[k] - short[k]).abs().max()):.3e}"
        else:
            assert_close(short[k], full[k], 1e-13, what=k)
    if mode == "default":
        mid = short["Sigma"][T // 2 - 2:T // 2 + 2]
        assert torch.equal(mid[0], mid[1]) and torch.equal(mid[1], mid[2])  # the shortcut was taken
        assert not torch.equal(mid[0][:, 0], mid[0][:, 1])                  # ... and the two systems differ
    x0 = oniw.niw_new((h,), (NB,), mu_init=m.x0.mu.cpu())
    A = omnw.mnw_new((h, h + 1), (NB,), mu_init=m.A.mu.cpu())
    obs = omnw.mnw_new((5, h + 1), (NB,), mu_init=m.obs_model.mu.cpu())
    yo, uo, ro = olds.reshape_inputs(y, None, None, (5,), 1, 1, batch_shape=(NB,), expand_to_batch=True)
    sm = olds.smoother(olds.latent_parms(A, h), x0, h, yo, uo, ro, obs, 0)
    for f in ("mu", "Sigma", "invSigma", "invSigmamu"):
        assert_close(short[f], sm[f], 1e-10, what=f)
    st = olds.latent_stats(sm, yo, uo, ro, (5,), 1, 1, (NB,), 0)
    for f in ("logZ", "SE_x_xpu", "SE_y_xr"):
        assert_close(short[f], st[f], 1e-10, what=f)


@pytest.mark.parametrize("dtype,tol,tol_modes", [(torch.float64, 1e-10, 1e-13), (torch.float32, 1e-4, 2e-6)])
def test_lds_estep_full_size_config4_properties(dtype, tol, tol_modes):
    """BASELINE configs[3] at FULL size -- T = 1000, 4096 series, hidden 6, Lorenz data (ref models/LinearDynamicalSystems.py:156-216,
    332-383) -- through update_latents, in the default (fixed-point shortcut) and in the literal-recursion mode.  At this size the
    shortcut's freeze point, the 4-deep request ring's tail and the flat-store transposition run over >= 1000 steps / 1024 waves.
    Size-independent properties on everything (Sigma symmetric, positive definite, invSigma Sigma = I, finite evidence), the two
    modes against each other on everything, and every 257-th series against the fp64 oracle at the north-star tolerance."""
    from oracle import lds as olds
    from oracle import mnw as omnw
    from oracle import niw as oniw
    from pyvbmp_amd import ops
    from pyvbmp_amd.models import LinearDynamicalSystems
    T, S, h = 1000, 4096, 6
    y = lorenz(T, S, torch.Generator().manual_seed(40))
    torch.manual_seed(12)
    m = LinearDynamicalSystems((6,), h, latent_noise='shared', device=DEV, dtype=dtype)
    yy, uu, rr = m.reshape_inputs(y.to(dtype).to(DEV))
    sel = torch.arange(0, S, 257)
    x0 = oniw.niw_new((h,), (), mu_init=m.x0.mu.cpu().double())
    A = omnw.mnw_new((h, h + 1), (), mu_init=m.A.mu.cpu().double())
    obs = omnw.mnw_new((6, h + 1), (), mu_init=m.obs_model.mu.cpu().double())
    yo, uo, ro = olds.reshape_inputs(y[:, sel].contiguous(), None, None, (6,), 1, 1)
    sm = olds.smoother(olds.latent_parms(A, h), x0, h, yo, uo, ro, obs, 0)
    keep = {}
    for mode in ("auto", "off"):
        m.fixed_point = mode
        m.update_latents(yy, uu, rr)
        px = m.px
        assert px.Sigma.shape == (T, S, h, h) and px.mu.shape == (T, S, h, 1)
        for f in ("mu", "Sigma", "invSigma", "invSigmamu"):
            assert bool(torch.isfinite(getattr(px, f)).all()), f"{mode}: {f} not finite"
            assert_close(getattr(px, f)[:, sel.to(DEV)], sm[f], tol, what=f"{mode}: {f} on every 257-th series")
        sym = float((px.Sigma - px.Sigma.transpose(-2, -1)).abs().max() / px.Sigma.abs().max())
        assert sym <= (1e-13 if dtype == torch.float64 else 1e-5), f"{mode}: Sigma asymmetric by {sym:.2e}"
        eye = torch.eye(h, dtype=dtype, device=DEV)
        res = float(((px.invSigma @ px.Sigma) - eye).abs().max())
        assert res <= (1e-9 if dtype == torch.float64 else 2e-3), f"{mode}: invSigma Sigma - I = {res:.2e}"
        # positive definite: the product's own batched elimination (K1) counts the matrices with a non-positive pivot
        cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
        ops.spd_inv_logdet(px.Sigma.reshape(-1, h, h), nonspd=cnt)
        assert int(cnt) == 0, f"{mode}: {int(cnt)} smoothed covariances are not positive definite"
        assert bool(torch.isfinite(m.logZ).all())
        keep[mode] = {f: getattr(px, f).clone() for f in ("mu", "Sigma")} | {f: getattr(m, f).clone() for f in ("logZ", "SE_x_x", "SE_x_xpu", "SE_xpu_xpu", "SE_x0_x0")}
        m.px = None
    for k in keep["auto"]:
        assert_close(keep["auto"][k], keep["off"][k], tol_modes, what=f"shortcut against the literal recursion: {k}")
    mid = keep["auto"]["Sigma"][T // 2 - 1:T // 2 + 2, 0]
    assert torch.equal(mid[0], mid[1]) and torch.equal(mid[1], mid[2])  # the shortcut was taken


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 1e-4)])
@pytest.mark.parametrize("one_minus_rate", [3e-1, 3e-2, 3e-3, 3e-4, 1e-6])
def test_lds_fixed_point_shortcut_contraction_sweep_against_the_oracle(one_minus_rate, dtype, tol):
    """sweep the contraction rate of the Riccati recursion (transition (1 - x) I, little process noise, weak
    observations) from fast to near-unit-root.  The default mode may take the rounding criterion only while the recursion is
    seen to contract (csrc/k_lds_g16.inc), so: (i) against the fp64 ORACLE every output meets the north-star tolerance or is as
    close as the literal recursion in the same precision is (an ill-conditioned system in fp32 is ill-conditioned for both);
    (ii) exact mode == literal recursion BIT FOR BIT, default mode within the contract's last-bit wander of it (1e-13 / 2e-6), however
    slowly the recursion contracts."""
    from oracle import lds as olds
    from oracle import mnw as omnw
    from oracle import niw as oniw
    from pyvbmp_amd.models import LinearDynamicalSystems
    from tests.helpers import relerr
    T, S, h = 600, 5, 4
    g = torch.Generator().manual_seed(21)
    y = 0.3 * lorenz(T, S, g)[..., :3].contiguous()
    torch.manual_seed(4)
    m = LinearDynamicalSystems((3,), h, latent_noise='shared', device=DEV, dtype=dtype)
    A0 = torch.cat(((1.0 - one_minus_rate) * torch.eye(h, dtype=dtype), torch.zeros(h, 1, dtype=dtype)), -1)
    m.A.mu = A0.to(DEV).reshape(m.A.mu.shape)
    m.obs_model.mu = (0.05 * torch.randn(m.obs_model.mu.shape, generator=g, dtype=torch.float64)).to(dtype).to(DEV)
    qs = 1e-3
    W = m.A.invU
    W.invU, W.U, W.logdet_invU = W.invU * qs, W.U / qs, W.logdet_invU + h * float(torch.log(torch.tensor(qs)))
    m.set_latent_parms()
    outs = {}
    for mode in ("off", "exact", "auto"):
        m.fixed_point = mode
        m.update_latents(*m.reshape_inputs(y.to(dtype).to(DEV)))
        outs[mode] = {f: getattr(m.px, f).clone() for f in ("mu", "Sigma", "invSigma", "invSigmamu")} | \
                     {f: getattr(m, f).clone() for f in ("logZ", "SE_x_x", "SE_x_xpu")}
    x0 = oniw.niw_new((h,), (), mu_init=m.x0.mu.cpu().double())
    A = omnw.mnw_new((h, h + 1), (), mu_init=m.A.mu.cpu().double())
    obs = omnw.mnw_new((3, h + 1), (), mu_init=m.obs_model.mu.cpu().double())
    A["W"] = dict(A["W"])
    A["W"]["invU"], A["W"]["U"] = A["W"]["invU"] * qs, A["W"]["U"] / qs
    A["W"]["logdet_invU"] = A["W"]["logdet_invU"] + h * float(torch.log(torch.tensor(qs)))
    yo, uo, ro = olds.reshape_inputs(y, None, None, (3,), 1, 1)
    sm = olds.smoother(olds.latent_parms(A, h), x0, h, yo, uo, ro, obs, 0)
    st = olds.latent_stats(sm, yo, uo, ro, (3,), 1, 1, (), 0)
    ref = {f: sm[f] for f in ("mu", "Sigma", "invSigma", "invSigmamu")} | {"logZ": st["logZ"], "SE_x_xpu": st["SE_x_xpu"]}
    for k, r in ref.items():
        e_auto, e_off = relerr(outs["auto"][k], r), relerr(outs["off"][k], r)
        assert e_auto <= max(tol, 1.25 * e_off), f"{k}: default mode {e_auto:.2e} from the oracle, literal recursion {e_off:.2e}"
        assert torch.equal(outs["exact"][k], outs["off"][k]), f"{k}: exact mode differs from the literal recursion"
    # The default against the literal recursion.  `move` is what the literal recursion's smoothed covariance still does per step in
    # the middle of the series: a last-bit wander where it has converged, rounding noise of the information-form update where the
    # process noise is small (the update subtracts terms ~1 / qs: ~1e3 ulp here), or a genuine creep.  A recursion stopped while it
    # creeps at rate r sits move / (1 - r) from the literal one -- far beyond the bound below for the slow systems of this sweep.
    Sg = outs["off"]["Sigma"]
    move = float((Sg[T // 2 + 1] - Sg[T // 2]).abs().max() / Sg[T // 2].abs().max())
    eps = 2.2e-16 if dtype == torch.float64 else 1.2e-7
    diffs = {k: relerr(outs["auto"][k], outs["off"][k]) for k in outs["off"]}
    print(f"1 - rate = {one_minus_rate:g}: literal recursion moves {move:.1e} per step at T / 2; default against literal:",
          {k: f"{v:.1e}" for k, v in diffs.items()})
    for k, v in diffs.items():
        assert v <= max(64 * eps, 16 * move), f"{k}: default mode {v:.2e} from the literal recursion, which moves {move:.1e} per step"
    # ... and never further than the accuracy contract of the default mode (include/vbmp_hip.h): the literal recursion's own last-bit
    # wander -- also where the smoothed covariance still creeps at T / 2 (the near-unit-root systems: there the forward filter may have
    # converged and stopped while the backward recursion, which has not, runs on in full)
    bound = 1e-13 if dtype == torch.float64 else 2e-6
    for k, v in diffs.items():
        assert v <= bound, f"{k}: default mode {v:.2e} from the literal recursion (contract {bound:.0e})"
