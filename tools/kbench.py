#!/usr/bin/env python3
"""Kernel micro-benchmarks (GPU box): average/min duration of single kernel launches measured with HIP
events at the launch, plus the implied fraction of the 8 TB/s HBM peak.  Development aid.

    python tools/kbench.py niw [--B 1000000] [--reps 30]
"""
import argparse
import time
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

from bench import algorithmic_bytes_per_update, make_inputs  # noqa: E402
from pyvbmp_amd import _lib, ops  # noqa: E402


def timed(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    ev = []
    _lib.launch_hooks = (lambda n: ev.append(_rec()), lambda n: ev.append(_rec()))
    for _ in range(reps):
        fn()
    _lib.launch_hooks = None
    torch.cuda.synchronize()
    ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(0, len(ev), 2)]
    per = len(ts) // reps
    tot = [sum(ts[i * per:(i + 1) * per]) for i in range(reps)]
    tot.sort()
    return sum(tot) / len(tot), tot[0], tot[len(tot) // 2]


def _rec():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def bench_niw(args):
    for dt, D, lr in [(torch.float64, 16, 1.0), (torch.float32, 16, 1.0), (torch.float64, 16, 0.5),
                      (torch.float64, 8, 1.0), (torch.float64, 32, 1.0), (torch.float32, 32, 1.0),
                      (torch.float64, 6, 1.0), (torch.float64, 2, 1.0), (torch.float64, 64, 1.0)]:
        B = args.B if D <= 16 else args.B // (D * D // 256)
        SExx, SEx, N = make_inputs(B, D, dt, "cuda")
        from pyvbmp_amd.dists import NormalInverseWishart
        q = NormalInverseWishart((D,), (B,), device="cuda", dtype=dt)
        q.ss_update(SExx, SEx, N, lr=1.0, beta=None)
        avg, mn, med = timed(lambda: q.ss_update(SExx, SEx, N, lr=lr, beta=None), args.reps)
        bpu = algorithmic_bytes_per_update(D, SExx.element_size(), lr)
        print(f"niw_ss_update {str(dt)[6:]:8s} D={D:3d} B={B:8d} lr={lr}: avg {avg:.4f} ms  min {mn:.4f}  med {med:.4f}"
              f"  -> {bpu * B / med / 1e6:8.1f} GB/s ({bpu * B / med / 1e6 / 80:.1f}% of 8 TB/s)", flush=True)
        del q, SExx, SEx, N
        torch.cuda.empty_cache()


def bench_copy(args):
    n = 6432 * args.B // 16
    a = torch.empty(n, 2, dtype=torch.float64, device="cuda").normal_()
    b = torch.empty_like(a)
    st, en = _rec(), None
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.reps):
        s = _rec()
        b.copy_(a)
        e = _rec()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    print(f"torch copy of {a.numel() * 8 / 1e9:.2f} GB: med {ts[len(ts) // 2]:.4f} ms -> "
          f"{2 * a.numel() * 8 / ts[len(ts) // 2] / 1e6:.1f} GB/s (read+write)")


def _time_call(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s = _rec()
        fn()
        e = _rec()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]


def bench_mnw(args):
    """BASELINE config 3: MatrixNormalWishart forward / backward / update, N=262144 messages, n=p=32, fp32."""
    from pyvbmp_amd.dists import MultivariateNormal_vector_format as VF
    from pyvbmp_amd.transforms import MatrixNormalWishart
    N, n, p = args.N, 32, 32
    dt = torch.float32
    g = torch.Generator(device="cuda").manual_seed(0)
    m = MatrixNormalWishart((n, p), (), device="cuda", dtype=dt)
    X = torch.randn(4096, p, 1, generator=g, device="cuda", dtype=dt)
    Y = torch.randn(n, p, generator=g, device="cuda", dtype=dt) @ X + 0.3 * torch.randn(4096, n, 1, generator=g, device="cuda", dtype=dt)
    m.raw_update(X, Y)

    def spd(d):
        A = torch.randn(N, d, d + 4, generator=g, device="cuda", dtype=dt)
        return A @ A.transpose(-2, -1) / (d + 4) + 0.5 * torch.eye(d, device="cuda", dtype=dt)
    Px, ex = spd(p), torch.randn(N, p, 1, generator=g, device="cuda", dtype=dt)
    Py, ey = spd(n), torch.randn(N, n, 1, generator=g, device="cuda", dtype=dt)
    bpm = (p * p + p + n * n + n + 1) * 4
    t = _time_call(lambda: m.forward(VF(invSigma=Px, invSigmamu=ex)))
    print(f"mnw.forward  N={N} n=p=32 fp32: {t:.3f} ms -> {N / t * 1e3:.3e} msgs/s, {bpm * N / t / 1e6:.1f} GB/s alg ({bpm * N / t / 1e6 / 80:.1f}% of 8 TB/s)", flush=True)
    t = _time_call(lambda: m.backward(VF(invSigma=Py, invSigmamu=ey)))
    print(f"mnw.backward N={N} n=p=32 fp32: {t:.3f} ms -> {N / t * 1e3:.3e} msgs/s, {bpm * N / t / 1e6:.1f} GB/s alg ({bpm * N / t / 1e6 / 80:.1f}% of 8 TB/s)", flush=True)
    Sx = torch.linalg.inv(Px).contiguous()  # LU inverse comes back column-major; messages on the path are row-major
    mux = Sx @ ex
    Sy = torch.linalg.inv(Py).contiguous()
    muy = Sy @ ey
    bpu = (p * p + p + n * n + n) * 4
    t = _time_call(lambda: m.update(VF(mu=mux, Sigma=Sx), VF(mu=muy, Sigma=Sy)))
    print(f"mnw.update   N={N} n=p=32 fp32: {t:.3f} ms -> {N / t * 1e3:.3e} samples/s, {bpu * N / t / 1e6:.1f} GB/s alg ({bpu * N / t / 1e6 / 80:.1f}% of 8 TB/s)", flush=True)
    t = _time_call(lambda: m.raw_update(mux, muy))
    print(f"mnw.raw_update N={N}: {t:.3f} ms -> {N / t * 1e3:.3e} samples/s", flush=True)


def bench_lds(args):
    """BASELINE config 4: LDS E-step, T=1000, 4096 series, hidden 6, obs 6 (Lorenz-like data)."""
    from pyvbmp_amd.models import LinearDynamicalSystems
    T, S, h = args.T, args.S, 6
    for dt in (torch.float64, torch.float32):
        g = torch.Generator(device="cuda").manual_seed(0)
        y = torch.randn(T, S, 6, generator=g, device="cuda", dtype=dt).cumsum(0) * 0.05
        m = LinearDynamicalSystems((6,), h, latent_noise='shared', device="cuda", dtype=dt)
        yy, uu, rr = m.reshape_inputs(y)
        t_e = _time_call(lambda: m.update_latents(yy, uu, rr), reps=3, warm=1)
        ev = []
        _lib.launch_hooks = (lambda n: ev.append((n, _rec())), lambda n: ev.append((n, _rec())))
        m.update_latents(yy, uu, rr)          # sums-only outputs (what the E-step asks for)
        m.forward_backward_loop(yy, uu, rr)   # the reference's dense outputs
        _lib.launch_hooks = None
        torch.cuda.synchronize()
        tk = [ev[i][1].elapsed_time(ev[i + 1][1]) for i in range(0, len(ev), 2) if ev[i][0] == "vbmp_lds_smoother"]
        b = 8 if dt == torch.float64 else 4
        mini = 720 // 8 * b
        print(f"lds E-step {str(dt)[6:]} T={T} S={S} h=6 (random-walk data): update_latents {t_e:.2f} ms; smoother kernel {tk[0]:.2f} ms "
              f"(dense outputs {tk[1]:.2f} ms) -> {T * S / tk[0] * 1e3:.3e} (t,series)/s; minimal-I/O {mini * T * S / tk[0] / 1e6:.1f} GB/s "
              f"({mini * T * S / tk[0] / 1e6 / 80:.2f}% of 8 TB/s)", flush=True)


from synth import boids  # noqa: E402  (tools/synth.py)


def bench_dmbd(args):
    """BASELINE config 5 (one GPU's share): DMBD with the Flocking_example hyper-parameters (6 objects, hidden 52,
    25 roles) on synthetic boids data, and with the Lorenz hyper-parameters (hidden 6)."""
    from pyvbmp_amd.models import DynamicMarkovBlanketDiscovery
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, (T, S, n_obs, od, roles, hid, nobj) in {
            "flocking (hidden 52, 25 roles)": (100, 20, 12, 4, (1, 2, 2), (4, 4, 4), 6),
            "lorenz-like (hidden 6, 4 roles)": (400, 64, 1, 6, (1, 2, 1), (2, 2, 2), 1)}.items():
        y = boids(T, S, n_obs, g) if od == 4 else torch.randn(T, S, n_obs, od, generator=g, device="cuda", dtype=torch.float64).cumsum(0) * 0.05
        m = DynamicMarkovBlanketDiscovery(obs_shape=(n_obs, od), role_dims=roles, hidden_dims=hid, number_of_objects=nobj,
                                          device="cuda", dtype=torch.float64)
        m.update(y, None, None, iters=1, lr=0.5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        its = 3
        m.update(y, None, None, iters=its, lr=0.5)
        torch.cuda.synchronize()
        dt_ = (time.perf_counter() - t0) / its
        print(f"dmbd {name}: T={T} S={S} n_obs={n_obs}: {dt_ * 1e3:.1f} ms / VB iteration, ELBO {float(m.ELBO_last):.4e}", flush=True)


def bench_lds0(args):
    """The reference's own LDS example size (T=399, 64 series, hidden 6, obs 6, fp64; reference CPU
    0.45 s per E-step, 1.7 s per 3 VB iterations): launch-bound on a GPU, so also timed as a HIP graph replay."""
    from pyvbmp_amd.models import LinearDynamicalSystems
    T, S, h = 399, 64, 6
    g = torch.Generator(device="cuda").manual_seed(0)
    y = torch.randn(T, S, 6, generator=g, device="cuda", dtype=torch.float64).cumsum(0) * 0.05
    for graphed in (False, True):
        torch.manual_seed(0)
        m = LinearDynamicalSystems((6,), h, device="cuda", dtype=torch.float64)
        m.update(y, iters=4, graphed=graphed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.update(y, iters=20, graphed=graphed)
        torch.cuda.synchronize()
        dt_ms = (time.perf_counter() - t0) / 20 * 1e3
        print(f"lds example size (T=399, S=64, h=6, fp64) {'HIP graph' if graphed else 'eager    '}: {dt_ms:.3f} ms per VB "
              f"iteration, ELBO {float(m.ELBO().sum()):.4f}", flush=True)


def bench_mixlt(args):
    """Mixture of linear transforms at scale: N=1e6 samples, 8 experts, n=p=8 (+bias): the E-step is
    the K3a quadratic form of z=[x;y] on the matrix cores, the M-step K4 moments with 8 weight columns."""
    from pyvbmp_amd.transforms import MixtureofLinearTransforms
    N, n, p, K = 1_000_000, 8, 8, 8
    for dt in (torch.float64, torch.float32):
        g = torch.Generator(device="cuda").manual_seed(0)
        X = torch.randn(N, p, 1, generator=g, device="cuda", dtype=dt)
        Ws = torch.randn(K, n, p, generator=g, device="cuda", dtype=dt)
        z = torch.randint(K, (N,), generator=g, device="cuda")
        Y = Ws[z] @ X + 0.1 * torch.randn(N, n, 1, generator=g, device="cuda", dtype=dt)
        m = MixtureofLinearTransforms(n, p, K, device="cuda", dtype=dt)
        m.raw_update(X, Y, iters=2)
        t_e = _time_call(lambda: m.update_assignments(X, Y), reps=10)
        t_i = _time_call(lambda: m.raw_update(X, Y, iters=1), reps=10)
        print(f"mixture of linear transforms {str(dt)[6:]} N={N} n=p=8 experts=8: E-step {t_e:.3f} ms, full VB iteration "
              f"{t_i:.3f} ms -> {N / t_i * 1e3:.3e} samples/s", flush=True)


def bench_dmix(args):
    """The reference's two_moons model class at scale: dMixtureofLinearTransforms (input-dependent Polya-Gamma gate),
    N=1e6 samples, 8 experts, n=p=8."""
    from pyvbmp_amd.transforms import dMixtureofLinearTransforms
    N, n, p, K = 1_000_000, 8, 8, 8
    for dt in (torch.float64, torch.float32):
        g = torch.Generator(device="cuda").manual_seed(0)
        X = torch.randn(N, p, generator=g, device="cuda", dtype=dt)
        Ws = torch.randn(K, n, p, generator=g, device="cuda", dtype=dt)
        z = ((X[:, :3] > 0).long() * torch.tensor([1, 2, 4], device="cuda")).sum(-1)
        Y = (Ws[z] @ X.unsqueeze(-1)).squeeze(-1) + 0.1 * torch.randn(N, n, generator=g, device="cuda", dtype=dt)
        m = dMixtureofLinearTransforms(n, p, K, device="cuda", dtype=dt)
        m.raw_update(X, Y, iters=2)
        t_i = _time_call(lambda: m.raw_update(X, Y, iters=1), reps=5)
        t_g = _time_call(lambda: m.pi.raw_update(X, m.pi.predict(X), iters=2), reps=5)
        print(f"gated mixture of linear transforms {str(dt)[6:]} N={N} n=p=8 experts=8: full VB iteration {t_i:.3f} ms -> "
              f"{N / t_i * 1e3:.3e} samples/s (gate predict + 2-sweep update alone {t_g:.3f} ms)", flush=True)


def bench_k1(args):
    """K1 (batched SPD inverse + logdet) and K2a (Wishart ss_update) at B=1e6: the two other HBM-bound entry points"""
    from pyvbmp_amd.dists import Wishart
    B = args.B
    for dt in (torch.float64, torch.float32):
        for D in (16, 8, 32):
            Bd = B if D <= 16 else B // 4
            SExx, _, N = make_inputs(Bd, D, dt, "cuda")
            A = SExx + torch.eye(D, device="cuda", dtype=dt)
            b = A.element_size()
            ev = []
            _lib.launch_hooks = (lambda n: ev.append((n, _rec())), lambda n: ev.append((n, _rec())))
            for _ in range(12):
                ops.spd_inv_logdet(A)
            _lib.launch_hooks = None
            torch.cuda.synchronize()
            t = sorted(ev[i][1].elapsed_time(ev[i + 1][1]) for i in range(4, len(ev), 2))
            t1 = t[len(t) // 2]
            by = (2 * D * D + 1) * b * Bd
            w = Wishart((D, D), (Bd,), device="cuda", dtype=dt)
            ev = []
            _lib.launch_hooks = (lambda n: ev.append((n, _rec())), lambda n: ev.append((n, _rec())))
            for _ in range(12):
                w.ss_update(SExx, N, lr=1.0)
            _lib.launch_hooks = None
            torch.cuda.synchronize()
            t = sorted(ev[i][1].elapsed_time(ev[i + 1][1]) for i in range(4, len(ev), 2))
            t2 = t[len(t) // 2]
            by2 = (3 * D * D + 3) * b * Bd
            print(f"K1 spd_inv_logdet {str(dt)[6:]} D={D:2d} B={Bd}: {t1:.4f} ms -> {by / t1 / 1e6:.0f} GB/s ({by / t1 / 1e6 / 80:.1f}%)   "
                  f"K2a wishart_ss_update: {t2:.4f} ms -> {by2 / t2 / 1e6:.0f} GB/s ({by2 / t2 / 1e6 / 80:.1f}%)", flush=True)
            del SExx, A, w


def bench_gmm0(args):
    """BASELINE configs[0]: GaussianMixtureModel(4, 2) on 400 two-cluster points, 20 VB iterations: launch-bound, so the
    iteration is also timed as a HIP graph replay (pyvbmp_amd.graph)."""
    from pyvbmp_amd.models import GaussianMixtureModel
    g = torch.Generator(device="cuda").manual_seed(0)
    X = torch.cat((torch.randn(200, 2, generator=g, device="cuda", dtype=torch.float64) + 2.5,
                   torch.randn(200, 2, generator=g, device="cuda", dtype=torch.float64) - 2.5))
    for graphed in (False, True):
        m = GaussianMixtureModel(4, 2, device="cuda", dtype=torch.float64)
        m.update(X, iters=5, graphed=graphed)  # builds the graph when asked for
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            m.update(X, iters=20, graphed=graphed)
        torch.cuda.synchronize()
        dt_ms = (time.perf_counter() - t0) / reps * 1e3
        print(f"gmm config0 (K=4, D=2, N=400, fp64) {'HIP graph' if graphed else 'eager    '}: {dt_ms:.3f} ms per 20 "
              f"iterations ({dt_ms / 20 * 1e3:.0f} us / iteration), ELBO {float(m.ELBO()):.4f}", flush=True)


def bench_gmm(args):
    """GMM at scale: fused E-step (K3), weighted moments (K4) and one full VB iteration, N samples, K=4, D=16."""
    from pyvbmp_amd.models import GaussianMixtureModel
    for dt in (torch.float64, torch.float32):
        for K, D, N in ((4, 16, 4_000_000), (16, 16, 1_000_000), (4, 2, 8_000_000)):
            g = torch.Generator(device="cuda").manual_seed(0)
            X = torch.randn(N, D, generator=g, device="cuda", dtype=dt) + 3.0 * torch.randint(0, K, (N, 1), generator=g, device="cuda")
            m = GaussianMixtureModel(K, D, device="cuda", dtype=dt)
            m.update(X, iters=2)
            es = X.element_size()
            t_e = _time_call(lambda: m.update_assignments(X))
            t_m = _time_call(lambda: m.dist.raw_moments(m._view(X), m.p))
            t_it = _time_call(lambda: m.update(X, iters=1))
            be, bm = (D + K) * es, (D + K) * es
            print(f"gmm {str(dt)[6:]} K={K} D={D} N={N}: E-step {t_e:.3f} ms ({be * N / t_e / 1e6:.0f} GB/s, {be * N / t_e / 1e6 / 80:.1f}%)  "
                  f"moments {t_m:.3f} ms ({bm * N / t_m / 1e6:.0f} GB/s, {bm * N / t_m / 1e6 / 80:.1f}%)  full iteration {t_it:.3f} ms "
                  f"-> {N / t_it * 1e3:.3e} samples/s", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["niw"])
    ap.add_argument("--B", type=int, default=1_000_000)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--N", type=int, default=262144)
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--S", type=int, default=4096)
    args = ap.parse_args()
    for w in args.what:
        {"niw": bench_niw, "copy": bench_copy, "mnw": bench_mnw, "lds": bench_lds, "dmbd": bench_dmbd, "gmm": bench_gmm, "gmm0": bench_gmm0, "lds0": bench_lds0, "mixlt": bench_mixlt, "dmix": bench_dmix, "k1": bench_k1}[w](args)
