"""Which library GEMMs does a DMBD (flocking) / LDS iteration call, with which shapes and for how long?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from tools.kbench import boids
from pyvbmp_amd.models import DynamicMarkovBlanketDiscovery, LinearDynamicalSystems

what = sys.argv[1] if len(sys.argv) > 1 else "dmbd"
g = torch.Generator(device="cuda").manual_seed(0)
if what == "dmbd6":
    T, S, n_obs = 400, 64, 1
    y = torch.randn(T, S, n_obs, 6, generator=g, device="cuda", dtype=torch.float64).cumsum(0) * 0.05
    m = DynamicMarkovBlanketDiscovery(obs_shape=(n_obs, 6), role_dims=(1, 2, 1), hidden_dims=(2, 2, 2), number_of_objects=1,
                                      device="cuda", dtype=torch.float64)
    run = lambda: m.update(y, None, None, iters=1, lr=0.5)
elif what == "dmbd":
    T, S, n_obs = 100, 20, 12
    y = boids(T, S, n_obs, g)
    m = DynamicMarkovBlanketDiscovery(obs_shape=(n_obs, 4), role_dims=(1, 2, 2), hidden_dims=(4, 4, 4), number_of_objects=6,
                                      device="cuda", dtype=torch.float64)
    run = lambda: m.update(y, None, None, iters=1, lr=0.5)
elif what == "gmm":
    from pyvbmp_amd.models import GaussianMixtureModel
    X = torch.randn(4_000_000, 16, generator=g, device="cuda", dtype=torch.float64) + 3.0 * torch.randint(0, 4, (4_000_000, 1), generator=g, device="cuda")
    m = GaussianMixtureModel(4, 16, device="cuda", dtype=torch.float64)
    run = lambda: m.update(X, iters=1)
elif what == "mixlt":
    from pyvbmp_amd.transforms import MixtureofLinearTransforms
    N, n, p, K = 1_000_000, 8, 8, 8
    X = torch.randn(N, p, 1, generator=g, device="cuda", dtype=torch.float64)
    Ws = torch.randn(K, n, p, generator=g, device="cuda", dtype=torch.float64)
    z = torch.randint(K, (N,), generator=g, device="cuda")
    Y = Ws[z] @ X + 0.1 * torch.randn(N, n, 1, generator=g, device="cuda", dtype=torch.float64)
    m = MixtureofLinearTransforms(n, p, K, device="cuda", dtype=torch.float64)
    run = lambda: m.raw_update(X, Y, iters=1)
elif what == "dmix":
    from pyvbmp_amd.transforms import dMixtureofLinearTransforms
    N, n, p, K = 1_000_000, 8, 8, 8
    X = torch.randn(N, p, generator=g, device="cuda", dtype=torch.float64)
    Ws = torch.randn(K, n, p, generator=g, device="cuda", dtype=torch.float64)
    z = ((X[:, :3] > 0).long() * torch.tensor([1, 2, 4], device="cuda")).sum(-1)
    Y = (Ws[z] @ X.unsqueeze(-1)).squeeze(-1) + 0.1 * torch.randn(N, n, generator=g, device="cuda", dtype=torch.float64)
    m = dMixtureofLinearTransforms(n, p, K, device="cuda", dtype=torch.float64)
    run = lambda: m.raw_update(X, Y, iters=1)
else:
    y = torch.randn(1000, 4096, 6, generator=g, device="cuda", dtype=torch.float64).cumsum(0) * 0.05
    m = LinearDynamicalSystems((6,), 6, latent_noise='shared', device="cuda", dtype=torch.float64)
    yy, uu, rr = m.reshape_inputs(y)
    run = lambda: m.update_latents(yy, uu, rr)
run(); run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    run()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if True:
        t = getattr(e, "device_time_total", None)
        if t is None:
            t = e.cuda_time_total
        rows.append((t, e.count, e.key, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
for t, c, k, sh in rows[:int(os.environ.get('TOPN', '34'))]:
    print(f"{t / 1e3:8.3f} ms {c:4d}x {k:14s} {sh}")
