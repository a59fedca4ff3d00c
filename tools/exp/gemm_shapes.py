"""Which library GEMMs does a DMBD (flocking) / LDS iteration call, with which shapes and for how long?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from tools.kbench import boids
from pyvbmp_amd.models import DynamicMarkovBlanketDiscovery, LinearDynamicalSystems

what = sys.argv[1] if len(sys.argv) > 1 else "dmbd"
g = torch.Generator(device="cuda").manual_seed(0)
if what == "dmbd":
    T, S, n_obs = 100, 20, 12
    y = boids(T, S, n_obs, g)
    m = DynamicMarkovBlanketDiscovery(obs_shape=(n_obs, 4), role_dims=(1, 2, 2), hidden_dims=(4, 4, 4), number_of_objects=6,
                                      device="cuda", dtype=torch.float64)
    run = lambda: m.update(y, None, None, iters=1, lr=0.5)
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
