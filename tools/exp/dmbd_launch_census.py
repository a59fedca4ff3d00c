"""device activities per part of one DMBD VB iteration at the Flocking_example hyper-parameters (torch.profiler counts + time)"""
import collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from pyvbmp_amd.models import DynamicMarkovBlanketDiscovery
from tools.synth import boids
T, S, n_obs = 100, 20, 12
g = torch.Generator(device="cuda").manual_seed(0)
y = boids(T, S, n_obs, g, device="cuda", dtype=torch.float64)
torch.manual_seed(0)
m = DynamicMarkovBlanketDiscovery(obs_shape=(n_obs, 4), role_dims=(1, 2, 2), hidden_dims=(4, 4, 4), regression_dim=-1, control_dim=0,
                                  number_of_objects=6, device="cuda", dtype=torch.float64)
for _ in range(2):
    m.update(y, None, None, iters=1, latent_iters=1, lr=1.0)
yy, uu, rr = m.reshape_inputs(y, None, None)


def census(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    ks = [ev for ev in prof.events() if ev.device_type == torch.autograd.DeviceType.CUDA]
    tot = sum(ev.device_time if hasattr(ev, "device_time") else ev.cuda_time for ev in ks) / 1e3
    c = collections.Counter(ev.name[:50] for ev in ks)
    print(f"== {name}: {len(ks)} device activities, {tot:.2f} ms of device time, {wall:.2f} ms wall (unprofiled)")
    for k, v in c.most_common(6):
        print(f"   {v:4d} {k}")
    tm = collections.Counter()
    for ev in ks:
        tm[ev.name[:90]] += (ev.device_time if hasattr(ev, "device_time") else ev.cuda_time) / 1e3
    for k, v in tm.most_common(8):
        print(f"   {v:7.3f} ms  {k}")


census("update_assignments", lambda: m.update_assignments(yy, rr))
census("update_obs_parms", lambda: m.update_obs_parms(yy, rr, lr=1.0))
census("update_latents", lambda: m.update_latents(yy, uu, rr))
census("ELBO", lambda: m.ELBO())
census("update_latent_parms", lambda: m.update_latent_parms(p=None, lr=1.0))



from tools.exp._census import by_source  # noqa: E402

if "--src" in sys.argv:
    by_source("update_obs_parms", lambda: m.update_obs_parms(yy, rr, lr=1.0))
    by_source("update_latent_parms", lambda: m.update_latent_parms(p=None, lr=1.0))
    by_source("update_latents", lambda: m.update_latents(yy, uu, rr))
    by_source("ELBO", lambda: m.ELBO())
    by_source("update_assignments", lambda: m.update_assignments(yy, rr))
