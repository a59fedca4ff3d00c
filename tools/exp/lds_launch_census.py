"""which Python lines of update_latents launch the small kernels? (torch.profiler, stacks grouped by the innermost pyvbmp_amd frame)"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from pyvbmp_amd.models import LinearDynamicalSystems
from tools.synth import lorenz
T, S = 1000, 4096
y = lorenz(T, S, torch.Generator(device="cuda").manual_seed(0), device="cuda", dtype=torch.float64)
m = LinearDynamicalSystems((6,), 6, latent_noise='shared', device="cuda", dtype=torch.float64)
inp = m.reshape_inputs(y)
for _ in range(3):
    m.update_latents(*inp)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    m.update_latents(*inp)
    torch.cuda.synchronize()
c = collections.Counter()
names = collections.defaultdict(collections.Counter)
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.cpu_parent is None and len(ev.kernels) > 0:
        fr = [f for f in (ev.stack or []) if "pyvbmp_amd" in f]
        key = (fr[0] if fr else "?")[-100:]
        c[key] += len(ev.kernels)
        names[key][ev.name] += len(ev.kernels)
for k, v in c.most_common(70):
    print(f"{v:4d}  {k}   {dict(names[k])}")
print(sum(c.values()))
