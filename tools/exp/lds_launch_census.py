"""how many device launches does each part of update_latents issue? (torch.profiler kernel counts of the parts run on their own)"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from pyvbmp_amd.models import LinearDynamicalSystems
from tools.synth import lorenz
T, S = 1000, 4096
y = lorenz(T, S, torch.Generator(device="cuda").manual_seed(0), device="cuda", dtype=torch.float64)
m = LinearDynamicalSystems((6,), 6, latent_noise='shared', device="cuda", dtype=torch.float64)
yy, uu, rr = m.reshape_inputs(y)
for _ in range(3):
    m.update_latents(yy, uu, rr)
torch.cuda.synchronize()


def census(name, fn):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    ks = [ev for ev in prof.events() if ev.device_type == torch.autograd.DeviceType.CUDA]
    c = collections.Counter(ev.name[:60] for ev in ks)
    print(f"== {name}: {len(ks)} device activities")
    for k, v in c.most_common(12):
        print(f"   {v:3d} {k}")
    tm = collections.Counter()
    for ev in ks:
        tm[ev.name[:80]] += (ev.device_time if hasattr(ev, "device_time") else ev.cuda_time) / 1e3
    print(f"   device time {sum(tm.values()):.3f} ms:")
    for k, v in tm.most_common(10):
        print(f"   {v:7.3f} ms  {k}")


om, x0, A = m.obs_model, m.x0, m.A
census("obs_model.EinvSigma", lambda: om.EinvSigma())
census("obs_model.EXTinvUX", lambda: om.EXTinvUX())
census("obs_model.EXTinvU", lambda: om.EXTinvU())
census("obs_model.ElogdetinvSigma", lambda: om.ElogdetinvSigma())
census("log_likelihood_function", lambda: m.log_likelihood_function(yy, rr))
census("x0.EXTinvUX + ElogdetinvSigma + EinvSigma + EinvSigmamu", lambda: (x0.EXTinvUX(), x0.ElogdetinvSigma(), x0.EinvSigma(), x0.EinvSigmamu()))
census("A.ElogdetinvSigma", lambda: A.ElogdetinvSigma())
census("forward_backward_loop", lambda: m.forward_backward_loop(yy, uu, rr, sums_only=True))
census("update_latents", lambda: m.update_latents(yy, uu, rr))
if "--src" in sys.argv:
    from tools.exp._census import by_source
    by_source("update_latents", lambda: m.update_latents(yy, uu, rr))
