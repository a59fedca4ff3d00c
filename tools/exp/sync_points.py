"""list the device->host synchronisation points of one VB iteration (they stop the CPU from running ahead of the
GPU and forbid HIP-graph capture): torch.cuda.set_sync_debug_mode('warn')"""
import os, sys, warnings, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kbench import boids
from pyvbmp_amd.models import DynamicMarkovBlanketDiscovery, LinearDynamicalSystems

which = sys.argv[1] if len(sys.argv) > 1 else "dmbd"
g = torch.Generator(device="cuda").manual_seed(0)
if which == "dmbd":
    y = boids(40, 8, 12, g)
    m = DynamicMarkovBlanketDiscovery(obs_shape=(12, 4), role_dims=(1, 2, 2), hidden_dims=(4, 4, 4), number_of_objects=6,
                                      device="cuda", dtype=torch.float64)
    step = lambda: m.update(y, None, None, iters=1, lr=0.5)
elif which == "lds":
    y = torch.randn(50, 16, 6, generator=g, device="cuda", dtype=torch.float64).cumsum(0) * 0.05
    m = LinearDynamicalSystems((6,), 6, device="cuda", dtype=torch.float64)
    step = lambda: m.update(y, iters=1)
elif which == "lds_big":  # hidden > 8: composed smoother
    y = torch.randn(30, 8, 12, generator=g, device="cuda", dtype=torch.float64).cumsum(0) * 0.05
    m = LinearDynamicalSystems((12,), 12, device="cuda", dtype=torch.float64)
    step = lambda: m.update(y, iters=1)
elif which == "gmm":
    from pyvbmp_amd.models import GaussianMixtureModel
    y = torch.randn(5000, 16, generator=g, device="cuda", dtype=torch.float64)
    m = GaussianMixtureModel(4, 16, device="cuda", dtype=torch.float64)
    step = lambda: m.update(y, iters=1)
elif which == "mixlds":
    from pyvbmp_amd.models import MixtureofLinearDynamicalSystems
    y = torch.randn(40, 12, 5, generator=g, device="cuda", dtype=torch.float64).cumsum(0) * 0.05
    m = MixtureofLinearDynamicalSystems(3, (5,), 3, 0, 0, device="cuda", dtype=torch.float64)
    step = lambda: m.update(y, None, None, iters=1, verbose=False)
elif which == "mixlt":
    from pyvbmp_amd.transforms import MixtureofLinearTransforms
    X = torch.randn(5000, 4, 1, generator=g, device="cuda", dtype=torch.float64)
    Y = torch.randn(5000, 3, 1, generator=g, device="cuda", dtype=torch.float64)
    m = MixtureofLinearTransforms(3, 4, 3, device="cuda", dtype=torch.float64)
    step = lambda: m.raw_update(X, Y, iters=1)
elif which == "arhmm":
    from pyvbmp_amd.models import ARHMM
    y = torch.randn(60, 10, 4, generator=g, device="cuda", dtype=torch.float64).cumsum(0) * 0.05
    m = ARHMM(5, 4, 4, device="cuda", dtype=torch.float64)
    XY = (y[:-1].unsqueeze(-1), y[1:].unsqueeze(-1))
    step = lambda: m.update(XY, iters=1)
step()
torch.cuda.synchronize()
seen = collections.Counter()
def hook(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" in str(message).lower():
        st = [f for f in traceback.extract_stack() if "/pyvbmp_amd/" in f.filename]
        key = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(st[-3:]))
        seen[key] += 1
warnings.showwarning = hook
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
step()
torch.cuda.set_sync_debug_mode("default")
for k, v in seen.most_common():
    print(v, k)
print("total sync points in one iteration:", sum(seen.values()))
