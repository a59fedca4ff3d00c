"""update_latents at BASELINE config 4 with / without the sums-only kernel outputs, interleaved in ONE process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd import ops, _lib
from pyvbmp_amd.models import LinearDynamicalSystems
from tools.synth import lorenz

T, S = 1000, 4096
orig = ops.lds_smoother


def dense(*a, **k):
    k["sums_only"] = False
    return orig(*a, **k)


def _r():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


for dt in (torch.float64, torch.float32):
    y = lorenz(T, S, torch.Generator(device="cuda").manual_seed(0), device="cuda", dtype=dt)
    torch.manual_seed(0)
    m = LinearDynamicalSystems((6,), 6, latent_noise='shared', device="cuda", dtype=dt)
    inp = m.reshape_inputs(y)
    for rnd in range(3):
        for name, fn in (("sums_only", orig), ("dense", dense)):
            ops.lds_smoother = fn
            for _ in range(3):
                m.update_latents(*inp)
            torch.cuda.synchronize()
            ev = []
            _lib.launch_hooks = (lambda n: ev.append((n, _r())), lambda n: ev.append((n, _r())))
            e0 = _r()
            for _ in range(10):
                m.update_latents(*inp)
            e1 = _r()
            _lib.launch_hooks = None
            torch.cuda.synchronize()
            ks = {}
            for i in range(0, len(ev), 2):
                ks.setdefault(ev[i][0], []).append(ev[i][1].elapsed_time(ev[i + 1][1]))
            print(f"{str(dt)[6:]} {name:10s} update_latents {e0.elapsed_time(e1) / 10:.3f} ms; " +
                  ", ".join(f"{k[5:]} {sorted(v)[len(v) // 2]:.3f}" for k, v in ks.items()), flush=True)
    ops.lds_smoother = orig
