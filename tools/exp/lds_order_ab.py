"""K9 (row-per-lane form) at BASELINE config 4: XCD-contiguous block -> series mapping against plain blockIdx order (debug flag 0x4),
sums-only (update_latents) and dense outputs, both precisions, full recursion too; outputs must be bitwise the same."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd import _lib
from pyvbmp_amd.models import LinearDynamicalSystems
from tools.synth import lorenz
lib = _lib.load()
lib.vbmp_debug_set_flags.argtypes = [ctypes.c_int]
T, S = 1000, 4096
for dt in (torch.float64, torch.float32):
    y = lorenz(T, S, torch.Generator(device="cuda").manual_seed(0), device="cuda", dtype=dt)
    torch.manual_seed(0)
    m = LinearDynamicalSystems((6,), 6, latent_noise='shared', device="cuda", dtype=dt)
    inp = m.reshape_inputs(y)
    m.update_latents(*inp)
    for mode in ("auto", "off"):
        m.fixed_point = mode
        for sums_only in (True, False):
            res = {}
            for rnd in range(3):
                for name, fl in (("xcd", 0), ("plain", 4)):
                    lib.vbmp_debug_set_flags(fl)
                    ev = []

                    def rec(n):
                        e = torch.cuda.Event(enable_timing=True)
                        e.record()
                        ev.append((n, e))
                    m.forward_backward_loop(*inp, sums_only=sums_only)
                    _lib.launch_hooks = (rec, rec)
                    for _ in range(4):
                        m.forward_backward_loop(*inp, sums_only=sums_only)
                    _lib.launch_hooks = None
                    torch.cuda.synchronize()
                    ts = [ev[i][1].elapsed_time(ev[i + 1][1]) for i in range(0, len(ev), 2) if ev[i][0] == "vbmp_lds_smoother"]
                    res.setdefault(name, []).extend(ts)
                    if rnd == 0:
                        res[name + "_out"] = (m.px.Sigma.clone(), m.px.mu.clone())
            lib.vbmp_debug_set_flags(0)
            same = all(torch.equal(a, b) for a, b in zip(res["xcd_out"], res["plain_out"]))
            med = {k: sorted(v)[len(v) // 2] for k, v in res.items() if not k.endswith("_out")}
            print(f"{str(dt)[6:]} fixed_point={mode} sums_only={sums_only}: xcd {med['xcd']:.3f} ms, plain {med['plain']:.3f} ms; outputs "
                  f"{'bitwise equal' if same else 'DIFFER'}", flush=True)
            del res
