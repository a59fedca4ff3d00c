"""K9 block form: forward filter and Gamma recursion in two blocks per series + the T-parallel rest (MODE 1, default for few series)
against the one-block sweep (VBMP_DBG_BLK_MONO = 0x4000000): outputs bitwise, kernel times."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd import _lib
from pyvbmp_amd.models import LinearDynamicalSystems
lib = _lib.load()
lib.vbmp_debug_set_flags.argtypes = [ctypes.c_int]
MONO = 0x4000000
for dt in (torch.float64, torch.float32):
    for (h, T, S) in ((52, 100, 20), (52, 20, 3), (12, 200, 64), (21, 50, 7), (60, 30, 5), (52, 1, 4), (52, 2, 4)):
        g = torch.Generator(device="cuda").manual_seed(0)
        y = torch.randn(T, S, 6, generator=g, device="cuda", dtype=dt).cumsum(0) * 0.05
        torch.manual_seed(1)
        m = LinearDynamicalSystems((6,), h, latent_noise='shared', device="cuda", dtype=dt)
        yy, uu, rr = m.reshape_inputs(y)
        m.update_latents(yy, uu, rr)  # creates the posterior container
        res = {}
        for name, flag in (("split", 0), ("mono", MONO)):
            lib.vbmp_debug_set_flags(flag)
            for sums_only in (False, True):
                outs = m.forward_backward_loop(yy, uu, rr, sums_only=sums_only)
                ev = []

                def rec(n):
                    e = torch.cuda.Event(enable_timing=True)
                    e.record()
                    ev.append((n, e))
                _lib.launch_hooks = (rec, rec)
                for _ in range(3):
                    outs = m.forward_backward_loop(yy, uu, rr, sums_only=sums_only)
                _lib.launch_hooks = None
                torch.cuda.synchronize()
                tk = sorted(ev[i][1].elapsed_time(ev[i + 1][1]) for i in range(0, len(ev), 2) if ev[i][0] == "vbmp_lds_smoother")
                keep = [m.px.Sigma.clone(), m.px.mu.clone(), m.px.invSigma.clone(), m.px.invSigmamu.clone(), outs[1].clone(), outs[2].clone(),
                        outs[3].clone(), m._time_sums[0].clone(), m._time_sums[1].clone()]
                if not sums_only:
                    keep.append(outs[0].clone())
                else:
                    keep.append(outs[0][-1].clone())
                res[(name, sums_only)] = (tk[len(tk) // 2], keep)
        lib.vbmp_debug_set_flags(0)
        for so in (False, True):
            a, b = res[("split", so)], res[("mono", so)]
            same = all(torch.equal(x, y) for x, y in zip(a[1], b[1]))
            worst = max(float((x - y).abs().max() / y.abs().max().clamp_min(1e-300)) for x, y in zip(a[1], b[1]))
            print(f"{str(dt)[6:]} h={h} T={T} S={S} sums_only={so}: split {a[0]:.3f} ms, one block {b[0]:.3f} ms; outputs "
                  f"{'bitwise equal' if same else f'DIFFER by {worst:.1e}'}", flush=True)
