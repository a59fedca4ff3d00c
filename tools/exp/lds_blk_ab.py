"""K9 block form at the flocking DMBD's hidden dimension (52): compile-time H instance vs the generic one (flag 0x2000000), interleaved"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd import _lib
from pyvbmp_amd.models import LinearDynamicalSystems
lib = _lib.load()
lib.vbmp_debug_set_flags.argtypes = [ctypes.c_int]
flags = [int(v, 0) for v in (sys.argv[1:] or ["0", "0x2000000"])]
for dt in (torch.float64, torch.float32):
    h, T, S = 52, 100, 20
    g = torch.Generator(device="cuda").manual_seed(0)
    y = torch.randn(T, S, 6, generator=g, device="cuda", dtype=dt).cumsum(0) * 0.05
    m = LinearDynamicalSystems((6,), h, latent_noise='shared', device="cuda", dtype=dt)
    yy, uu, rr = m.reshape_inputs(y)
    m.update_latents(yy, uu, rr)
    ref = None
    for rnd in range(2):
        for f in flags:
            lib.vbmp_debug_set_flags(f)
            ev = []

            def rec(n):
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ev.append((n, e))
            m.forward_backward_loop(yy, uu, rr)
            _lib.launch_hooks = (rec, rec)
            for _ in range(3):
                m.forward_backward_loop(yy, uu, rr)
            _lib.launch_hooks = None
            torch.cuda.synchronize()
            tk = sorted(ev[i][1].elapsed_time(ev[i + 1][1]) for i in range(0, len(ev), 2) if ev[i][0] == "vbmp_lds_smoother")
            out = m.px.Sigma.clone()
            ref = out if ref is None else ref
            print(f"{str(dt)[6:]} flags={f:#x}: block kernel {tk[len(tk) // 2]:.2f} ms; max |Sigma - first| {float((out - ref).abs().max()):.1e}", flush=True)
    lib.vbmp_debug_set_flags(0)
