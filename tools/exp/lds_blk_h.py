import ctypes, os, sys
sys.path.insert(0, "/root/repo")
import torch
from pyvbmp_amd import _lib
from pyvbmp_amd.models import LinearDynamicalSystems
lib = _lib.load(); lib.vbmp_debug_set_flags.argtypes = [ctypes.c_int]
for dt in (torch.float64, torch.float32):
  for h, T, S in ((12, 200, 256), (14, 200, 256), (24, 100, 128), (40, 100, 20), (60, 100, 20)):
    g = torch.Generator(device="cuda").manual_seed(0)
    y = torch.randn(T, S, 6, generator=g, device="cuda", dtype=dt).cumsum(0) * 0.05
    m = LinearDynamicalSystems((6,), h, latent_noise='shared', device="cuda", dtype=dt)
    yy, uu, rr = m.reshape_inputs(y); m.update_latents(yy, uu, rr)
    for f in (0, 0x2000000, 0, 0x2000000):
        lib.vbmp_debug_set_flags(f)
        ev = []
        def rec(n):
            e = torch.cuda.Event(enable_timing=True); e.record(); ev.append((n, e))
        m.forward_backward_loop(yy, uu, rr)
        _lib.launch_hooks = (rec, rec)
        for _ in range(3): m.forward_backward_loop(yy, uu, rr)
        _lib.launch_hooks = None; torch.cuda.synchronize()
        tk = sorted(ev[i][1].elapsed_time(ev[i+1][1]) for i in range(0, len(ev), 2) if ev[i][0] == "vbmp_lds_smoother")
        print(f"{str(dt)[6:]} h={h} T={T} S={S} flags={f:#x}: {tk[len(tk)//2]:.2f} ms", flush=True)
    lib.vbmp_debug_set_flags(0)
