"""time of the smoother for hidden dims beyond the register forms: block-per-series kernel vs composed recursion"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd import _lib
from pyvbmp_amd.models import LinearDynamicalSystems


def tm(f, reps=3):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for dt in (torch.float64, torch.float32):
    for (h, T, S) in ((52, 100, 20), (12, 200, 256), (32, 100, 512)):
        g = torch.Generator(device="cuda").manual_seed(0)
        y = torch.randn(T, S, 6, generator=g, device="cuda", dtype=dt).cumsum(0) * 0.05
        m = LinearDynamicalSystems((6,), h, latent_noise='shared', device="cuda", dtype=dt)
        yy, uu, rr = m.reshape_inputs(y)
        m.update_latents(yy, uu, rr)  # creates px
        ev = []
        def rec(n):
            e = torch.cuda.Event(enable_timing=True); e.record(); ev.append((n, e))
        _lib.launch_hooks = (rec, rec)
        m.forward_backward_loop(yy, uu, rr)
        _lib.launch_hooks = None
        torch.cuda.synchronize()
        tk = [ev[i][1].elapsed_time(ev[i + 1][1]) for i in range(0, len(ev), 2) if ev[i][0] == "vbmp_lds_smoother"]
        t_blk = tm(lambda: m.forward_backward_loop(yy, uu, rr))
        old = _lib.LDS_MAX_H_BLOCK
        _lib.LDS_MAX_H_BLOCK = 0
        t_cmp = tm(lambda: m.forward_backward_loop(yy, uu, rr))
        _lib.LDS_MAX_H_BLOCK = old
        print(f"{str(dt)[6:]} h={h} T={T} S={S}: block kernel {tk[0]:.2f} ms (whole forward_backward_loop {t_blk:.2f} ms), "
              f"composed {t_cmp:.2f} ms", flush=True)
