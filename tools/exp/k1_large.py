"""K1 at D > 32: block-per-matrix form vs one-wave form, latency (small batch) and throughput (large batch)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd import ops, _lib
lib = _lib.load()
lib.vbmp_debug_set_flags.argtypes = [ctypes.c_int]


def tm(f, reps=20):
    """kernel time from HIP events recorded by the launch hooks right around the enqueue"""
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ev = []

    def rec(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev.append(e)
    _lib.launch_hooks = (rec, rec)
    for _ in range(reps):
        f()
    _lib.launch_hooks = None
    torch.cuda.synchronize()
    return sum(ev[i].elapsed_time(ev[i + 1]) for i in range(0, len(ev), 2)) / (len(ev) // 2)


for dt in (torch.float64, torch.float32):
    for D in (52, 64):
        for B in (20, 2000, 62500):
            g = torch.Generator(device="cuda").manual_seed(0)
            X = torch.randn(B, D, D + 8, generator=g, device="cuda", dtype=dt)
            A = X @ X.transpose(-2, -1) / (D + 8) + 0.5 * torch.eye(D, device="cuda", dtype=dt)
            out = []
            for flag in (0x80, 0x40):
                lib.vbmp_debug_set_flags(flag)
                out.append(tm(lambda: ops.spd_inv_logdet(A)))
            lib.vbmp_debug_set_flags(0)
            print(f"{str(dt)[6:]} D={D} B={B}: block form {out[0]*1e3:.1f} us, one-wave form {out[1]*1e3:.1f} us", flush=True)
