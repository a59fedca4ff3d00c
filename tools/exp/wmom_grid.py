"""K4 on the matrix cores: kernel time against the grid cap (same-address atomics of the final combine vs occupancy)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd import ops, _lib
lib = _lib.load()
lib.vbmp_debug_set_blocks_per_cu.argtypes = [ctypes.c_int]


def tm(f, reps=20):
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
    return sum(ev[i].elapsed_time(ev[i + 1]) for i in range(0, len(ev), 2)) / reps


for dt in (torch.float64, torch.float32):
    for (N, K, D) in ((4_000_000, 4, 16), (8_000_000, 4, 2), (1_000_000, 1, 32)):
        g = torch.Generator(device="cuda").manual_seed(0)
        X = torch.randn(N, 1, D, generator=g, device="cuda", dtype=dt)
        p = torch.rand(N, K, generator=g, device="cuda", dtype=dt)
        line = []
        for cap in (1, 2, 4, 8, 16):
            lib.vbmp_debug_set_blocks_per_cu(cap)
            line.append(f"{cap}/CU {tm(lambda: ops.weighted_moments(X, p, 1, (K,))) * 1e3:.0f} us")
        lib.vbmp_debug_set_blocks_per_cu(0)
        print(f"{str(dt)[6:]} N={N} K={K} D={D}: " + ", ".join(line), flush=True)
