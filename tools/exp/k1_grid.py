"""K1 (batched SPD inverse + log det) with a capped, persistent grid: blocks per CU 0 (= one tile per wave), 4, 8, 16, 32, interleaved"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd import _lib, ops
lib = _lib.load()
lib.vbmp_debug_set_blocks_per_cu.argtypes = [ctypes.c_int]
caps = [int(v) for v in (sys.argv[1:] or ["0", "4", "8", "16", "32"])]


def _r():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


for dt in (torch.float64, torch.float32):
    for D, B in ((32, 250000), (16, 1000000), (8, 1000000)):
        g = torch.Generator(device="cuda").manual_seed(0)
        X = torch.randn(B, D, D + 2, generator=g, device="cuda", dtype=dt)
        A = X @ X.transpose(-2, -1) / (D + 2) + 0.5 * torch.eye(D, device="cuda", dtype=dt)
        del X
        times = {c: [] for c in caps}
        for rnd in range(6):
            for c in caps:
                lib.vbmp_debug_set_blocks_per_cu(c)
                ev = []
                _lib.launch_hooks = (lambda n: ev.append(_r()), lambda n: ev.append(_r()))
                for _ in range(5):
                    ops.spd_inv_logdet(A)
                _lib.launch_hooks = None
                torch.cuda.synchronize()
                if rnd >= 1:
                    times[c] += [ev[i].elapsed_time(ev[i + 1]) for i in range(0, len(ev), 2)]
        lib.vbmp_debug_set_blocks_per_cu(0)
        by = 2 * B * D * D * A.element_size()
        print(f"{str(dt)[6:]} D={D} B={B}: " + "  ".join(f"cap {c}: {sorted(t)[len(t)//2]:.4f} ms ({by / sorted(t)[len(t)//2] / 1e6 / 80:.1f}%)" for c, t in times.items()), flush=True)
