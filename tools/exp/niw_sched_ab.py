"""K2 tile schedule at the other shapes: XCD-contiguous / plain blockIdx order x one tile per wave / 32 blocks per CU, interleaved"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import make_inputs
from pyvbmp_amd import _lib
from pyvbmp_amd.dists import NormalInverseWishart
lib = _lib.load()
lib.vbmp_debug_set_flags.argtypes = [ctypes.c_int]
lib.vbmp_debug_set_blocks_per_cu.argtypes = [ctypes.c_int]
VAR = [("xcd, one tile/wave", 0, 0), ("plain, one tile/wave", 4, 0), ("plain, 32 blocks/CU", 4, 32), ("xcd, 32 blocks/CU", 0, 32), ("xcd, 128 blocks/CU", 0, 128)]
for dt in (torch.float64, torch.float32):
    for D, B in ((6, 1_000_000), (8, 1_000_000), (12, 1_000_000), (16, 1_000_000), (20, 500_000), (32, 250_000), (64, 62_500), (2, 4_000_000)):
        SExx, SEx, N = make_inputs(B, D, dt, "cuda")
        q = NormalInverseWishart((D,), (B,), device="cuda", dtype=dt)
        times = {v[0]: [] for v in VAR}
        for rnd in range(6):
            for name, fl, cap in VAR:
                lib.vbmp_debug_set_flags(fl)
                lib.vbmp_debug_set_blocks_per_cu(cap)
                ev = []

                def _r(n):
                    e = torch.cuda.Event(enable_timing=True)
                    e.record()
                    ev.append(e)
                _lib.launch_hooks = (_r, _r)
                for _ in range(4):
                    q.ss_update(SExx, SEx, N, lr=1.0, beta=None)
                _lib.launch_hooks = None
                torch.cuda.synchronize()
                if rnd >= 1:
                    times[name] += [ev[i].elapsed_time(ev[i + 1]) for i in range(0, len(ev), 2)]
        lib.vbmp_debug_set_flags(0)
        lib.vbmp_debug_set_blocks_per_cu(0)
        it = 8 if dt == torch.float64 else 4
        by = (3 * D * D + 2 * D + 4) * it * B
        print(f"{str(dt)[6:]} D={D:2d} B={B}: " + "   ".join(f"{n}: {sorted(t)[len(t) // 2]:.4f} ms ({by / sorted(t)[len(t) // 2] / 8e7:.1f} %)" for n, t in times.items()), flush=True)
        del SExx, SEx, N, q
        torch.cuda.empty_cache()
