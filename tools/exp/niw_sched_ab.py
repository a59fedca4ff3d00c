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
LOOP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab", "libvbmp_k2loop.so")  # -DVBMP_K2_LOOP=1: tile loop, grid caps
DEF = _lib.LIB_PATH
VAR = [("straight-line, xcd", DEF, 0, 0), ("straight-line, plain order", DEF, 4, 0), ("loop, xcd, one tile/wave", LOOP, 0, 0),
       ("loop, plain, 32 blocks/CU (r02)", LOOP, 4, 32), ("loop, xcd, 128 blocks/CU", LOOP, 0, 128)]
LIBS = {}


def use(path):
    if path not in LIBS:
        _lib._lib = None
        _lib.LIB_PATH = path
        l = _lib.load()
        l.vbmp_debug_set_flags.argtypes = [ctypes.c_int]
        l.vbmp_debug_set_blocks_per_cu.argtypes = [ctypes.c_int]
        LIBS[path] = l
    _lib._lib = LIBS[path]
    _lib.LIB_PATH = path
    return LIBS[path]
for dt in (torch.float64, torch.float32):
    for D, B in ((6, 1_000_000), (8, 1_000_000), (12, 1_000_000), (16, 1_000_000), (20, 500_000), (32, 250_000), (64, 62_500), (2, 4_000_000)):
        SExx, SEx, N = make_inputs(B, D, dt, "cuda")
        q = NormalInverseWishart((D,), (B,), device="cuda", dtype=dt)
        times = {v[0]: [] for v in VAR}
        for rnd in range(6):
            for name, path, fl, cap in VAR:
                lib = use(path)
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
                lib.vbmp_debug_set_flags(0)
                lib.vbmp_debug_set_blocks_per_cu(0)
                if rnd >= 1:
                    times[name] += [ev[i].elapsed_time(ev[i + 1]) for i in range(0, len(ev), 2)]
        use(DEF)
        it = 8 if dt == torch.float64 else 4
        by = (3 * D * D + 2 * D + 4) * it * B
        print(f"{str(dt)[6:]} D={D:2d} B={B}: " + "   ".join(f"{n}: {sorted(t)[len(t) // 2]:.4f} ms ({by / sorted(t)[len(t) // 2] / 8e7:.1f} %)" for n, t in times.items()), flush=True)
        del SExx, SEx, N, q
        torch.cuda.empty_cache()
