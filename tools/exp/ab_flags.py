"""Interleaved A/B of K2 debug flags (vbmp_debug_set_flags) in ONE process on ONE device."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import make_inputs
from pyvbmp_amd import _lib
from pyvbmp_amd.dists import NormalInverseWishart
lib = _lib.load()
lib.vbmp_debug_set_flags.argtypes = [ctypes.c_int]
variants = [int(v, 0) for v in (sys.argv[1:] or ["0", "4"])]
for dt, D in ((torch.float64, 16), (torch.float32, 16)):
    B = 1_000_000
    SExx, SEx, N = make_inputs(B, D, dt, "cuda")
    q = NormalInverseWishart((D,), (B,), device="cuda", dtype=dt)
    times = {v: [] for v in variants}
    def _r():
        e = torch.cuda.Event(enable_timing=True); e.record(); return e
    for rnd in range(12):
        for v in variants:
            lib.vbmp_debug_set_flags(v)
            ev = []
            _lib.launch_hooks = (lambda n: ev.append(_r()), lambda n: ev.append(_r()))
            for _ in range(5):
                q.ss_update(SExx, SEx, N, lr=1.0, beta=None)
            _lib.launch_hooks = None
            torch.cuda.synchronize()
            if rnd >= 2:
                times[v] += [ev[i].elapsed_time(ev[i + 1]) for i in range(0, len(ev), 2)]
    for v in variants:
        t = sorted(times[v])
        print(f"{str(dt)[6:]} D={D} flags={v:#x}: median {t[len(t)//2]:.4f} ms  min {t[0]:.4f}", flush=True)
    lib.vbmp_debug_set_flags(0)
