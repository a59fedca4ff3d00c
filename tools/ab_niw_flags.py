#!/usr/bin/env python3
"""Interleaved A/B of K2 debug flags in ONE process on ONE device: python tools/ab_niw_flags.py 0 0x200 ...
(flags: 1 = non-temporal loads, 2 = non-temporal stores, 0x200 = software pipeline off)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_inputs
from pyvbmp_amd import _lib
from pyvbmp_amd.dists import NormalInverseWishart

lib = _lib.load()
lib.vbmp_debug_set_flags.argtypes = [ctypes.c_int]
variants = [int(v, 0) for v in (sys.argv[1:] or ["0", "0x200"])]
cases = [(torch.float64, 16, 1.0), (torch.float32, 16, 1.0), (torch.float64, 8, 1.0), (torch.float64, 16, 0.5)]
for dt, D, lr in cases:
    B = 1_000_000
    SExx, SEx, N = make_inputs(B, D, dt, "cuda")
    q = NormalInverseWishart((D,), (B,), device="cuda", dtype=dt)
    q.ss_update(SExx, SEx, N, lr=1.0, beta=None)
    times = {v: [] for v in variants}

    def _r():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e
    for rnd in range(12):
        for v in variants:
            lib.vbmp_debug_set_flags(v)
            ev = []
            _lib.launch_hooks = (lambda n: ev.append(_r()), lambda n: ev.append(_r()))
            for _ in range(5):
                q.ss_update(SExx, SEx, N, lr=lr, beta=None)
            _lib.launch_hooks = None
            torch.cuda.synchronize()
            if rnd >= 2:
                times[v] += [ev[i].elapsed_time(ev[i + 1]) for i in range(0, len(ev), 2)]
    it = 8 if dt == torch.float64 else 4
    bpu = (3 * D * D + 2 * D + 4 + (0 if lr == 1.0 else D * D + D + 2)) * it
    for v in variants:
        t = sorted(times[v])
        med = t[len(t) // 2]
        print(f"{str(dt)[6:]} D={D} lr={lr} flags={v:#x}: median {med:.4f} ms  min {t[0]:.4f}  -> {bpu * B / med / 1e9 * 1e3 / 8000:.3f} of 8 TB/s", flush=True)
    lib.vbmp_debug_set_flags(0)
    del q, SExx, SEx, N
