"""K2 headline: kernel time of back-to-back launches against launches separated by idle gaps (host sleep after a synchronise)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import make_inputs
from pyvbmp_amd import _lib
from pyvbmp_amd.dists import NormalInverseWishart
B, D, dt = 1_000_000, 16, torch.float64
SExx, SEx, N = make_inputs(B, D, dt, "cuda")
q = NormalInverseWishart((D,), (B,), device="cuda", dtype=dt)


def _r():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


for rnd in range(3):
    for gap_ms in (0, 1, 5, 20):
        ev = []
        _lib.launch_hooks = (lambda n: ev.append(_r()), lambda n: ev.append(_r()))
        for _ in range(12):
            q.ss_update(SExx, SEx, N, lr=1.0, beta=None)
            if gap_ms:
                torch.cuda.synchronize()
                time.sleep(gap_ms * 1e-3)
        _lib.launch_hooks = None
        torch.cuda.synchronize()
        ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(4, len(ev), 2))
        print(f"gap {gap_ms:2d} ms: median {ts[len(ts) // 2]:.4f} ms  min {ts[0]:.4f}  max {ts[-1]:.4f}", flush=True)
