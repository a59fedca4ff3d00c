"""K2 / K1 at the unpadded sizes 6, 12, 20: D as a compile-time constant (default) against the run-time-D generic instance
(flag 0x2000), one process, interleaved"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import make_inputs
from pyvbmp_amd import _lib, ops
from pyvbmp_amd.dists import NormalInverseWishart
lib = _lib.load()
lib.vbmp_debug_set_flags.argtypes = [ctypes.c_int]


def _r():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def timed(fn):
    ts = []
    for rnd in range(8):
        ev = []
        _lib.launch_hooks = (lambda n: ev.append(_r()), lambda n: ev.append(_r()))
        for _ in range(5):
            fn()
        _lib.launch_hooks = None
        torch.cuda.synchronize()
        if rnd >= 2:
            ts += [ev[i].elapsed_time(ev[i + 1]) for i in range(0, len(ev), 2)]
    ts.sort()
    return ts[len(ts) // 2]


for dt in (torch.float64, torch.float32):
    for D in (6, 12, 20):
        B = 1_000_000 if D <= 12 else 400_000
        SExx, SEx, N = make_inputs(B, D, dt, "cuda")
        q = NormalInverseWishart((D,), (B,), device="cuda", dtype=dt)
        it = 8 if dt == torch.float64 else 4
        for flag in (0x2000, 0, 0x2000, 0):
            lib.vbmp_debug_set_flags(flag)
            t2 = timed(lambda: q.ss_update(SExx, SEx, N, lr=1.0, beta=None))
            t1 = timed(lambda: ops.spd_inv_logdet(SExx))
            print(f"{str(dt)[6:]} D={D:2d} B={B} {'run-time D   ' if flag else 'compile-time D'}: K2 {t2:.4f} ms ({(3*D*D+2*D+4)*it*B/t2/8e9*1e3:.3f} of 8 TB/s)"
                  f"   K1 {t1:.4f} ms ({(2*D*D+1)*it*B/t1/8e9*1e3:.3f})", flush=True)
        lib.vbmp_debug_set_flags(0)
        del q, SExx, SEx, N
