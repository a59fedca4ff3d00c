"""K5b with weight columns at the DMBD shapes, against the library GEMM"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd import ops


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for dt in (torch.float64, torch.float32):
    for S, d, NB in ((24000, 53, 25), (25600, 7, 4), (262144, 32, 8), (1_000_000, 8, 8)):
        C = torch.randn(S, d, d, device="cuda", dtype=dt)
        W = torch.rand(S, NB, device="cuda", dtype=dt)
        a = t(lambda: ops.weighted_matsum_cols(C, W))
        b = t(lambda: W.t() @ C.reshape(S, -1))
        gb = C.numel() * C.element_size() / 1e9
        print(f"{str(dt)[6:]} S={S} d={d} NB={NB}: K5b {a * 1e3:.0f} us ({gb / a * 1e3:.0f} GB/s)   library {b * 1e3:.0f} us", flush=True)
