"""K12 (vbmp_rows_affine) against the library GEMM for tall-skinny products"""
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
    for S, k, n in ((4_096_000, 6, 6), (1_000_000, 7, 8), (1_000_000, 9, 7), (1_000_000, 32, 32), (262_144, 64, 64), (100_000, 8, 8)):
        X = torch.randn(S, k, device="cuda", dtype=dt)
        W = torch.randn(k, n, device="cuda", dtype=dt)
        Wt = W.t().contiguous()
        a, b = t(lambda: ops.rows_affine(X, Wt)), t(lambda: X @ W)
        gb = S * (k + n) * X.element_size() / 1e9
        print(f"{str(dt)[6:]} S={S} k={k} n={n}: K12 {a * 1e3:.0f} us ({gb / a * 1e3:.0f} GB/s)   library {b * 1e3:.0f} us ({gb / b * 1e3:.0f} GB/s)", flush=True)
