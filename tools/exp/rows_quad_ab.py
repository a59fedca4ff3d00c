"""vbmp_rows_affine_quad: rows staged through LDS (default) against a row per thread from global memory (VBMP_DBG_ROWS_DIRECT = 0x8)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd import _lib, ops
lib = _lib.load()
lib.vbmp_debug_set_flags.argtypes = [ctypes.c_int]
for dt in (torch.float64, torch.float32):
    for (S, k, n) in ((4096000, 6, 6), (4096000, 4, 12), (1000000, 12, 6), (4096000, 2, 2)):
        torch.manual_seed(0)
        X = torch.randn(S, k, device="cuda", dtype=dt)
        M, c = torch.randn(n, k, device="cuda", dtype=dt), torch.randn(n, device="cuda", dtype=dt)
        A = torch.randn(k, k, device="cuda", dtype=dt)
        P, b, c0 = A @ A.T, torch.randn(k, device="cuda", dtype=dt), torch.randn((), device="cuda", dtype=dt)
        res = {}
        for name, flag in (("staged", 0), ("direct", 0x8), ("staged", 0), ("direct", 0x8)):
            lib.vbmp_debug_set_flags(flag)
            out = ops.rows_affine_quad(X, M, c, P, b, c0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(7):
                e0.record(); out = ops.rows_affine_quad(X, M, c, P, b, c0); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            t = sorted(ts)[3]
            byts = S * (k + n + 1) * X.element_size()
            res.setdefault(name, (t, out))
            print(f"{str(dt)[6:]} S={S} k={k} n={n} {name}: {t:.4f} ms -> {byts / t / 1e6:.0f} GB/s", flush=True)
        lib.vbmp_debug_set_flags(0)
        same = all(torch.equal(a, b) for a, b in zip(res["staged"][1], res["direct"][1]))
        print("   outputs", "bitwise equal" if same else "DIFFER")
