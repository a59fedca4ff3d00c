"""mixture E-step (K3) kernel time: fused MFMA form (default) against MFMA + separate softmax pass (flag 0x4000) and the
VALU form (0x100), one process, interleaved"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd import _lib, ops
lib = _lib.load()
lib.vbmp_debug_set_flags.argtypes = [ctypes.c_int]


NAMES = {0x4000: "two kernels", 0: "fused MFMA ", 0x100: "VALU form  ", -1: "sym VALU   "}


def _r():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


for dt in (torch.float64, torch.float32):
    for N, K, D in ((4_000_000, 4, 16), (1_000_000, 16, 16), (2_000_000, 8, 32)):
        g = torch.Generator(device="cuda").manual_seed(0)
        X = torch.randn(N, D, generator=g, device="cuda", dtype=dt)
        A = torch.randn(K, D, D + 2, generator=g, device="cuda", dtype=dt)
        P = A @ A.transpose(-2, -1) / D
        b = torch.randn(K, D, generator=g, device="cuda", dtype=dt)
        c = torch.randn(K, generator=g, device="cuda", dtype=dt)
        ref = None
        for flag in (0, 0x100, -1, 0, 0x100, -1):
            ops._estep_sym_off = flag != -1
            if flag == -1 and (K > ops.ESTEP_SYM_MAX_K):
                continue
            lib.vbmp_debug_set_flags(max(flag, 0))
            for _ in range(3):
                ops.mixture_estep(X, P, b, c)
            ev = []
            _lib.launch_hooks = (lambda n: ev.append(_r()), lambda n: ev.append(_r()))
            for _ in range(10):
                ops.mixture_estep(X, P, b, c)
            _lib.launch_hooks = None
            torch.cuda.synchronize()
            ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(0, len(ev), 2))
            pp, NA, lz = ops.mixture_estep(X, P, b, c)
            if ref is None:
                ref = (pp.clone(), NA.clone(), lz.clone())
            err = max(float((pp - ref[0]).abs().max()), float(((NA - ref[1]) / ref[1]).abs().max()), float(((lz - ref[2]) / ref[2]).abs()))
            it = 8 if dt == torch.float64 else 4
            print(f"{str(dt)[6:]} N={N} K={K} D={D} {NAMES[flag]}: {ts[len(ts)//2]:.3f} ms "
                  f"({(D + K) * it * N / ts[len(ts)//2] / 8e9 * 1e3:.3f} of 8 TB/s on (D + K) values per sample); max dev from the first form {err:.1e}", flush=True)
        lib.vbmp_debug_set_flags(0)
