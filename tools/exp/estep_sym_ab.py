"""symmetric-packed E-step (k_estep_sym) with several builds (samples per thread): python tools/exp/estep_sym_ab.py default lib.so .."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd import _lib, ops


def _r():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


default = _lib.LIB_PATH
for dt in (torch.float64, torch.float32):
    for N, K, D in ((4_000_000, 4, 16), (4_000_000, 8, 8), (2_000_000, 8, 16), (2_000_000, 4, 32)):
        g = torch.Generator(device="cuda").manual_seed(0)
        X = torch.randn(N, D, generator=g, device="cuda", dtype=dt)
        A = torch.randn(K, D, D + 2, generator=g, device="cuda", dtype=dt)
        P = A @ A.transpose(-2, -1) / D
        b = torch.randn(K, D, generator=g, device="cuda", dtype=dt)
        c = torch.randn(K, generator=g, device="cuda", dtype=dt)
        out = []
        for rnd in range(2):
            for path in sys.argv[1:] or ["default"]:
                _lib._lib = None
                _lib.LIB_PATH = default if path == "default" else os.path.abspath(path)
                for _ in range(3):
                    ops.mixture_estep(X, P, b, c)
                ev = []
                _lib.launch_hooks = (lambda n: ev.append(_r()), lambda n: ev.append(_r()))
                for _ in range(10):
                    ops.mixture_estep(X, P, b, c)
                _lib.launch_hooks = None
                torch.cuda.synchronize()
                ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(0, len(ev), 2))
                if rnd == 1:
                    out.append(f"{os.path.basename(_lib.LIB_PATH)[8:-3] or 'default'} {ts[len(ts) // 2]:.3f}")
        print(f"{str(dt)[6:]} N={N} K={K} D={D}: " + "  ".join(out) + " ms", flush=True)
