"""K9 at BASELINE config 4 with this build and another library (e.g. the round-2 one), default mode and literal recursion (debug flag
0x8000), interleaved in one process:  python tools/exp/lds_vs_r02.py tools/exp/ab/libvbmp_r02.so"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd import _lib
from pyvbmp_amd.models import LinearDynamicalSystems
from tools.synth import lorenz
from pyvbmp_amd import ops
default = _lib.LIB_PATH
_k13 = ops.niw_estep_params


def _k13_torch(U, nu, mu, lam, logdet_invU, alpha=None):  # (an older library has no K13: the getters' arithmetic in torch)
    D = mu.shape[-1]
    P = U * nu.reshape(-1, 1, 1)
    b = (P @ mu.unsqueeze(-1)).squeeze(-1)
    ar = torch.arange(D, device=mu.device, dtype=mu.dtype)
    c = -0.5 * ((b * mu).sum(-1) + D / lam) + 0.5 * (D * 0.6931471805599453 - logdet_invU + torch.digamma(0.5 * nu.unsqueeze(-1) - 0.5 * ar).sum(-1)) \
        - 0.5 * D * 1.8378770664093453
    return P, b, c


T, S = 1000, 4096
for dt in (torch.float64, torch.float32):
    y = lorenz(T, S, torch.Generator(device="cuda").manual_seed(0), device="cuda", dtype=dt)
    torch.manual_seed(0)
    m = LinearDynamicalSystems((6,), 6, latent_noise='shared', device="cuda", dtype=dt)
    inp = m.reshape_inputs(y)
    for rnd in range(2):
        for path in ["default"] + sys.argv[1:]:
            for flag in (0, 0x8000):
                _lib._lib = None
                _lib.LIB_PATH = default if path == "default" else os.path.abspath(path)
                try:
                    lib = _lib.load()
                except Exception as e:  # ABI version of an older build, symbols it does not have yet
                    _lib.ABI_VERSION, keep = int(str(e).split("ABI ")[1].split()[0]), _lib.ABI_VERSION
                    syms = dict(_lib.SYMBOLS)
                    _lib.SYMBOLS.pop("vbmp_niw_estep_params", None)
                    _lib._lib = None
                    lib = _lib.load()
                    _lib.ABI_VERSION = keep
                    _lib.SYMBOLS.update(syms)
                ops.niw_estep_params = _k13 if hasattr(lib, "vbmp_niw_estep_params_f64") else _k13_torch
                lib.vbmp_debug_set_flags.argtypes = [ctypes.c_int]
                lib.vbmp_debug_set_flags(flag)
                for _ in range(2):
                    m.update_latents(*inp)
                ev = []

                def rec(n):
                    e = torch.cuda.Event(enable_timing=True)
                    e.record()
                    ev.append((n, e))
                _lib.launch_hooks = (rec, rec)
                for _ in range(5):
                    m.update_latents(*inp)
                _lib.launch_hooks = None
                torch.cuda.synchronize()
                lib.vbmp_debug_set_flags(0)
                ts = sorted(ev[i][1].elapsed_time(ev[i + 1][1]) for i in range(0, len(ev), 2) if ev[i][0] == "vbmp_lds_smoother")
                print(f"{str(dt)[6:]} {os.path.basename(_lib.LIB_PATH):24s} flags {flag:#x}: smoother median {ts[len(ts) // 2]:.3f} ms  min {ts[0]:.3f}", flush=True)
    _lib._lib = None
    _lib.LIB_PATH = default
