"""K9 fixed-point shortcut: does the bench model reach a bitwise fixed point, and what does it buy?"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd import _lib
from pyvbmp_amd.models import LinearDynamicalSystems
from tools.synth import lorenz
lib = _lib.load()
lib.vbmp_debug_set_flags.argtypes = [ctypes.c_int]


def _r():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


for dt in ((torch.float64, torch.float32) if 'cycle' not in sys.argv else ()):
    for T, S, seed in ((1000, 4096, 0), (1000, 4096, 1), (400, 512, 2)):
        y = lorenz(T, S, torch.Generator(device="cuda").manual_seed(seed), device="cuda", dtype=dt)
        torch.manual_seed(seed)
        m = LinearDynamicalSystems((6,), 6, latent_noise='shared', device="cuda", dtype=dt)
        inp = m.reshape_inputs(y)
        for flag in (0x8000, 0):
            lib.vbmp_debug_set_flags(flag)
            for _ in range(2):
                m.update_latents(*inp)
            ev = []
            _lib.launch_hooks = (lambda n: ev.append((n, _r())), lambda n: ev.append((n, _r())))
            for _ in range(3):
                m.update_latents(*inp)
            _lib.launch_hooks = None
            torch.cuda.synchronize()
            ts = sorted(ev[i][1].elapsed_time(ev[i + 1][1]) for i in range(0, len(ev), 2) if ev[i][0] == "vbmp_lds_smoother")
            Sg = m.px.Sigma[:, 0]
            d = (Sg[1:] - Sg[:-1]).abs().amax((-1, -2))
            P = m.px.invSigma[:, 0]
            dP = (P[1:] - P[:-1]).abs().amax((-1, -2))
            print(f"{str(dt)[6:]} T={T} S={S} seed={seed} flag={flag:#x}: smoother {ts[len(ts)//2]:.3f} ms; steps where smoothed Sigma changes: {int((d > 0).sum())} "
                  f"(precision: {int((dP > 0).sum())}), mid-range max change {float(d[T//3:2*T//3].max()):.2e}", flush=True)
        lib.vbmp_debug_set_flags(0)

# periodicity of the mid-range (is the non-converging case a last-bit limit cycle?)
for dt, T, S, seed in ((torch.float64, 1000, 64, 0), (torch.float64, 400, 64, 2), (torch.float32, 400, 64, 2)):
    y = lorenz(T, S, torch.Generator(device="cuda").manual_seed(seed), device="cuda", dtype=dt)
    torch.manual_seed(seed)
    m = LinearDynamicalSystems((6,), 6, latent_noise='shared', device="cuda", dtype=dt)
    lib.vbmp_debug_set_flags(0x8000)
    m.update_latents(*m.reshape_inputs(y))
    lib.vbmp_debug_set_flags(0)
    for name in ("invSigma", "Sigma"):
        M = getattr(m.px, name)[:, 0]
        mid = M[T // 3: 2 * T // 3]
        per = {p: bool(torch.equal(mid[p:], mid[:-p])) for p in (1, 2, 3, 4, 6, 8)}
        amp = float((mid - mid[0]).abs().max() / mid.abs().max())
        print(f"{str(dt)[6:]} seed={seed} {name}: periodic with period {[p for p, v in per.items() if v]}, amplitude {amp:.1e} relative")
