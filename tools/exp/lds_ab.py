"""K9 smoother kernel time at BASELINE config 4 (T = 1000, 4096 series, h = 6) with several builds of the library in ONE process:
python tools/exp/lds_ab.py default tools/exp/ab/libvbmp_X.so ...   (builds: tools/exp/build_variant.sh X "k_lds_f64 k_lds_f32" -D...)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd import _lib
from pyvbmp_amd.models import LinearDynamicalSystems
from tools.synth import lorenz


def _r():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


default = _lib.LIB_PATH
T, S = 1000, 4096
for dt in (torch.float64, torch.float32):
    y = lorenz(T, S, torch.Generator(device="cuda").manual_seed(0), device="cuda", dtype=dt)
    torch.manual_seed(0)
    m = LinearDynamicalSystems((6,), 6, latent_noise='shared', device="cuda", dtype=dt)
    inp = m.reshape_inputs(y)
    for rnd in range(2):
        for path in sys.argv[1:] or ["default"]:
            _lib._lib = None
            _lib.LIB_PATH = default if path == "default" else os.path.abspath(path)
            for _ in range(2):
                m.update_latents(*inp)
            ev = []
            _lib.launch_hooks = (lambda n: ev.append((n, _r())), lambda n: ev.append((n, _r())))
            for _ in range(5):
                m.update_latents(*inp)
            _lib.launch_hooks = None
            torch.cuda.synchronize()
            ts = sorted(ev[i][1].elapsed_time(ev[i + 1][1]) for i in range(0, len(ev), 2) if ev[i][0] == "vbmp_lds_smoother")
            print(f"{str(dt)[6:]} {os.path.basename(_lib.LIB_PATH):28s} smoother median {ts[len(ts)//2]:.3f} ms  min {ts[0]:.3f}", flush=True)
