"""K9 block form at h = 52 with several builds of the library (phase-skipping builds: -DVBMP_BLK_SKIP=1|2|4, WRONG results, timing only)
python tools/exp/lds_blk_libs.py default tools/exp/ab/libvbmp_X.so ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd import _lib
from pyvbmp_amd.models import LinearDynamicalSystems
default = _lib.LIB_PATH
h, T, S = 52, 100, 20
for dt in (torch.float64, torch.float32):
    g = torch.Generator(device="cuda").manual_seed(0)
    y = torch.randn(T, S, 6, generator=g, device="cuda", dtype=dt).cumsum(0) * 0.05
    m = LinearDynamicalSystems((6,), h, latent_noise='shared', device="cuda", dtype=dt)
    yy, uu, rr = m.reshape_inputs(y)
    m.update_latents(yy, uu, rr)
    for rnd in range(2):
        for path in sys.argv[1:] or ["default"]:
            _lib._lib = None
            _lib.LIB_PATH = default if path == "default" else os.path.abspath(path)
            ev = []

            def rec(n):
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ev.append((n, e))
            m.forward_backward_loop(yy, uu, rr)
            _lib.launch_hooks = (rec, rec)
            for _ in range(3):
                m.forward_backward_loop(yy, uu, rr)
            _lib.launch_hooks = None
            torch.cuda.synchronize()
            tk = sorted(ev[i][1].elapsed_time(ev[i + 1][1]) for i in range(0, len(ev), 2) if ev[i][0] == "vbmp_lds_smoother")
            out = (m.px.Sigma.clone(), m.px.mu.clone(), m.px.invSigma.clone())
            if rnd == 0 and path == (sys.argv[1:] or ["default"])[0]:
                first = out
            same = all(torch.equal(a, b) for a, b in zip(out, first))
            diff = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(out, first))
            print(f"{str(dt)[6:]} {os.path.basename(_lib.LIB_PATH):28s} block kernel {tk[len(tk) // 2]:.2f} ms; outputs "
                  f"{'bitwise equal to' if same else f'differ by {diff:.1e} from'} the first library's", flush=True)
