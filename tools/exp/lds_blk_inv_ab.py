"""K9 block form, library built with -DVBMP_BLK_INV_MFMA (tools/exp/build_variant.sh NAME "k_lds_f64 k_lds_f32" -DVBMP_BLK_INV_MFMA
-mllvm -amdgpu-mfma-vgpr-form; pass its path as argv[2]): inverses as block Gauss-Jordan on the matrix cores against one wave per matrix
(VBMP_DBG_BLK_INV_WAVES = 0x40000000): smoother time per call and the difference of the outputs."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd import _lib
from pyvbmp_amd.models import LinearDynamicalSystems
cases = ((52, 100, 20), (52, 100, 300), (12, 200, 64), (21, 50, 7), (33, 50, 20), (60, 30, 5))
if len(sys.argv) > 1:
    cases = cases[:int(sys.argv[1])]
if len(sys.argv) > 2:
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
lib = _lib.load()
lib.vbmp_debug_set_flags.argtypes = [ctypes.c_int]
for dt in (torch.float64, torch.float32):
    for (h, T, S) in cases:
        g = torch.Generator(device="cuda").manual_seed(0)
        y = torch.randn(T, S, 6, generator=g, device="cuda", dtype=dt).cumsum(0) * 0.05
        torch.manual_seed(1)
        m = LinearDynamicalSystems((6,), h, latent_noise='shared', device="cuda", dtype=dt)
        yy, uu, rr = m.reshape_inputs(y)
        m.update_latents(yy, uu, rr)
        res = {}
        for name, flag in (("mfma", 0), ("waves", 0x40000000)):
            lib.vbmp_debug_set_flags(flag)
            outs = m.forward_backward_loop(yy, uu, rr)
            ev = []

            def rec(n):
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ev.append((n, e))
            _lib.launch_hooks = (rec, rec)
            for _ in range(3):
                outs = m.forward_backward_loop(yy, uu, rr)
            _lib.launch_hooks = None
            torch.cuda.synchronize()
            tk = sorted(ev[i][1].elapsed_time(ev[i + 1][1]) for i in range(0, len(ev), 2) if ev[i][0] == "vbmp_lds_smoother")
            res[name] = (tk[len(tk) // 2], [m.px.Sigma.clone(), m.px.mu.clone(), m.px.invSigma.clone(), outs[0].clone(), outs[1].clone()])
        lib.vbmp_debug_set_flags(0)
        a, b = res["mfma"], res["waves"]
        worst = max(float((x - y).abs().max() / y.abs().max().clamp_min(1e-300)) for x, y in zip(a[1], b[1]))
        print(f"{str(dt)[6:]} h={h} T={T} S={S}: matrix cores {a[0]:.3f} ms, wave per matrix {b[0]:.3f} ms; outputs differ by {worst:.1e}", flush=True)
