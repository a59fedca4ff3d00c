"""Error growth of the LDS filter / smoother recursion over T (BASELINE configs[3] shape: hidden 6, obs 6, Lorenz data):
the K9 kernel in fp32 and fp64 against the fp64 oracle, next to the ORACLE ITSELF run in fp32 (the reference's algorithm
at that precision, CPU) against the fp64 oracle.  If the two fp32 columns grow alike, the looser tolerance of the long
recursion tests is the problem's conditioning at fp32, not the kernel's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import lds as olds, mnw as omnw, niw as oniw
from pyvbmp_amd.models import LinearDynamicalSystems
from tests.helpers import relerr
from tools.synth import lorenz

S, h = 32, 6
print("| T | kernel fp64 vs oracle fp64 | kernel fp32 vs oracle fp64 | oracle fp32 vs oracle fp64 |")
print("|---|---|---|---|")
for T in (1, 2, 5, 20, 50, 100, 200, 400):
    y = lorenz(T + 3, S, torch.Generator().manual_seed(7), device="cpu")[3:]
    out = {}
    for dt in (torch.float64, torch.float32):
        torch.manual_seed(12)
        m = LinearDynamicalSystems((6,), h, latent_noise='shared', device="cuda", dtype=dt)
        if dt == torch.float32:  # the same (rounded) initial state as the fp64 model
            src = out[torch.float64][0]
            m.x0.mu, m.A.mu, m.obs_model.mu = src.x0.mu.float(), src.A.mu.float(), src.obs_model.mu.float()
            m.set_latent_parms()
        m.update_latents(*m.reshape_inputs(y.to(dt).cuda()))
        out[dt] = (m, {f: getattr(m.px, f).cpu().double() for f in ("mu", "Sigma", "invSigma", "invSigmamu")}, m.logZ.cpu().double())
    m64 = out[torch.float64][0]

    def oracle(dt):
        x0 = oniw.niw_new((h,), (), mu_init=m64.x0.mu.cpu().to(dt), dtype=dt)
        A = omnw.mnw_new((h, h + 1), (), mu_init=m64.A.mu.cpu().to(dt), dtype=dt)
        obs = omnw.mnw_new((6, h + 1), (), mu_init=m64.obs_model.mu.cpu().to(dt), dtype=dt)
        yo, uo, ro = olds.reshape_inputs(y.to(dt), None, None, (6,), 1, 1)
        sm = olds.smoother(olds.latent_parms(A, h), x0, h, yo, uo, ro, obs, 0)
        st = olds.latent_stats(sm, yo, uo, ro, (6,), 1, 1, (), 0)
        return sm, st["logZ"]
    ref, ref_lz = oracle(torch.float64)
    o32, o32_lz = oracle(torch.float32)

    def worst(px, lz):
        return max([relerr(px[f], ref[f]) for f in ("mu", "Sigma", "invSigma", "invSigmamu")] + [relerr(lz, ref_lz)])
    k64 = worst(out[torch.float64][1], out[torch.float64][2])
    k32 = worst(out[torch.float32][1], out[torch.float32][2])
    r32 = worst({f: o32[f].double() for f in ("mu", "Sigma", "invSigma", "invSigmamu")}, o32_lz.double())
    print(f"| {T} | {k64:.1e} | {k32:.1e} | {r32:.1e} |", flush=True)
