"""update_latents at BASELINE config 4: eager vs HIP-graph replay"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd.graph import GraphedStep
from pyvbmp_amd.models import LinearDynamicalSystems
from tools.synth import lorenz
T, S = 1000, 4096
for dt in (torch.float64, torch.float32):
    y = lorenz(T, S, torch.Generator(device="cuda").manual_seed(0), device="cuda", dtype=dt)
    torch.manual_seed(0)
    m = LinearDynamicalSystems((6,), 6, latent_noise='shared', device="cuda", dtype=dt)
    inp = m.reshape_inputs(y)

    def tm(fn, n=20):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    te = tm(lambda: m.update_latents(*inp))
    g = GraphedStep(m, lambda mm: mm.update_latents(*inp))
    tg = tm(lambda: g.run(1))
    tg10 = tm(lambda: g.run(10), n=3) / 10
    print(f"{str(dt)[6:]}: eager {te:.3f} ms, graph replay {tg:.3f} ms (10 per call: {tg10:.3f} ms)", flush=True)
