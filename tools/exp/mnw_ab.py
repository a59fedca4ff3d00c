"""MNW forward / backward at BASELINE config 3, optionally with another build of the library (A/B on one box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from pyvbmp_amd.dists import MultivariateNormal_vector_format as VF
from pyvbmp_amd.transforms import MatrixNormalWishart


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for dt in (torch.float32, torch.float64):
    for n in (32, 16):
        N, p = 262144, n
        g = torch.Generator(device="cuda").manual_seed(0)
        mw = MatrixNormalWishart((n, p), (), device="cuda", dtype=dt)
        A = torch.randn(N, p, p + 4, generator=g, device="cuda", dtype=dt)
        Px = A @ A.transpose(-2, -1) / (p + 4) + 0.5 * torch.eye(p, device="cuda", dtype=dt)
        ex = torch.randn(N, p, 1, generator=g, device="cuda", dtype=dt)
        f = t(lambda: mw.forward(VF(invSigma=Px, invSigmamu=ex)))
        b = t(lambda: mw.backward(VF(invSigma=Px, invSigmamu=ex)))
        print(f"{os.path.basename(_lib.LIB_PATH)} {str(dt)[6:]} n=p={n}: forward {f:.3f} ms  backward {b:.3f} ms", flush=True)
