"""config-3 forward / backward (fp32, n = p = 32, 262144 messages) with several builds of the library in ONE process on
one box: python tools/exp/mnw_fp32_ab.py lib1.so lib2.so ...  (the first argument may be 'default')"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd import _lib
from pyvbmp_amd.dists import MultivariateNormal_vector_format as VF
from pyvbmp_amd.transforms import MatrixNormalWishart

N, n, p, dt = 262144, 32, 32, torch.float32
g = torch.Generator(device="cuda").manual_seed(0)
mw = MatrixNormalWishart((n, p), (), device="cuda", dtype=dt)
A = torch.randn(N, p, p + 4, generator=g, device="cuda", dtype=dt)
Px = A @ A.transpose(-2, -1) / (p + 4) + 0.5 * torch.eye(p, device="cuda", dtype=dt)
ex = torch.randn(N, p, 1, generator=g, device="cuda", dtype=dt)
del A


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


default = _lib.LIB_PATH
for rnd in range(2):
    for path in sys.argv[1:] or ["default"]:
        _lib._lib = None
        _lib.LIB_PATH = default if path == "default" else os.path.abspath(path)
        f = t(lambda: mw.forward(VF(invSigma=Px, invSigmamu=ex)))
        b = t(lambda: mw.backward(VF(invSigma=Px, invSigmamu=ex)))
        print(f"{os.path.basename(_lib.LIB_PATH):28s} forward {f:.3f} ms  backward {b:.3f} ms", flush=True)
