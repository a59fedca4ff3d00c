"""config-3 messages against the size of the persistent grid (blocks per CU), one process, interleaved"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd import _lib
from pyvbmp_amd.dists import MultivariateNormal_vector_format as VF
from pyvbmp_amd.transforms import MatrixNormalWishart
lib = _lib.load()
lib.vbmp_debug_set_blocks_per_cu.argtypes = [ctypes.c_int]
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


for rnd in range(2):
    for v in [int(x) for x in (sys.argv[1:] or ["0", "2", "4", "8", "32"])]:
        lib.vbmp_debug_set_blocks_per_cu(v)
        f = t(lambda: mw.forward(VF(invSigma=Px, invSigmamu=ex)))
        b = t(lambda: mw.backward(VF(invSigma=Px, invSigmamu=ex)))
        print(f"blocks/CU {v:3d}: forward {f:.3f} ms  backward {b:.3f} ms", flush=True)
lib.vbmp_debug_set_blocks_per_cu(0)
