"""K2 headline: does the kernel time depend on WHERE the caching allocator put the two 2 GB outputs of the launch?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import make_inputs
from pyvbmp_amd import _lib
from pyvbmp_amd.dists import NormalInverseWishart
B, D, dt = 1_000_000, 16, torch.float64
SExx, SEx, N = make_inputs(B, D, dt, "cuda")
q = NormalInverseWishart((D,), (B,), device="cuda", dtype=dt)
if len(sys.argv) > 1:  # grid cap in blocks per CU (0 = the library default)
    import ctypes
    _l = _lib.load(); _l.vbmp_debug_set_blocks_per_cu.argtypes = [ctypes.c_int]; _l.vbmp_debug_set_blocks_per_cu(int(sys.argv[1]))
    print("grid cap", sys.argv[1])


def _r():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


rows = []
for it in range(40):
    ev = []
    _lib.launch_hooks = (lambda n: ev.append(_r()), lambda n: ev.append(_r()))
    q.ss_update(SExx, SEx, N, lr=1.0, beta=None)
    _lib.launch_hooks = None
    torch.cuda.synchronize()
    rows.append((ev[0].elapsed_time(ev[1]), q.invU.invU.data_ptr(), q.invU.U.data_ptr()))
base = SExx.data_ptr()
by = {}
for t, a, b in rows[4:]:
    by.setdefault(((a - base) >> 20, (b - base) >> 20), []).append(t)
print("SExx at", hex(base))
for k, v in sorted(by.items()):
    v.sort()
    print(f"invU at +{k[0]} MiB, U at +{k[1]} MiB: {len(v)} launches, median {v[len(v) // 2]:.4f} ms, min {v[0]:.4f}, max {v[-1]:.4f}")
