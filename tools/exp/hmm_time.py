"""K11 timing at the two DMBD shapes (role chain of the flocking and the Lorenz-like case)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd import ops, _lib

if len(sys.argv) > 1:  # A/B against another build of the library
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])

dev = torch.device("cuda")
for dt in (torch.float64, torch.float32):
    for (T, C, K) in ((100, 240, 25), (400, 64, 4), (1000, 4096, 8), (100, 240, 16), (100, 240, 12), (100, 240, 32), (100, 240, 40)):
        torch.manual_seed(0)
        lg = torch.randn(T, C, K, dtype=dt, device=dev) * 3
        tr = torch.log_softmax(torch.randn(K, K, dtype=dt, device=dev), -1)
        ini = torch.log_softmax(torch.randn(K, dtype=dt, device=dev), -1)
        for _ in range(3):
            ops.hmm_forward_backward(lg, tr, ini, ())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.hmm_forward_backward(lg, tr, ini, ())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"hmm {dt} T={T} C={C} K={K}: {ms:.3f} ms  ({ms / T * 1e3:.2f} us / time step)")
