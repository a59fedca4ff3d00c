"""a few launches of the fp32 / fp64 MFMA moments kernel (for rocprofv3 --pmc runs)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd import ops
for dt in (torch.float32, torch.float64):
    N, K, D = 4_000_000, 4, 16
    g = torch.Generator(device="cuda").manual_seed(0)
    X = torch.randn(N, 1, D, generator=g, device="cuda", dtype=dt)
    p = torch.rand(N, K, generator=g, device="cuda", dtype=dt)
    for _ in range(5):
        ops.weighted_moments(X, p, 1, (K,))
    torch.cuda.synchronize()
