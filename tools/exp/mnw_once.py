"""a few launches of the MNW message kernel at BASELINE config 3 (for rocprofv3 --pmc runs)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd.dists import MultivariateNormal_vector_format as VF
from pyvbmp_amd.transforms import MatrixNormalWishart
N, n, p, dt = 262144, 32, 32, torch.float32
g = torch.Generator(device="cuda").manual_seed(0)
mw = MatrixNormalWishart((n, p), (), device="cuda", dtype=dt)
A = torch.randn(N, p, p + 4, generator=g, device="cuda", dtype=dt)
Px = A @ A.transpose(-2, -1) / (p + 4) + 0.5 * torch.eye(p, device="cuda", dtype=dt)
ex = torch.randn(N, p, 1, generator=g, device="cuda", dtype=dt)
for _ in range(3):
    mw.forward(VF(invSigma=Px, invSigmamu=ex))
for _ in range(3):
    mw.backward(VF(invSigma=Px, invSigmamu=ex))
torch.cuda.synchronize()
