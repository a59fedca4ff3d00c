"""where does MatrixNormalWishart.update spend its time (BASELINE config 3 shape)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd import ops
from pyvbmp_amd.dists import MultivariateNormal_vector_format as VF
from pyvbmp_amd.transforms import MatrixNormalWishart
from pyvbmp_amd.transforms.MatrixNormalWishart import _cov_of


def tm(f, reps=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


N, n, p, dt = 262144, 32, 32, torch.float32
g = torch.Generator(device="cuda").manual_seed(0)
m = MatrixNormalWishart((n, p), (), device="cuda", dtype=dt)
A = torch.randn(N, p, p + 4, generator=g, device="cuda", dtype=dt)
Sx = A @ A.transpose(-2, -1) / (p + 4) + 0.5 * torch.eye(p, device="cuda", dtype=dt)
Sy = Sx.clone()
mux = torch.randn(N, p, 1, generator=g, device="cuda", dtype=dt)
muy = torch.randn(N, n, 1, generator=g, device="cuda", dtype=dt)
pX, pY = VF(mu=mux, Sigma=Sx), VF(mu=muy, Sigma=Sy)
print("update        ", tm(lambda: m.update(pX, pY)))
print("EX,EX,cov,cov ", tm(lambda: (pX.EX(), pY.EX(), _cov_of(pX), _cov_of(pY))))
st = m._moments(pX.EX(), pY.EX(), _cov_of(pX), _cov_of(pY), None)
print("_moments      ", tm(lambda: m._moments(pX.EX(), pY.EX(), _cov_of(pX), _cov_of(pY), None)))
print("ss_update     ", tm(lambda: m.ss_update(*st)))
z = torch.cat((mux, muy), -2).squeeze(-1)
print("cat           ", tm(lambda: torch.cat((mux, muy), -2).squeeze(-1)))
print("wmoments      ", tm(lambda: ops.weighted_moments(z, None, 1, ())))
print("matsum        ", tm(lambda: ops.weighted_matsum(Sx.reshape(N, p, p), None)))
print("torch sum(0)  ", tm(lambda: Sx.sum(0)))
