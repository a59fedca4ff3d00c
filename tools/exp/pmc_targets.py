"""one pass over the other hot kernels (for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs): LDS smoother, MNW messages,
GMM E-step and moments at the BASELINE shapes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd import ops
from pyvbmp_amd.dists import MultivariateNormal_vector_format as VF
from pyvbmp_amd.models import LinearDynamicalSystems, GaussianMixtureModel
from pyvbmp_amd.transforms import MatrixNormalWishart

g = torch.Generator(device="cuda").manual_seed(0)
# config 4
y = torch.randn(1000, 4096, 6, generator=g, device="cuda", dtype=torch.float64).cumsum(0) * 0.05
m = LinearDynamicalSystems((6,), 6, latent_noise='shared', device="cuda", dtype=torch.float64)
yy, uu, rr = m.reshape_inputs(y)
for _ in range(2):
    m.update_latents(yy, uu, rr)
del m, y, yy
# config 3
N, n, p, dt = 262144, 32, 32, torch.float32
mw = MatrixNormalWishart((n, p), (), device="cuda", dtype=dt)
A = torch.randn(N, p, p + 4, generator=g, device="cuda", dtype=dt)
Px = A @ A.transpose(-2, -1) / (p + 4) + 0.5 * torch.eye(p, device="cuda", dtype=dt)
ex = torch.randn(N, p, 1, generator=g, device="cuda", dtype=dt)
for _ in range(2):
    mw.forward(VF(invSigma=Px, invSigmamu=ex))
    mw.backward(VF(invSigma=Px, invSigmamu=ex))
del A, Px
# GMM at scale
X = torch.randn(4_000_000, 16, generator=g, device="cuda", dtype=torch.float64)
gm = GaussianMixtureModel(4, 16, device="cuda", dtype=torch.float64)
gm.update(X, iters=2)
torch.cuda.synchronize()
