"""the mixtures of linear transforms run SLOWER in fp32 than in fp64.  A few VB iterations of one model in one
dtype, to be run under `rocprofv3 --kernel-trace --stats` once per dtype (the kernel lists are then diffed):
    python3 tools/exp/mixlt_dtype_prof.py mixlt|dmix f64|f32"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
which, dts = sys.argv[1], sys.argv[2]
dt = torch.float64 if dts == "f64" else torch.float32
N, n, p, K = 1_000_000, 8, 8, 8
g = torch.Generator(device="cuda").manual_seed(0)
if which == "mixlt":
    from pyvbmp_amd.transforms import MixtureofLinearTransforms
    X = torch.randn(N, p, 1, generator=g, device="cuda", dtype=dt)
    Ws = torch.randn(K, n, p, generator=g, device="cuda", dtype=dt)
    z = torch.randint(K, (N,), generator=g, device="cuda")
    Y = Ws[z] @ X + 0.1 * torch.randn(N, n, 1, generator=g, device="cuda", dtype=dt)
    m = MixtureofLinearTransforms(n, p, K, device="cuda", dtype=dt)
else:
    from pyvbmp_amd.transforms import dMixtureofLinearTransforms
    X = torch.randn(N, p, generator=g, device="cuda", dtype=dt)
    Ws = torch.randn(K, n, p, generator=g, device="cuda", dtype=dt)
    z = ((X[:, :3] > 0).long() * torch.tensor([1, 2, 4], device="cuda")).sum(-1)
    Y = (Ws[z] @ X.unsqueeze(-1)).squeeze(-1) + 0.1 * torch.randn(N, n, generator=g, device="cuda", dtype=dt)
    m = dMixtureofLinearTransforms(n, p, K, device="cuda", dtype=dt)
m.raw_update(X, Y, iters=2)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
m.raw_update(X, Y, iters=10)
e1.record()
torch.cuda.synchronize()
print(f"{which} {dts}: {e0.elapsed_time(e1) / 10:.3f} ms per VB iteration")
