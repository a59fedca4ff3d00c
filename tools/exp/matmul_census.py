"""Which library GEMMs does one VB iteration of a model issue, with which shapes, and what does each cost?  (synchronising timer
around every torch matmul / bmm / mm / addmm call; development aid)
    python3 tools/exp/matmul_census.py mixlt|dmix f64|f32"""
import os, sys, time, collections
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
stats = collections.defaultdict(lambda: [0, 0.0])
import traceback


def wrap(name, fn):
    def f(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        dtm = (time.perf_counter() - t0) * 1e3
        shapes = tuple(tuple(x.shape) for x in a if isinstance(x, torch.Tensor))
        strides = tuple(tuple(x.stride()) for x in a if isinstance(x, torch.Tensor))
        fr = traceback.extract_stack(limit=4)[0:3]
        where = " <- ".join(f"{os.path.basename(x.filename)}:{x.lineno}" for x in reversed(fr))
        key = (name, shapes, strides, where)
        stats[key][0] += 1
        stats[key][1] += dtm
        return r
    return f


torch.matmul = wrap("matmul", torch.matmul)
torch.Tensor.__matmul__ = wrap("@", torch.Tensor.__matmul__)
torch.Tensor.matmul = wrap("T.matmul", torch.Tensor.matmul)
torch.bmm = wrap("bmm", torch.bmm)
torch.mm = wrap("mm", torch.mm)
torch.einsum = wrap("einsum", torch.einsum)
m.raw_update(X, Y, iters=1)
for k, (c, t) in sorted(stats.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{t:8.3f} ms {c:3d}x {k[0]:8s} {k[1]} strides {k[2]}  at {k[3]}")
