"""config-3 backward message: the kernel with the marginal precision eliminated explicitly (Add2 given) against Schur mode
(Add2 = NULL), same library, same box, interleaved"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd import ops
from pyvbmp_amd.transforms import MatrixNormalWishart

N, n, p, dt = 262144, 32, 32, torch.float32
g = torch.Generator(device="cuda").manual_seed(0)
m = MatrixNormalWishart((n, p), (), device="cuda", dtype=dt)
X = torch.randn(4096, p, 1, generator=g, device="cuda", dtype=dt)
Y = torch.randn(n, p, generator=g, device="cuda", dtype=dt) / p ** 0.5 @ X + 0.3 * torch.randn(4096, n, 1, generator=g, device="cuda", dtype=dt)
m.raw_update(X, Y)
A = torch.randn(N, n, n + 4, generator=g, device="cuda", dtype=dt)
Py = A @ A.transpose(-2, -1) / (n + 4) + 0.5 * torch.eye(n, device="cuda", dtype=dt)
etay = torch.randn(N, n, 1, generator=g, device="cuda", dtype=dt)
del A
Rm, G, H = m.EinvSigma(), m.EinvUX(), m.EXTinvUX()
jx = torch.zeros(p, 1, device="cuda", dtype=dt)
Hinv = ops.spd_inverse(H)
K = Rm - G @ Hinv @ G.T
eta_y = etay + (G @ Hinv) @ jx


def run(add2):
    return ops.mnw_message(Py, etay.squeeze(-1), etay.squeeze(-1), eta_y.squeeze(-1), Rm, add2, G.T.contiguous(), H, jx.squeeze(-1), -1.0, ())


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


a, b = run(K), run(None)
ldH = torch.logdet(H.double()).float()
print("q3 rel diff", float(((a[2][:, 4] - b[2][:, 4]).abs().max() / a[2][:, 4].abs().max())),
      "ld3 rel diff", float(((a[2][:, 5] - (b[2][:, 5] - ldH)).abs().max() / a[2][:, 5].abs().max())))
for rnd in range(3):
    print(f"explicit (4 eliminations) {t(lambda: run(K)):.3f} ms   Schur mode (3 eliminations) {t(lambda: run(None)):.3f} ms", flush=True)
# accuracy of the two routes against fp64 torch on a sample
idx = torch.arange(0, N, 1021, device="cuda")
Pd, Kd, e3d = Py[idx].double(), K.double(), eta_y[idx].double()
Kd64 = Rm.double() - G.double() @ torch.linalg.inv(H.double()) @ G.double().T
for name, Kref in (("K rounded to fp32 (what the explicit route is given)", Kd), ("K in fp64", Kd64)):
    Mx = Pd + Kref
    q3 = (e3d.transpose(-2, -1) @ torch.linalg.solve(Mx, e3d)).squeeze(-1).squeeze(-1)
    ld3 = torch.logdet(Mx)
    for nm, r, off in (("explicit", a, 0.0), ("schur", b, float(ldH))):
        eq = float((r[2][idx, 4].double() - q3).abs().max() / q3.abs().max())
        el = float((r[2][idx, 5].double() - off - ld3).abs().max() / ld3.abs().max())
        print(f"{name}: {nm:8s} q3 err {eq:.2e}  ld3 err {el:.2e}")
