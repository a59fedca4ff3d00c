"""K2 headline with its three 2 GB streams (SExx read; invU, U written) at CONTROLLED relative offsets inside one allocation: which offsets are
slow?  Direct C-ABI calls (the product's own argument marshalling, outputs as views of the pool)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import make_inputs
from pyvbmp_amd import _lib as L
from pyvbmp_amd.dists import NormalInverseWishart
B, D, dt = 1_000_000, 16, torch.float64
SExx0, SEx, N = make_inputs(B, D, dt, "cuda")
q = NormalInverseWishart((D,), (B,), device="cuda", dtype=dt)
lib = L.load()
fn = lib.vbmp_niw_ss_update_f64
nbytes = B * D * D * 8
GiB = 1 << 30
pool = torch.empty(3 * nbytes + 3 * GiB, dtype=torch.uint8, device="cuda")
base = (pool.data_ptr() + (1 << 21) - 1) // (1 << 21) * (1 << 21) - pool.data_ptr()  # 2 MiB aligned start


def view(off):
    return pool[base + off: base + off + nbytes].view(dt).view(B, D, D)


SExx = view(0)
SExx.copy_(SExx0)
del SExx0
lam = torch.empty(B, dtype=dt, device="cuda"); mu = torch.empty(B, D, dtype=dt, device="cuda")
nu = torch.empty(B, dtype=dt, device="cuda"); logdet = torch.empty(B, dtype=dt, device="cuda")
W = q.invU
lam0 = q.lambda_mu_0.expand(B).contiguous(); mu0 = q.mu_0.expand(B, D).contiguous()
invU0 = W.invU_0; nu0 = W.nu_0.expand(B).contiguous()
i0 = invU0.expand(B, D, D)
s_i0 = 0 if i0.stride(0) == 0 else D * D
i0c = i0[0].contiguous() if s_i0 == 0 else i0.contiguous()


def run(d1, d2, reps=6):
    invU, U = view(nbytes + GiB // 2 + d1), view(2 * nbytes + GiB + d2)
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(L.ptr(SExx), D * D, L.ptr(SEx), D, L.ptr(N), 1, L.ptr(lam0), 1, L.ptr(mu0), D, L.ptr(i0c), s_i0, L.ptr(nu0), 1,
                None, 0, None, 0, None, 0, None, 0, ctypes.c_double(1.0), L.ptr(lam), L.ptr(mu), L.ptr(invU), L.ptr(nu), L.ptr(U),
                L.ptr(logdet), B, D, 0, None, L.stream_ptr(torch.device("cuda", 0)))
        e1.record()
        assert rc == 0, rc
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


run(0, 0)
print("outputs at SExx + 2 GiB-ish + d1 / + 4 GiB-ish + d2 (bytes)")
for d1, d2 in ((0, 0), (256, 512), (4096, 8192), (65536, 131072), (1 << 20, 2 << 20), (3 << 20, 5 << 20), (16 << 20, 32 << 20), (64 << 20, 128 << 20),
               (100 << 20, 200 << 20), (256 << 20, 512 << 20), (300 << 20, 700 << 20), (-nbytes % (1 << 21), (-2 * nbytes) % (1 << 21))):
    print(f"d1 = {d1:>10d}  d2 = {d2:>10d}: median {run(d1, d2):.4f} ms", flush=True)
