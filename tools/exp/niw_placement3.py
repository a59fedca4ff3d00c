"""K2 headline: kernel time per OUTPUT BLOCK -- several separately allocated (invU, U) pairs, and pairs that are halves of one 4 GB block --
all kept alive so that every candidate is a different piece of memory; the input SExx stays where it is."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import make_inputs
from pyvbmp_amd import _lib as L
from pyvbmp_amd.dists import NormalInverseWishart
B, D, dt = 1_000_000, 16, torch.float64
SExx, SEx, N = make_inputs(B, D, dt, "cuda")
q = NormalInverseWishart((D,), (B,), device="cuda", dtype=dt)
lib = L.load()
fn = lib.vbmp_niw_ss_update_f64
lam = torch.empty(B, dtype=dt, device="cuda"); mu = torch.empty(B, D, dtype=dt, device="cuda")
nu = torch.empty(B, dtype=dt, device="cuda"); logdet = torch.empty(B, dtype=dt, device="cuda")
W = q.invU
lam0 = q.lambda_mu_0.expand(B).contiguous(); mu0 = q.mu_0.expand(B, D).contiguous()
nu0 = W.nu_0.expand(B).contiguous()
i0 = W.invU_0.expand(B, D, D)
s_i0 = 0 if i0.stride(0) == 0 else D * D
i0c = i0[0].contiguous() if s_i0 == 0 else i0.contiguous()
args_fixed = None


def run(invU, U, reps=7):
    pre = (L.ptr(SExx), D * D, L.ptr(SEx), D, L.ptr(N), 1, L.ptr(lam0), 1, L.ptr(mu0), D, L.ptr(i0c), s_i0, L.ptr(nu0), 1,
           None, 0, None, 0, None, 0, None, 0, ctypes.c_double(1.0), L.ptr(lam), L.ptr(mu), L.ptr(invU), L.ptr(nu), L.ptr(U),
           L.ptr(logdet), B, D, 0, None, L.stream_ptr(torch.device("cuda", 0)))
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*pre)
        e1.record()
        assert rc == 0
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


keep = []
for i in range(6):
    a = torch.empty(B, D, D, dtype=dt, device="cuda"); b = torch.empty(B, D, D, dtype=dt, device="cuda")
    keep += [a, b]
    print(f"separate blocks {i}: invU at {a.data_ptr():#x}, U at {b.data_ptr():#x}: median {run(a, b):.4f} ms", flush=True)
for i in range(4):
    ab = torch.empty(2, B, D, D, dtype=dt, device="cuda")
    keep.append(ab)
    print(f"one 4 GB block  {i}: at {ab.data_ptr():#x}: median {run(ab[0], ab[1]):.4f} ms", flush=True)
# mixed: invU from one pair, U from another
print(f"mixed (invU of pair 0, U of pair 3): {run(keep[0], keep[7]):.4f} ms; (invU of pair 2, U of pair 5): {run(keep[4], keep[11]):.4f} ms")

print("invU of pair i (rows) with U of pair j (columns), ms:")
for i in range(6):
    print("  " + " ".join(f"{run(keep[2 * i], keep[2 * j + 1], reps=3):.3f}" for j in range(6)), flush=True)
print("invU of pair i with U = the invU buffer of pair j (columns), ms:")
for i in range(6):
    print("  " + " ".join((f"{run(keep[2 * i], keep[2 * j], reps=3):.3f}" if i != j else "  -  ") for j in range(6)), flush=True)
