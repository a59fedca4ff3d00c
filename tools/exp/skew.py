"""K2 at B=1e6, D=16, fp64: does the kernel time depend on the relative placement of the three 2 GB streams
(SExx in, invU out, U out)?  DRAM row-buffer conflicts would show as a dependence on the skew between them."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyvbmp_amd import _lib
from bench import make_inputs
lib = _lib.load()
B, D = 1_000_000, 16
dev = "cuda"
dt = torch.float64
SExx0, SEx, N = make_inputs(B, D, dt, dev)
pool = torch.empty(3 * B * D * D + (64 << 20), dtype=dt, device=dev)  # one arena: placements are explicit offsets
lam0 = torch.ones(1, dtype=dt, device=dev); mu0 = torch.zeros(D, dtype=dt, device=dev)
invU0 = torch.eye(D, dtype=dt, device=dev).contiguous(); nu0 = torch.full((1,), D + 2.0, dtype=dt, device=dev)
lam = torch.empty(B, dtype=dt, device=dev); mu = torch.empty(B, D, dtype=dt, device=dev)
nu = torch.empty(B, dtype=dt, device=dev); ld = torch.empty(B, dtype=dt, device=dev)
fn = lib.vbmp_niw_ss_update_f64
P = lambda t: ctypes.c_void_p(t.data_ptr())
n = B * D * D


def run(sk1, sk2, reps=15):
    """streams at element offsets 0, n + sk1, 2n + sk1 + sk2 of the arena"""
    a = pool[0:n].view(B, D, D); a.copy_(SExx0)
    b = pool[n + sk1:2 * n + sk1].view(B, D, D)
    c = pool[2 * n + sk1 + sk2:3 * n + sk1 + sk2].view(B, D, D)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    def go():
        rc = fn(P(a), D * D, P(SEx), D, P(N), 1, P(lam0), 0, P(mu0), 0, P(invU0), 0, P(nu0), 0,
                None, 0, None, 0, None, 0, None, 0, ctypes.c_double(1.0), P(lam), P(mu), P(b), P(nu), P(c), P(ld),
                B, D, 0, None, st)
        assert rc == 0
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): go()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for sk_bytes in (0, 256, 1024, 4096, 16384, 65536, 1 << 20, (1 << 20) + 4096, 3 << 20, (5 << 20) + 8192):
    sk = sk_bytes // 8
    t = run(sk, sk)
    print(f"skew {sk_bytes:>9d} B between consecutive streams: {t:.4f} ms -> {6432 * B / t / 1e6 / 80:.1f}% of 8 TB/s", flush=True)
