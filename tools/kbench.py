#!/usr/bin/env python3
"""Kernel micro-benchmarks (GPU box): average/min duration of single kernel launches measured with HIP
events at the launch, plus the implied fraction of the 8 TB/s HBM peak.  Development aid.

    python tools/kbench.py niw [--B 1000000] [--reps 30]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import algorithmic_bytes_per_update, make_inputs  # noqa: E402
from pyvbmp_amd import _lib, ops  # noqa: E402


def timed(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    ev = []
    _lib.launch_hooks = (lambda n: ev.append(_rec()), lambda n: ev.append(_rec()))
    for _ in range(reps):
        fn()
    _lib.launch_hooks = None
    torch.cuda.synchronize()
    ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(0, len(ev), 2)]
    per = len(ts) // reps
    tot = [sum(ts[i * per:(i + 1) * per]) for i in range(reps)]
    tot.sort()
    return sum(tot) / len(tot), tot[0], tot[len(tot) // 2]


def _rec():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def bench_niw(args):
    for dt, D, lr in [(torch.float64, 16, 1.0), (torch.float32, 16, 1.0), (torch.float64, 16, 0.5),
                      (torch.float64, 8, 1.0), (torch.float64, 32, 1.0), (torch.float32, 32, 1.0),
                      (torch.float64, 6, 1.0), (torch.float64, 2, 1.0), (torch.float64, 64, 1.0)]:
        B = args.B if D <= 16 else args.B // (D * D // 256)
        SExx, SEx, N = make_inputs(B, D, dt, "cuda")
        from pyvbmp_amd.dists import NormalInverseWishart
        q = NormalInverseWishart((D,), (B,), device="cuda", dtype=dt)
        q.ss_update(SExx, SEx, N, lr=1.0, beta=None)
        avg, mn, med = timed(lambda: q.ss_update(SExx, SEx, N, lr=lr, beta=None), args.reps)
        bpu = algorithmic_bytes_per_update(D, SExx.element_size(), lr)
        print(f"niw_ss_update {str(dt)[6:]:8s} D={D:3d} B={B:8d} lr={lr}: avg {avg:.4f} ms  min {mn:.4f}  med {med:.4f}"
              f"  -> {bpu * B / med / 1e6:8.1f} GB/s ({bpu * B / med / 1e6 / 80:.1f}% of 8 TB/s)", flush=True)
        del q, SExx, SEx, N
        torch.cuda.empty_cache()


def bench_copy(args):
    n = 6432 * args.B // 16
    a = torch.empty(n, 2, dtype=torch.float64, device="cuda").normal_()
    b = torch.empty_like(a)
    st, en = _rec(), None
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.reps):
        s = _rec()
        b.copy_(a)
        e = _rec()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    print(f"torch copy of {a.numel() * 8 / 1e9:.2f} GB: med {ts[len(ts) // 2]:.4f} ms -> "
          f"{2 * a.numel() * 8 / ts[len(ts) // 2] / 1e6:.1f} GB/s (read+write)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["niw"])
    ap.add_argument("--B", type=int, default=1_000_000)
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    for w in args.what:
        {"niw": bench_niw, "copy": bench_copy}[w](args)
