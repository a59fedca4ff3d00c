#!/usr/bin/env python3
"""Headline benchmark: conjugate updates/s of batched NormalInverseWishart.ss_update
(BASELINE.json config 2: batch = 1e6 per GPU, D = 16, fp64, lr = 1, default priors).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--dim D] [--dtype f64|f32] [--lr LR]

One "step" = one pass of the hot path over the whole synthetic batch through the product class
(pyvbmp_amd.dists.NormalInverseWishart.ss_update -> libvbmp_hip.so vbmp_niw_ss_update_*), inputs
already resident in HBM.  N > 1 is launched by torch.distributed.run (one rank per GPU); the batch
axis is sharded (independent posteriors, no data-path collective) => weak scaling.
Rank 0 prints ONE JSON line (metric/value/... + roofline + cpu_baseline).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling


def algorithmic_bytes_per_update(D, itemsize, lr):
    """reads SExx D^2 + SEx D + N 1; writes invU D^2 + U D^2 + mu D + lambda, nu, logdet.
    lr < 1 additionally reads the old state (invU D^2 + mu D + lambda + nu)."""
    elems = 3 * D * D + 2 * D + 4
    if lr != 1.0:
        elems += D * D + D + 2
    return elems * itemsize


def make_inputs(B, D, dtype, device, n=32, seed=0):
    """torch.manual_seed(0); A ~ N(0,1) (B,D,n) in chunks; SExx = A A^T, SEx = A.sum(-1), N = n."""
    g = torch.Generator(device=device).manual_seed(seed)
    SExx = torch.empty(B, D, D, dtype=dtype, device=device)
    SEx = torch.empty(B, D, dtype=dtype, device=device)
    chunk = 131072
    for s in range(0, B, chunk):
        e = min(B, s + chunk)
        A = torch.randn(e - s, D, n, generator=g, dtype=dtype, device=device)
        SExx[s:e] = A @ A.transpose(-2, -1)
        SEx[s:e] = A.sum(-1)
    N = torch.full((B,), float(n), dtype=dtype, device=device)
    return SExx, SEx, N


def cpu_baseline(D, dtype, lr, target_s=12.0):
    """The CPU oracle (torch-CPU restatement of the reference's op sequence, pinned to the reference by
    tests/golden) timed on this host's cores on a bounded sample of the same workload."""
    from oracle import niw as oniw
    threads = torch.get_num_threads()
    Bc = 20_000
    SExx, SEx, N = make_inputs(Bc, D, dtype, "cpu")
    st = oniw.niw_new((D,), (Bc,), dtype=dtype)
    t0 = time.perf_counter()
    oniw.niw_ss_update(st, SExx, SEx, N, lr=lr, beta=None)
    t_small = time.perf_counter() - t0
    # scale the sample so that the timed part is roughly target_s, capped for memory (~17 GB RSS at 1e6)
    Bc = int(min(500_000, max(20_000, Bc * target_s / 2.0 / max(t_small, 1e-3))))
    SExx, SEx, N = make_inputs(Bc, D, dtype, "cpu")
    st = oniw.niw_new((D,), (Bc,), dtype=dtype)
    best = float("inf")
    for _ in range(2):
        t0 = time.perf_counter()
        oniw.niw_ss_update(st, SExx, SEx, N, lr=lr, beta=None)
        best = min(best, time.perf_counter() - t0)
    return {"value": Bc / best, "unit": "updates/s", "cores": threads, "kind": "port",
            "sample": f"oracle.niw.niw_ss_update on B={Bc} of the same synthetic batch (D={D}, {str(dtype)[6:]}, "
                      f"lr={lr}), best of 2, {best:.2f} s; host has {os.cpu_count()} logical cpus"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1_000_000, help="batch per GPU")
    ap.add_argument("--dim", type=int, default=16)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--lr", type=float, default=1.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    from pyvbmp_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    from pyvbmp_amd.dists import NormalInverseWishart

    dtype = torch.float64 if args.dtype == "f64" else torch.float32
    B, D = args.batch, args.dim
    SExx, SEx, N = make_inputs(B, D, dtype, device, seed=rank)
    q = NormalInverseWishart((D,), (B,), device=device, dtype=dtype)

    def step():
        q.ss_update(SExx, SEx, N, lr=args.lr, beta=None)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # HIP events recorded by the launch hooks right around the kernel enqueue, on torch's current
    # stream (= the stream handed to the C-ABI), so each pair brackets exactly one kernel launch.
    starts, stops = [], []

    def before(name):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        starts.append(ev)

    def after(name):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        stops.append(ev)

    barrier()
    _lib.launch_hooks = (before, after)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    _lib.launch_hooks = None
    assert len(starts) == args.steps == len(stops), "one kernel launch per step expected"
    kernel_ms = sum(s.elapsed_time(e) for s, e in zip(starts, stops)) / args.steps
    if dist is not None:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])

    if rank == 0:
        itemsize = 8 if dtype == torch.float64 else 4
        bpu = algorithmic_bytes_per_update(D, itemsize, args.lr)
        achieved = bpu * B / (kernel_ms * 1e-3) / 1e9  # GB/s, one launch = B updates
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = f"niw_ss_update_{args.dtype}_D{D}_B{B}_lr{args.lr}"
                traffic = tj.get(key)
            except Exception:
                traffic = None
        out = {
            "metric": "conjugate updates/sec (batch=1e6, D=16 NIW)",
            "value": world * B * args.steps / elapsed,
            "unit": "updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: batched NormalInverseWishart.ss_update, batch={B} per GPU, "
                                   f"D={D}, {args.dtype}, lr={args.lr}, beta=None, default priors",
                       "batch_per_gpu": B, "dim": D, "lr": args.lr, "parallelism": f"batch-sharded x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_niw_ss_update", "kernel_ms": kernel_ms, "bytes_per_update": bpu},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(D, dtype, args.lr)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
