#!/usr/bin/env python3
"""HBM traffic of the dominant kernel of every bench workload, measured with rocprofv3 PMC counters the way
MI355X_MICROARCH.md (section HBM) prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), each pass
`rocprofv3 --kernel-trace --pmc <counter> -- python3 bench.py ...` (the program itself after `--`), per-dispatch median
over the kernel's launches, FETCH_SIZE doubled (gfx950 tallies the 128-byte requests of wide coalesced reads at 64 bytes).
Writes gpurun_out/pmc_traffic.json; copy it to profiles/pmc_traffic.json (bench.py reports `roofline.traffic` from there,
with this provenance string).

    VBMP_COMMIT=$(git rev-parse --short HEAD) python3 tools/pmc_traffic.py        # on the GPU box
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
CONFIGS = [
    ["--workload", "niw"],
    ["--workload", "niw", "--dtype", "f32"],
    ["--workload", "niw", "--lr", "0.5"],
    ["--workload", "niw", "--dim", "32", "--batch", "250000"],
    ["--workload", "niw", "--dim", "8"],
    ["--workload", "mnw_fwd"],
    ["--workload", "mnw_bwd"],
    ["--workload", "lds"],
    ["--workload", "lds", "--dtype", "f32"],
    ["--workload", "dmbd"],
]


def one_pass(counter, args, tag):
    d = os.path.join(OUT, f"pmc_{tag}_{counter}")
    shutil.rmtree(d, ignore_errors=True)
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--",
           "python3", os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline"] + args
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    line = next((ln for ln in r.stdout.splitlines() if ln.startswith("{")), None)
    if r.returncode != 0 or line is None:
        raise RuntimeError(f"{' '.join(cmd)} failed:\n{r.stderr[-2000:]}")
    bench = json.loads(line)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    vals = []
    for f in files:
        for row in csv.DictReader(open(f)):
            if bench["roofline"]["kernel"] in row["Kernel_Name"] and row["Counter_Name"] == counter:
                vals.append(float(row["Counter_Value"]))
    vals.sort()
    shutil.rmtree(d, ignore_errors=True)
    if not vals:
        raise RuntimeError(f"no {counter} rows for kernel {bench['roofline']['kernel']}")
    return bench, vals[len(vals) // 2], len(vals)


def main():
    os.makedirs(OUT, exist_ok=True)
    commit = os.environ.get("VBMP_COMMIT", "unknown")
    result = {}
    for i, args in enumerate(CONFIGS):
        bench, fetch_kib, n1 = one_pass("FETCH_SIZE", args, i)
        _, write_kib, n2 = one_pass("WRITE_SIZE", args, i)
        key = bench["roofline"]["traffic_key"]
        total = fetch_kib * 1024.0 * 2.0 + write_kib * 1024.0
        alg = bench["roofline"]["achieved"] * 1e9 * bench["roofline"]["kernel_ms"] * 1e-3
        result[key] = {
            "bytes": total, "fetch_kib_raw": fetch_kib, "write_kib": write_kib, "dispatches": [n1, n2],
            "algorithmic_bytes": alg, "ratio": total / alg,
            "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, per-dispatch median, FETCH x2 on gfx950), "
                      f"tools/pmc_traffic.py at commit {commit}, {time.strftime('%Y-%m-%d')}",
        }
        print(f"{key}: {total:.4e} B per launch, algorithmic {alg:.4e}, ratio {total / alg:.3f}", flush=True)
        json.dump(result, open(os.path.join(OUT, "pmc_traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
