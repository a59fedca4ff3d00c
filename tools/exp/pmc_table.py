"""Tabulate rocprofv3 --pmc passes over place_probe's fixed "pmc" dispatch sequence (tools/exp/place_pmc.sh): one row per
configuration (the second launch of each pair), one column per counter, plus the kernel duration from the same pass."""
import csv, glob, os, sys
from collections import OrderedDict, defaultdict
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r3/pmc"
names = None
table = OrderedDict()
for d in sorted(glob.glob(os.path.join(root, "p[0-9]*"))):
    if not os.path.isdir(d):
        continue
    f = glob.glob(os.path.join(d, "*counter_collection.csv"))
    if not f:
        continue
    log = [ln.split("  ")[0][4:].strip() for ln in open(d + ".log") if ln.startswith("pmc ")]
    log = [" ".join(ln.split()[:-2]) for ln in open(d + ".log") if ln.startswith("pmc ")]
    rows = defaultdict(dict)
    order = []
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if not any(t in k for t in ("k_mix", "k_skel", "k_niw_ss_update")):
            continue
        did = int(r["Dispatch_Id"])
        if did not in rows:
            order.append(did)
        rows[did][r["Counter_Name"]] = float(r["Counter_Value"])
        rows[did]["_ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    # two launches per configuration: keep the second
    for i, cfg in enumerate(log):
        did = order[2 * i + 1]
        table.setdefault(cfg, {}).update({k: v for k, v in rows[did].items() if k != "_ms"})
        table[cfg].setdefault("_ms", []).append(rows[did]["_ms"])
cols = sorted({k for v in table.values() for k in v if k != "_ms"})
print("| configuration | kernel ms (per pass) | " + " | ".join(cols) + " |")
print("|---|---|" + "---|" * len(cols))
for cfg, v in table.items():
    print(f"| {cfg} | " + " ".join(f"{m:.3f}" for m in v["_ms"]) + " | " + " | ".join(f"{v.get(c, float('nan')):.4g}" for c in cols) + " |")
