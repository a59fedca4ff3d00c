#!/bin/bash
# One pass of the round's measured evidence on a GPU box (run from the repo root through gpurun):
#   bench.py for every workload (one JSON line each), the same command under `rocprofv3 --kernel-trace --stats` (kernel summary
#   CSV + the bench line it printed while profiled), tools/kbench.py.  Everything lands under gpurun_out/evidence/; copy what is
#   to be judged into profiles/ (tracked).
#   usage: tools/collect_evidence.sh [workload ...]      (default: all five)
set -o pipefail
cd "$(dirname "$0")/.."
out=gpurun_out/evidence; mkdir -p $out
export TMPDIR=/tmp
wl=${@:-niw mnw_fwd mnw_bwd lds dmbd}
: > $out/bench_lines.jsonl
[ -n "$SKIP_BENCH" ] || for w in $wl; do
  python3 bench.py --workload $w 2> $out/bench_$w.err | tail -n 1 >> $out/bench_lines.jsonl || exit 1
  echo "bench $w done" >&2
done
for w in $wl; do
  rm -rf /tmp/prof_$w
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o $w -- python3 $OLDPWD/bench.py --workload $w --no-cpu-baseline) > $out/${w}_under_rocprof.out 2> $out/${w}_rocprof.err || { tail -5 $out/${w}_rocprof.err >&2; exit 1; }
  tail -n 1 $out/${w}_under_rocprof.out > $out/${w}_under_rocprof.json
  f=$(find /tmp/prof_$w -name "*kernel_stats.csv" | head -n 1)
  [ -n "$f" ] && cp "$f" $out/${w}_kernel_stats.csv
  echo "rocprof $w done" >&2
done
[ -n "$SKIP_BENCH" ] || python3 tools/kbench.py niw copy k1 mnw gmm gmm0 lds lds0 dmbd mixlt dmix > $out/kbench.txt 2> $out/kbench.err || { tail -5 $out/kbench.err >&2; }
echo "kbench done" >&2
