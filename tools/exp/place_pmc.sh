#!/bin/bash
# r03: counter passes over tools/exp/place_probe's fixed "pmc" dispatch sequence (one rocprofv3 run per counter group; the
# microarch guide's rule: --pmc alone with --kernel-trace).  Usage (GPU box, repo root): tools/exp/place_pmc.sh OUTDIR
out=${1:-gpurun_out/r3/pmc}; mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
i=0
while read -r group; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $group -d $out/p$i -o p$i --output-format csv -- tools/exp/place_probe pmc > $out/p$i.log 2>&1 || echo "pass $i failed: $group" >> $out/failed.txt
  echo "$group" > $out/p$i.counters
done <<'GROUPS'
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum
TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum
TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCC_TAG_STALL_sum
GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_SERIALIZATION_STALL_sum
GROUPS
ls -R $out | head -40
