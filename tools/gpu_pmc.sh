#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu_pmc.sh <tag> [bench args...]
# PMC counters per kernel, one rocprofv3 pass per counter group (never combined with tracing
# domains other than --kernel-trace).  bench.py runs with --no-extra-legs: ONE regime per run -- collect the culled
# (default) and the unculled (--cull 0) regime with two tags and merge them with tools/merge_traffic.py.  Summaries -> gpurun_out/pmc_<tag>/summary.txt
set -u
TAG=${1:-r1}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for GROUP in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
  "GRBM_GUI_ACTIVE FETCH_SIZE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" ; do
  i=$((i+1))
  rm -rf /tmp/pmc_${TAG}_$i
  rocprofv3 --kernel-trace --pmc $GROUP --output-format csv -d /tmp/pmc_${TAG}_$i -o pmc -- \
      python "$REPO/bench.py" --no-cpu-baseline --no-extra-legs "$@" > "$OUT/pass$i.log" 2>&1
  f=$(find /tmp/pmc_${TAG}_$i -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then cp "$f" "$OUT/pass$i.csv"; else echo "pass $i produced no counter csv" ; tail -5 "$OUT/pass$i.log"; fi
done
cd "$REPO"
python tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
rm -f "$OUT"/pass*.csv    # raw per-dispatch dumps are large; the summary is what is kept
