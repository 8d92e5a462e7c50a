#!/bin/bash
# Usage (GPU box): bash tools/gpu_kstats.sh <config> [bench args]: per-kernel average durations of one bench run (rocprofv3 --kernel-trace --stats)
CFG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ks && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --no-cpu-baseline --config $CFG --steps 100 --warmup 10 "$@" > /tmp/ks.log 2>&1
python - <<'PY'
import csv, glob
import os
f = max(glob.glob('/tmp/ks/**/*kernel_stats.csv', recursive=True), key=os.path.getsize)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:22]:
    print("%-60s calls %5s avg %8.1f us total %8.2f ms" % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
