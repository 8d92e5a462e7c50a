#!/bin/bash
# Usage (GPU box): bash tools/gpu_timeline.sh <frame index> [bench args]: timeline of one frame of the TIMED leg of bench.py
IDX=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tl && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $R/bench.py --no-cpu-baseline --steps 60 --warmup 5 "$@" > /tmp/tl.log 2>&1
T=$(find /tmp/tl -name '*kernel_trace.csv' -printf '%s %p\n' | sort -n | tail -1 | cut -d' ' -f2)   # (the largest: bench.py's own process, not the HBM micro-benchmark it spawns)
python $R/tools/frame_timeline.py $T $IDX
