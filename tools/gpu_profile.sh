#!/bin/bash
# Usage (on the GPU box, from the repo root): bash tools/gpu_profile.sh <tag> [bench args...]
# Runs bench.py un-profiled, then (with --no-extra-legs: one regime per run) under rocprofv3 --kernel-trace --stats, and leaves the
# summaries in gpurun_out/prof_<tag>/ (copy what should be judged into profiles/).
set -u
TAG=${1:-r1}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 3000 "$OUT/bench.json"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$TAG -o trace -- python "$REPO/bench.py" --no-cpu-baseline --no-extra-legs "$@" > "$OUT/bench_profiled.json" 2> "$OUT/rocprof.err"
cd "$REPO"
cp "$(find /tmp/rp_$TAG -name '*kernel_stats*' -printf '%s %p\n' | sort -n | tail -1 | cut -d' ' -f2)" "$OUT/kernel_stats.csv"
find /tmp/rp_$TAG -name '*domain_stats*' -exec cp {} "$OUT/domain_stats.csv" \;
T=$(find /tmp/rp_$TAG -name '*kernel_trace.csv' -printf '%s %p\n' | sort -n | tail -1 | cut -d' ' -f2)   # (the largest: bench.py's own process, not the HBM micro-benchmark it spawns)
[ -n "$T" ] && python tools/frame_timeline.py "$T" median > "$OUT/frame_timeline.txt" 2>&1   # (the frame of median period among the last 40)
[ -n "$T" ] && python tools/frame_timeline.py "$T" --gaps > "$OUT/launch_gaps.txt" 2>&1   # (idle time in front of every kernel, over the whole trace)
ls -la /tmp/rp_$TAG/* | head
head -30 "$OUT/kernel_stats.csv"
