#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu_r5_timelines.sh   -- one-pass frame timelines of C4 (--cull 0), T1 and S1 -> gpurun_out/tl_<tag>.txt
set -u
mkdir -p gpurun_out
for spec in "c4:--config C4 --cull 0" "t1:--config T1" "s1:--config S1" "c3:--config C3 --cull 0"; do
  tag=${spec%%:*}; args=${spec#*:}
  bash tools/gpu_timeline.sh median --no-extra-legs $args > gpurun_out/tl_$tag.txt 2>&1
  tail -3 /tmp/tl.log > gpurun_out/tl_$tag.log 2>&1
done
tail -n 30 gpurun_out/tl_*.txt
