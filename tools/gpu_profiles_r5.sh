#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu_profiles_r5.sh   -- the three regimes of round 5, kernel stats + timeline + PMC each
set -u
for spec in "culled:" "unculled:--cull 0" "slab:--cull 3"; do
  tag=r5_${spec%%:*}; args=${spec#*:}
  bash tools/gpu_profile.sh $tag $args > gpurun_out/prof_$tag.log 2>&1
  bash tools/gpu_pmc.sh $tag $args > gpurun_out/pmc_$tag.log 2>&1
done
python tools/merge_traffic.py gpurun_out/pmc_r5_culled gpurun_out/pmc_r5_unculled gpurun_out/pmc_r5_slab
mkdir -p gpurun_out/profiles_pmc && cp profiles/pmc_traffic.json gpurun_out/profiles_pmc/pmc_traffic.json
for t in culled unculled slab; do tail -3 gpurun_out/prof_r5_$t/frame_timeline.txt; done
# ... and the capture-shaped scene (one-pass regime), kernel stats + timeline only
bash tools/gpu_profile.sh r5_r1 --config R1 > gpurun_out/prof_r5_r1.log 2>&1
tail -3 gpurun_out/prof_r5_r1/frame_timeline.txt
