#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu_profiles_r6.sh   -- round 6: the three regimes (kernel stats + timeline + PMC each), the
# capture-shaped scene, and the two depth-tested frames the viewport hook issues (a cleared depth buffer; an opaque sphere among the splats)
set -u
for spec in "culled:" "unculled:--cull 0" "slab:--cull 3"; do
  tag=r6_${spec%%:*}; args=${spec#*:}
  bash tools/gpu_profile.sh $tag $args > gpurun_out/prof_$tag.log 2>&1
  bash tools/gpu_pmc.sh $tag $args > gpurun_out/pmc_$tag.log 2>&1
done
python tools/merge_traffic.py gpurun_out/pmc_r6_culled gpurun_out/pmc_r6_unculled gpurun_out/pmc_r6_slab
mkdir -p gpurun_out/profiles_pmc && cp profiles/pmc_traffic.json gpurun_out/profiles_pmc/pmc_traffic.json
for t in culled unculled slab; do tail -3 gpurun_out/prof_r6_$t/frame_timeline.txt; done
bash tools/gpu_profile.sh r6_r1 --config R1 > gpurun_out/prof_r6_r1.log 2>&1
tail -3 gpurun_out/prof_r6_r1/frame_timeline.txt
bash tools/gpu_profile.sh r6_depth_far --depth far > gpurun_out/prof_r6_depth_far.log 2>&1
bash tools/gpu_profile.sh r6_depth_occluder --depth occluder > gpurun_out/prof_r6_depth_occluder.log 2>&1
bash tools/gpu_pmc.sh r6_depth_occluder --depth occluder > gpurun_out/pmc_r6_depth_occluder.log 2>&1
tail -3 gpurun_out/prof_r6_depth_far/frame_timeline.txt gpurun_out/prof_r6_depth_occluder/frame_timeline.txt
