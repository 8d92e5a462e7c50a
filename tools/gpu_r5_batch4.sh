#!/bin/bash
set -u
rm -f gpurun_out/ab_bin2.txt gpurun_out/ab_rs.txt
for cfg in "--config R1" "--cull 0"; do
  bash tools/gpu_ab_kernel.sh "k_bin_count|k_bin_place" "$cfg" orig nomany >> gpurun_out/ab_bin2.txt 2>&1
done
for cfg in "--config T1" "--config S1" "--config R1" "--config C3 --cull 0" "--cull 0"; do
  bash tools/gpu_ab_kernel.sh "k_radix_hist|k_radix_scatter|k_scan_rows" "$cfg" orig rs8 rs4 >> gpurun_out/ab_rs.txt 2>&1
done
cat gpurun_out/ab_bin2.txt gpurun_out/ab_rs.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_sanitizers.py -m gpu -x -q -k "super_tile_lists or baseline_config or live_policy or framebuffers_beyond or largest_framebuffer or randomised_exactness or tsan or shard" 2>&1 | tail -4
