#!/bin/bash
set -u
rm -f gpurun_out/ab_bin3.txt gpurun_out/ab_rs.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "super_tile_lists or baseline_config or framebuffers_beyond or largest_framebuffer or randomised_exactness or shard or front_slab or one_rank_through" 2>&1 | tail -4 > gpurun_out/tests_bin3.txt
cat gpurun_out/tests_bin3.txt
for cfg in "--config R1" "--cull 0" "--config T1" "" "--config C2" "--config C5"; do
  bash tools/gpu_ab_kernel.sh "k_bin_count|k_bin_place" "$cfg" orig prevbin orig prevbin >> gpurun_out/ab_bin3.txt 2>&1
done
cat gpurun_out/ab_bin3.txt
for cfg in "--config T1" "--config S1" "--config C3 --cull 0" "--cull 0"; do
  bash tools/gpu_ab_kernel.sh "k_radix_hist|k_radix_scatter|k_scan_rows" "$cfg" orig rs8 rs4 >> gpurun_out/ab_rs.txt 2>&1
done
cut -c1-260 gpurun_out/ab_rs.txt
