#!/bin/bash
# binning: adaptive big-splat threshold A/B (fps; k_bin_count / k_bin_place durations) on R1, C4 unculled, T1, C4
set -u
for cfg in "--config R1" "--cull 0" "--config T1" ""; do
  bash tools/gpu_ab_kernel.sh "k_bin_count|k_bin_place" "$cfg" orig nomany many3 bm16 >> gpurun_out/ab_binmany.txt 2>&1
done
cat gpurun_out/ab_binmany.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "super_tile_lists or baseline_config or headline or randomised_exactness" 2>&1 | tail -3
