#!/bin/bash
set -u
rm -f gpurun_out/ab_k1lazy.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lazy or records_bit_exact or baseline_config or randomised_exactness or depth_tested or options" 2>&1 | tail -4 > gpurun_out/tests_k1lazy.txt
cat gpurun_out/tests_k1lazy.txt
for cfg in "--cull 0" "--config T1" "--config S1" "--config R1" "--config C5 --cull 0" "--lazy 2"; do
  bash tools/gpu_ab_kernel.sh "k_preprocess" "$cfg" orig prevk1 orig prevk1 >> gpurun_out/ab_k1lazy.txt 2>&1
done
cat gpurun_out/ab_k1lazy.txt | cut -c1-160
