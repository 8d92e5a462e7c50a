#!/bin/bash
set -u
rm -f gpurun_out/r1_big.txt
for nb in 300 0 30; do
  echo "== GSR_R1_BIG=$nb" >> gpurun_out/r1_big.txt
  GSR_R1_BIG=$nb bash tools/gpu_timeline.sh median --no-extra-legs --config R1 >> gpurun_out/r1_big.txt 2>&1
done
cat gpurun_out/r1_big.txt
bash tools/gpu_r5_emulate.sh > gpurun_out/emul_summary.txt 2>&1
cat gpurun_out/emul_summary.txt
