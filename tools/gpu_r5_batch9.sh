#!/bin/bash
set -u
rm -f gpurun_out/ab_mid.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "depth_order or radix_sort_pairs or randomised_records or baseline_config or randomised_exactness or front_slab" 2>&1 | tail -4 > gpurun_out/tests_mid.txt
cat gpurun_out/tests_mid.txt
for cfg in "--config S1" "--config R1" "--config T1" "--config C3 --cull 0" "--config C2 --cull 0"; do
  echo "## $cfg" >> gpurun_out/ab_mid.txt
  for w in 1 0 1 0; do
    GSR_MID_SORT=$w timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 150 --warmup 15 $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mid=$w fps %.1f ms %.4f ok %s' % (d['value'] or -1, d['ms_per_step'], d.get('timed_frame_bit_identical')))" >> gpurun_out/ab_mid.txt
  done
done
cat gpurun_out/ab_mid.txt
