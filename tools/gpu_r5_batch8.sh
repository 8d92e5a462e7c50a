#!/bin/bash
set -u
rm -f gpurun_out/ab_wide.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "depth_order or radix_sort_pairs or randomised_records or storage_order or baseline_config_c3" 2>&1 | tail -4 > gpurun_out/tests_wide.txt
cat gpurun_out/tests_wide.txt
for cfg in "--config R1" "--config T1" "--cull 0"; do
  echo "## $cfg" >> gpurun_out/ab_wide.txt
  for w in 1 0 1 0; do
    GSR_WIDE_DIGITS=$w timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 100 --warmup 10 $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wide=$w fps %.1f ms %.4f ok %s' % (d['value'] or -1, d['ms_per_step'], d.get('timed_frame_bit_identical')))" >> gpurun_out/ab_wide.txt
  done
done
cat gpurun_out/ab_wide.txt
bash tools/gpu_timeline.sh median --no-extra-legs --config R1 > gpurun_out/tl_r1_b.txt 2>&1; cat gpurun_out/tl_r1_b.txt
