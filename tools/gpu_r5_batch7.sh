#!/bin/bash
set -u
rm -f gpurun_out/ab_lb.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "depth_order or radix_sort_pairs or super_tile_lists or baseline_config or randomised_records or randomised_exactness or front_slab or shard or framebuffers_beyond or adversarial" 2>&1 | tail -5 > gpurun_out/tests_lb.txt
cat gpurun_out/tests_lb.txt
for cfg in "--cull 0" "--config T1" "--config S1" "--config R1" "--config C3 --cull 0"; do
  echo "## $cfg" >> gpurun_out/ab_lb.txt
  for lb in 1 0 1 0; do
    GSR_LB_SORT=$lb timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 100 --warmup 10 $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lb=$lb fps %.1f ms %.4f ok %s resorted?' % (d['value'] or -1, d['ms_per_step'], d.get('timed_frame_bit_identical')))" >> gpurun_out/ab_lb.txt
  done
  bash tools/gpu_ab_blend.sh "$cfg" prev2 >> gpurun_out/ab_lb.txt 2>&1
done
cat gpurun_out/ab_lb.txt
for cfg in "--cull 0" "--config R1"; do GSR_LB_SORT=1 bash tools/gpu_timeline.sh median --no-extra-legs $cfg; done > gpurun_out/tl_lb.txt 2>&1
cat gpurun_out/tl_lb.txt
