#!/bin/bash
set -u
rm -f gpurun_out/ab_fuse.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile_order or baseline_config or options or randomised_exactness or headline or front_slab or occlusion" 2>&1 | tail -4 > gpurun_out/tests_fuse.txt
cat gpurun_out/tests_fuse.txt
for cfg in "--config C3" "--config T1" "--config S1" "--config C5" ""; do
  echo "## $cfg" >> gpurun_out/ab_fuse.txt
  for w in 1 0 1 0; do
    GSR_FUSE_ORDER=$w timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 200 --warmup 20 $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse=$w fps %.1f ms %.4f ok %s' % (d['value'] or -1, d['ms_per_step'], d.get('timed_frame_bit_identical')))" >> gpurun_out/ab_fuse.txt
  done
done
cat gpurun_out/ab_fuse.txt
bash tools/gpu_timeline.sh median --no-extra-legs --config C3 | tail -6
