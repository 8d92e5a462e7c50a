#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu_r5_emulate.sh   -- per-rank frames of an N-GPU run on ONE GPU (bench.py --emulate-shard N --emulate-rank R:
# rank R's band only, no gather): the heaviest and the lightest band of BASELINE C5 (4K) at 8 / 4 / 2 ranks, both layouts, and C4's at 8
set -u
mkdir -p gpurun_out/emul
for spec in "C5 8 0 1" "C5 8 3 1" "C5 8 4 1" "C5 8 7 1" "C5 8 0 0" "C5 8 4 0" "C5 4 0 1" "C5 4 1 1" "C5 2 0 1" "C5 2 1 1" "C4 8 0 1" "C4 8 4 1" "C4 4 1 1" "C4 2 0 1"; do
  set -- $spec
  python bench.py --config $1 --emulate-shard $2 --emulate-rank $3 --shard-layout $4 --no-cpu-baseline --no-extra-legs --steps 100 --warmup 10 \
      > gpurun_out/emul/bench_$(echo $1 | tr A-Z a-z)_rank$3of$2_layout$4_emulated.json 2>/dev/null
  python - "$spec" gpurun_out/emul/bench_$(echo $1 | tr A-Z a-z)_rank$3of$2_layout$4_emulated.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-12s ms_per_step %.4f  fps %.0f  blend %.4f ms  visible %d" % (sys.argv[1], d["ms_per_step"], d["value"], d["roofline"]["avg_launch_ms"], d["n_visible"]))
PY
done
