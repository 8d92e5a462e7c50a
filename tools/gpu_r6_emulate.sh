#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu_r6_emulate.sh   -- per-rank frames of an N-GPU run on ONE GPU (bench.py --emulate-shard N --emulate-rank R:
# rank R's band only, no gather) for the BLEND-BOUND scenes, where row sharding should pay most: R1 (a capture-shaped cloud, the camera inside
# the room: k_blend is 75 % of its frame) and T1 (a landscape under open sky: k_blend is half of the frame and as long as its heaviest tile),
# every band at 2 and 4 ranks, the bands 0 / 2 / 5 / 7 at 8; the single-GPU frame of each first.  Lines are kept in profiles/r6_emulated/.
set -u
mkdir -p gpurun_out/emul6
for spec in "R1 1 0" "R1 2 0" "R1 2 1" "R1 4 0" "R1 4 1" "R1 4 2" "R1 4 3" "R1 8 0" "R1 8 2" "R1 8 5" "R1 8 7" \
            "T1 1 0" "T1 2 0" "T1 2 1" "T1 4 0" "T1 4 1" "T1 4 2" "T1 4 3" "T1 8 0" "T1 8 2" "T1 8 5" "T1 8 7"; do
  set -- $spec
  out=gpurun_out/emul6/bench_$(echo $1 | tr A-Z a-z)_rank$3of$2_emulated.json
  if [ $2 = 1 ]; then sh=""; else sh="--emulate-shard $2 --emulate-rank $3 --shard-layout 1"; fi
  python bench.py --config $1 $sh --no-cpu-baseline --no-extra-legs --steps 100 --warmup 10 > $out 2>/dev/null
  python - "$spec" $out <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-10s ms_per_step %.4f  fps %.0f  blend %.4f ms  visible %d" % (sys.argv[1], d["ms_per_step"], d["value"], d["roofline"]["avg_launch_ms"], d["n_visible"]))
PY
done
