#!/bin/bash
# Usage (GPU box, repo root, after tools/build_tree.sh <ref> in the build container): bash tools/gpu_ab_tree.sh [rounds] -- "<bench args>" ["<bench args>" ...]
# Alternates this tree's bench.py and _tree/'s on the same GPU: fps and the blend kernel's launch time per run.
ROUNDS=2
if [ "$1" != "--" ]; then ROUNDS=$1; shift; fi
shift
one() { (cd "$1" && python bench.py --no-cpu-baseline --no-extra-legs $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-8s %-34s fps %8.1f  blend %.4f ms' % ('$3', '$2'[:34], d['value'], d['roofline']['avg_launch_ms']))"); }
for args in "$@"; do
  for r in $(seq $ROUNDS); do one . "$args" this; one _tree "$args" tree; done
done
