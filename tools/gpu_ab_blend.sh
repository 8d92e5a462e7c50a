#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu_ab_blend.sh "<bench args>" name1 name2 ...: fps + k_blend ms per variant library
ARGS=$1; shift
P=houdini-gsplat-renderer_amd
L=$P/libgsplat_hip.so
cp $L /tmp/orig.so
for v in "$@"; do
  if [ $v = orig ]; then cp /tmp/orig.so $L; else cp $P/variants/libgsplat_hip_$v.so $L; fi
  python bench.py --no-cpu-baseline --no-extra-legs $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-12s %-6s fps %7.1f  blend %.3f ms' % ('$v', d['config']['workload'][:2], d['value'], d['roofline']['avg_launch_ms']))"
done
cp /tmp/orig.so $L
