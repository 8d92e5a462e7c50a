#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu_ab.sh "<bench args>" name1 name2 ...
# Runs bench.py once per variant library built by tools/build_variants.sh (interleaved twice to
# average out box drift) and prints fps + per-stage ms.  "orig" = the in-tree library.
ARGS=$1; shift
P=houdini-gsplat-renderer_amd
L=$P/libgsplat_hip.so
cp $L /tmp/orig.so
for rep in 1 2; do
for v in "$@"; do
  if [ $v = orig ]; then cp /tmp/orig.so $L; else cp $P/variants/libgsplat_hip_$v.so $L; fi
  python bench.py --no-cpu-baseline $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d.get('stages_ms_last_frame',{})
print('%-12s fps %7.1f  blend %.3f ' % ('$v', d['value'], d['roofline']['avg_launch_ms']) + ' '.join('%s %.3f' % (k[3:],v) for k,v in s.items() if k != 'note'))"
done
done
cp /tmp/orig.so $L
