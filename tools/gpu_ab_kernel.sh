#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu_ab_kernel.sh "<kernel name substrings, |-separated>" "<bench args>" name1 name2 ...
# fps + average duration of the named kernels (rocprofv3 kernel trace of the timed region) per variant library
PAT=$1; ARGS=$2; shift; shift
P=houdini-gsplat-renderer_amd
L=$P/libgsplat_hip.so
R=$GRAFT_REPO_ROOT
cp $L /tmp/orig.so
for v in "$@"; do
  if [ $v = orig ]; then cp /tmp/orig.so $L; else cp $P/variants/libgsplat_hip_$v.so $L; fi
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/abk && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk -- python $R/bench.py --no-cpu-baseline --no-extra-legs --time-every 1000 $ARGS > /tmp/abk.json 2>/dev/null)
  S=$(find /tmp/abk -name '*kernel_stats.csv' -printf '%s %p\n' | sort -n | tail -1 | cut -d' ' -f2)
  python - "$v" "$PAT" "$S" <<'PY'
import csv, json, sys
v, pat, path = sys.argv[1:4]
d = json.loads(open('/tmp/abk.json').read().strip().splitlines()[-1])
out = []
for r in csv.DictReader(open(path)):
    if any(p in r['Name'] for p in pat.split('|')):
        out.append('%s %.1f us' % (r['Name'].split('(')[0][:24], float(r['AverageNs']) / 1e3))
print('%-10s fps %7.1f | %s' % (v, d['value'], '; '.join(out)))
PY
done
cp /tmp/orig.so $L
