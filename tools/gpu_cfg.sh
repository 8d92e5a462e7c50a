#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu_cfg.sh "<label>" <bench args>: one line with fps + culling counters of a bench run
L=$1; shift
python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); o=d['occlusion_culling']; s=d.get('stages_ms_last_frame',{})
print('%-22s fps %7.1f culled %3d repaired %2d of %3d bits %d dil %d hold %d  without %7.1f  clusters kept %6d/%6d visible %7d | ' % ('$L', d['value'], o['frames_culled'], o['frames_repaired'], o['frames'], o['policy_bits'], o['dilate_tiles'], o['holdoff_frames'], (o['without'] or {}).get('value', 0), d['cluster_culling']['kept_last_frame'], d['cluster_culling']['clusters'], d['n_visible']) + ' '.join('%s %.3f' % (k[3:],v) for k,v in s.items() if k != 'note'))"
