#!/bin/bash
# Usage (GPU box, repo root): bash tools/blend_stats.sh <bench args>: what k_blend did per launch (entries scanned, records gathered, evaluations)
python bench.py --no-cpu-baseline --no-extra-legs "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; v=r['valu']
print('%s fps %.1f blend %.3f ms | pairs in lists %d, entries scanned %.2f M, records gathered %.2f M, wave evals %.2f M | scan amp %.1f | issue bound %.3f ms (%.2f)' % (
  d['config']['workload'][:28], d['value'], r['avg_launch_ms'], r['pairs_sorted_last_frame'], r['entries_scanned_per_launch']/1e6, r['pairs_consumed_per_launch']/1e6,
  v['wave_record_evals_per_launch']/1e6, r['scan_amplification'] or 0, v['inner_loop_issue_bound_ms'], v['inner_loop_issue_frac']))"
