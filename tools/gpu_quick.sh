#!/bin/bash
# Usage (GPU box, repo root): bash tools/gpu_quick.sh [bench args]: fps + cluster / culling counters of a short run, then the kernel timeline of one frame
R=$GRAFT_REPO_ROOT
python $R/bench.py --no-cpu-baseline --no-extra-legs --steps 60 "$@" > /tmp/q.json 2> /tmp/q.err || tail -5 /tmp/q.err
python - <<'PY'
import json
d = json.load(open('/tmp/q.json'))
print("fps %.1f  ms %.4f  visible %d  blend %.4f ms  pairs %d  scan amp %.2f" % (d["value"], d["ms_per_step"], d["n_visible"], d["roofline"]["avg_launch_ms"],
      d["roofline"]["pairs_sorted_last_frame"], d["roofline"]["scan_amplification"] or 0))
print(d.get("cluster_culling"), {k: d["occlusion_culling"][k] for k in ("frames_culled", "frames_repaired", "frames")})
PY
bash $R/tools/gpu_timeline.sh 30 --no-extra-legs "$@"
