#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the per-regime PMC runs of tools/gpu_pmc.sh:
    python tools/merge_traffic.py gpurun_out/pmc_r4_culled gpurun_out/pmc_r4_unculled [gpurun_out/pmc_r4_slab]
-> {"culled": {kernel: {...}}, "unculled": {...}, "slab": {...}, "_kernel_source_sha": ..., "_collected": ...}; bench.py replays the entry of
the regime its timed region ran in, and only while the kernel sources are the ones the counters were collected with."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
shas = set()
for regime, d in zip(("culled", "unculled", "slab"), sys.argv[1:4]):
    t = json.load(open(os.path.join(d, "pmc_traffic.json")))
    shas.add(t.pop("_kernel_source_sha", None))
    t.pop("_collected", None)
    out[regime] = {k: v for k, v in t.items() if k.startswith("k_") or "k_" in k}
if len(shas) != 1:
    sys.exit(f"the runs were collected with different kernel sources: {shas}")
out["_kernel_source_sha"] = shas.pop()
out["_collected"] = "rocprofv3 --kernel-trace --pmc passes of `python bench.py --no-cpu-baseline --no-extra-legs` (culled), `... --cull 0` (unculled) and `... --cull 3` (slab: every frame a front-slab frame): tools/gpu_pmc.sh + tools/merge_traffic.py"
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print("wrote profiles/pmc_traffic.json:", {k: len(v) for k, v in out.items() if isinstance(v, dict)})
