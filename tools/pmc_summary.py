#!/usr/bin/env python3
"""Aggregate rocprofv3 counter_collection CSVs (one per PMC pass) into per-kernel means."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in sorted(glob.glob(os.path.join(out, "pass*.csv"))):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0]
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
names = sorted({c for k in acc for c in acc[k]})
print("per-kernel MEAN counter value per dispatch (rocprofv3 --pmc, gfx950); FETCH_SIZE/WRITE_SIZE in KiB as reported")
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", [0, 1])[0]):
    print(f"\n== {k}")
    for c in names:
        if c in acc[k]:
            s, n = acc[k][c]
            print(f"   {c:28s} {s / n:18.1f}   (dispatches {n})")

# HBM traffic per kernel launch, corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950:
# FETCH_SIZE (KiB) under-reports wide coalesced reads by exactly 2x, WRITE_SIZE (KiB) is taken as is.
import json
traffic = {}
for k in acc:
    if "FETCH_SIZE" in acc[k] and "WRITE_SIZE" in acc[k]:
        f = acc[k]["FETCH_SIZE"][0] / acc[k]["FETCH_SIZE"][1]
        w = acc[k]["WRITE_SIZE"][0] / acc[k]["WRITE_SIZE"][1]
        traffic[k] = {"fetch_KiB_reported": f, "write_KiB_reported": w,
                      "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
                      "correction": "2*FETCH_SIZE + WRITE_SIZE (KiB->bytes); calibrated on k_preprocess whose reads are a pure stream"}
# stamp: which kernel sources produced these numbers (bench.py replays them only for the same sources)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import bench
    traffic["_kernel_source_sha"] = bench.kernel_source_sha()
except Exception as e:  # noqa: BLE001
    traffic["_kernel_source_sha"] = "unknown: " + str(e)
traffic["_collected"] = "rocprofv3 --kernel-trace --pmc passes of `python bench.py --no-cpu-baseline --no-extra-legs ...` (tools/gpu_pmc.sh " + os.path.basename(out.rstrip("/")) + ")"
json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
