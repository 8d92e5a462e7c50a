#!/usr/bin/env python3
"""Aggregate rocprofv3 counter_collection CSVs (one per PMC pass) into per-kernel figures of ONE regime.

A bench run holds warm-up frames (cold: the first frames of a cloud are unculled / front-slab frames) in front of the steady regime
its timed region runs in; round 4 averaged them in (k_preprocess: min 37 us, max 217 us in one "culled" run).  Per kernel and pass the
dispatches whose duration exceeds 1.5 x the kernel's median duration in that pass are DROPPED before averaging (the count is printed);
a CSV without timestamps falls back to the median dispatch of every counter."""
import csv
import glob
import os
import statistics
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))      # kernel -> counter -> [sum, dispatches kept]
dropped = defaultdict(lambda: [0, 0])                          # kernel -> [dispatch rows dropped, rows seen]
for path in sorted(glob.glob(os.path.join(out, "pass*.csv"))):
    rows = list(csv.DictReader(open(path)))
    have_t = bool(rows) and "Start_Timestamp" in rows[0] and "End_Timestamp" in rows[0]
    per = defaultdict(lambda: defaultdict(list))               # kernel -> counter -> [(value, duration)]
    for row in rows:
        k = row["Kernel_Name"].split("(")[0]
        dur = (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) if have_t else None
        per[k][row["Counter_Name"]].append((float(row["Counter_Value"]), dur))
    for k, counters in per.items():
        for c, vals in counters.items():
            if have_t and len(vals) >= 4:
                med = statistics.median(d for _, d in vals)
                keep = [v for v, d in vals if d <= 1.5 * med]
            elif len(vals) >= 4:
                keep = [statistics.median(v for v, _ in vals)]
            else:
                keep = [v for v, _ in vals]
            dropped[k][0] += len(vals) - len(keep) if (have_t or len(vals) < 4) else 0
            dropped[k][1] += len(vals)
            a = acc[k][c]
            a[0] += sum(keep)
            a[1] += len(keep)
names = sorted({c for k in acc for c in acc[k]})
print("per-kernel MEAN counter value per dispatch of the STEADY regime (rocprofv3 --pmc, gfx950; dispatches longer than 1.5 x the kernel's "
      "median duration -- warm-up / cold frames -- dropped); FETCH_SIZE/WRITE_SIZE in KiB as reported")
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", [0, 1])[0]):
    print(f"\n== {k}   (dropped {dropped[k][0]} of {dropped[k][1]} counter rows as cold dispatches)")
    for c in names:
        if c in acc[k]:
            s, n = acc[k][c]
            print(f"   {c:28s} {s / max(n, 1):18.1f}   (dispatches {n})")

# HBM traffic per kernel launch, corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950:
# FETCH_SIZE (KiB) under-reports wide coalesced reads by exactly 2x, WRITE_SIZE (KiB) is taken as is.
import json
traffic = {}
for k in acc:
    if "FETCH_SIZE" in acc[k] and "WRITE_SIZE" in acc[k]:
        f = acc[k]["FETCH_SIZE"][0] / max(acc[k]["FETCH_SIZE"][1], 1)
        w = acc[k]["WRITE_SIZE"][0] / max(acc[k]["WRITE_SIZE"][1], 1)
        traffic[k] = {"fetch_KiB_reported": f, "write_KiB_reported": w, "dispatches": acc[k]["FETCH_SIZE"][1],
                      "cold_rows_dropped": dropped[k][0], "rows_seen": dropped[k][1],
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
