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
