#!/usr/bin/env python3
"""One steady-state frame as a timeline, from a rocprofv3 --kernel-trace CSV: python tools/frame_timeline.py <kernel_trace.csv>
Start of every kernel relative to the frame's first kernel, its duration, and the idle gap before it."""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
if len(sys.argv) > 2 and sys.argv[2] == "--gaps":
    # idle time in front of every kernel, over the whole trace: median / 90th percentile per kernel name
    import statistics
    gaps = {}
    for p_, r in zip(rows, rows[1:]):
        g = (int(r["Start_Timestamp"]) - int(p_["End_Timestamp"])) / 1e3
        gaps.setdefault(r["Kernel_Name"].split("(")[0][:40], []).append(g)
    print("%-40s %6s %9s %9s %9s" % ("idle in front of", "n", "median us", "p90 us", "min us"))
    for k, v in sorted(gaps.items(), key=lambda kv: -statistics.median(kv[1])):
        if len(v) >= 20:
            v.sort()
            print("%-40s %6d %9.1f %9.1f %9.1f" % (k, len(v), statistics.median(v), v[int(0.9 * (len(v) - 1))], v[0]))
    sys.exit(0)
# a frame = from one k_preprocess to the next; take the last complete one
starts = [i for i, r in enumerate(rows) if "k_preprocess" in r["Kernel_Name"]]
if len(starts) < 3:
    sys.exit("not enough frames in the trace")
# optional second argument: which frame (index into the frames of the trace; default: the last complete one), or "median": the
# frame of median period among the last 40 complete frames (a single frame can carry a host hiccup or the events of a timed frame)
if len(sys.argv) > 2 and sys.argv[2] == "median":
    cand = list(range(max(0, len(starts) - 41), len(starts) - 1))
    per = sorted((int(rows[starts[i + 1]]["Start_Timestamp"]) - int(rows[starts[i]]["Start_Timestamp"]), i) for i in cand)
    which = per[len(per) // 2][1]
else:
    which = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 2
nseg = int(sys.argv[3]) if len(sys.argv) > 3 else 1          # optional third argument: consecutive segments to print (a front-slab frame is two)
nseg = max(1, min(nseg, len(starts) - 1 - which))
a, b = starts[which], starts[which + nseg]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = None
busy = 0
print("%-34s %9s %9s %8s" % ("kernel", "start us", "dur us", "gap us"))
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%-34s %9.1f %9.1f %8.1f" % (r["Kernel_Name"].split("(")[0][:34], (s - t0) / 1e3, (e - s) / 1e3, gap))
    busy += e - s
    prev_end = e
span = int(rows[b]["Start_Timestamp"]) - t0
print("frame period %.1f us, kernels busy %.1f us, idle between kernels %.1f us (%d launches)" %
      (span / 1e3, busy / 1e3, (span - busy) / 1e3, b - a))
