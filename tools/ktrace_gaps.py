#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace run (csv): for every kernel the idle time in FRONT of it -- its start minus the end of the
dispatch before it (by start time, any queue) -- median over the steady-state launches, next to its own median duration.
What HIP events on the stream charge to a stage and the kernel trace does not.  usage: tools/ktrace_gaps.py <dir>"""
import csv, glob, sys, collections, statistics
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Grid_Size_X", "?")))
rows.sort()
gap, dur, prevname = collections.defaultdict(list), collections.defaultdict(list), collections.defaultdict(collections.Counter)
for i in range(1, len(rows)):
    s, e, n, g = rows[i]
    key = (n, g)
    gap[key].append((s - rows[i - 1][1]) / 1e3)
    dur[key].append((e - s) / 1e3)
    prevname[key][rows[i - 1][2][:24]] += 1
print(f"{'kernel':30s} {'grid x':>9s} {'calls':>6s} {'gap_before_us':>14s} {'duration_us':>12s}  usually behind")
for key in sorted(gap, key=lambda k: -sum(dur[k])):
    if len(gap[key]) < 5 or not key[0].startswith("k_"):
        continue
    g = gap[key][len(gap[key]) // 5:]
    print(f"{key[0][:30]:30s} {key[1]:>9s} {len(gap[key]):6d} {statistics.median(g):14.2f} {statistics.median(dur[key][len(dur[key]) // 5:]):12.2f}  {prevname[key].most_common(1)[0][0]}")
