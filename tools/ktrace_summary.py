#!/usr/bin/env python3
"""Per-kernel launch durations from a rocprofv3 --kernel-trace run (csv): for every kernel the dispatches are grouped by grid
size, and per group the count, MEDIAN, mean, min and max are printed -- so that the 1024-frame launches of the timed region are
not averaged with the 32-frame set-up launches or the one-frame leg (the plain --stats average of k_blur_groups
read 778 us against 578 us in the bench line).  usage: tools/ktrace_summary.py <dir> [min_calls=3]"""
import csv, glob, sys, collections, statistics
d = sys.argv[1]
min_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = [x for x in glob.glob(d + "/**/*kernel_trace.csv", recursive=True)][0]
g = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    if name.startswith("void "):
        name = name[5:]
    grid = (r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
    g[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = []
for (name, grid), v in g.items():
    if len(v) < min_calls or not name.startswith("k_"):
        continue
    v2 = v[len(v) // 5:]  # the first fifth of a group's launches: cold caches / first touch
    rows.append((sum(v2), name, grid, len(v), statistics.median(v2), sum(v2) / len(v2), min(v2), max(v2)))
print(f"{'kernel':34s} {'grid (threads x,y,z)':>24s} {'calls':>6s} {'median_us':>10s} {'mean_us':>9s} {'min_us':>9s} {'max_us':>9s}")
for tot, name, grid, n, med, mean, lo, hi in sorted(rows, reverse=True)[:60]:
    print(f"{name[:34]:34s} {'x'.join(x for x in grid if x):>24s} {n:6d} {med:10.2f} {mean:9.2f} {lo:9.2f} {hi:9.2f}")
