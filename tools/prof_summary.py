#!/usr/bin/env python3
"""Print the per-kernel summary of a rocprofv3 --kernel-trace --stats run (csv output)."""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
print(f"{'kernel':44s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>9s} {'pct':>6s}")
for r in list(csv.DictReader(open(f)))[:n]:
    name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0][:44]
    print(f"{name:44s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.2f} {float(r['TotalDurationNs'])/1e6:9.3f} {float(r['Percentage']):6.2f}")
