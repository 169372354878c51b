#!/usr/bin/env python3
"""Timeline of the BA kernels from a rocprofv3 kernel trace: per-kernel duration and idle gaps."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_ba_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
start = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n = int(sys.argv[3]) if len(sys.argv) > 3 else 24
w = rows[start:start + n]
t0 = int(w[0]["Start_Timestamp"]); prev = None
for r in w:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    print(f"{name:16s} t={(s-t0)/1e3:9.2f} dur={(e-s)/1e3:7.2f} gap={(s-prev)/1e3 if prev else 0:6.2f}")
    prev = e
