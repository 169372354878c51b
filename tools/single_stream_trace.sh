#!/bin/bash
# Developer tool (GPU box): kernel timeline of ONE frame through vo_tracker_track (tools/latency_probe.py under rocprofv3).
R="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/latency_probe.py 2>&1 | grep -v amdgpu.ids | tail -4
d=$R/gpurun_out/sstrace; rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace -d $d --output-format csv -- python3 $R/tools/latency_probe.py > $d/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$d/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last complete frame: from behind the previous k_track_count (the last kernel of a frame) to the last one
idx = [i for i, r in enumerate(rows) if "k_track_count" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[a]["Start_Timestamp"]); prev = None; tot = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:28]
    print(f"{name:28s} t={(s-t0)/1e3:8.1f} dur={(e-s)/1e3:7.1f} gap={(s-prev)/1e3 if prev else 0:6.1f}")
    prev = e; tot += e - s
print("kernels", b - a, "sum of durations %.1f us, span %.1f us" % (tot / 1e3, (int(rows[b-1]["End_Timestamp"]) - t0) / 1e3))
PY
rm -rf $d
