#!/bin/bash
# Developer tool (GPU box): kernel trace of a short bench run; average duration per (kernel, grid size).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
d=$R/gpurun_out/ktrace; rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace -d $d --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-ba --no-cpu-baseline --no-bruteforce "$@" > $d/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$d/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    if "at::" in name or "elementwise" in name: continue
    key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
    acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]) / len(kv[1])):
    print("%-40s grid=%s,%s,%s  n=%d  avg=%.1f us  min=%.1f" % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v), min(v)))
PY
