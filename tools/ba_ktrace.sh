#!/bin/bash
# Developer tool (GPU box): config-3 local BA, LM-iterations/s untraced, then per-kernel average durations under rocprofv3.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
python3 $R/tools/ba_run.py 20 2>&1 | grep -v amdgpu.ids
d=$R/gpurun_out/ba_ktrace; rm -rf $d; mkdir -p $d
timeout 300 rocprofv3 --kernel-trace -d $d --output-format csv -- python3 $R/tools/ba_run.py 10 > $d/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$d/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    if "k_ba" not in name: continue
    acc[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print("%-24s n=%5d  avg=%6.2f us  min=%6.2f" % (k, len(v), sum(v) / len(v), min(v)))
PY
