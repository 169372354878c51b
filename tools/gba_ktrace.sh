#!/bin/bash
# Developer tool (GPU box): config-4 global BA, per-kernel average durations under rocprofv3.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
d=$R/gpurun_out/gba_ktrace; rm -rf $d; mkdir -p $d
timeout 300 rocprofv3 --kernel-trace -d $d --output-format csv -- python3 $R/tools/gba_run.py "$@" > $d/log.txt 2>&1
tail -3 $d/log.txt
python3 - <<PY
import csv, glob, collections
f = glob.glob("$d/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    if "at::" in name: continue
    acc[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print("%-28s n=%5d  avg=%8.2f us  min=%8.2f  total=%9.1f" % (k, len(v), sum(v) / len(v), min(v), sum(v)))
PY
