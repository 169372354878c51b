#!/bin/bash
# Developer tool (GPU box): per-launch durations of one kernel of the config-4 global BA (tools/gba_run.py) for a list of settings
#   tools/dev/probe_run.sh k_ba_pairs "VO_PAIRS=0" "VO_PAIRS=1"
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
K=$1; shift
for v in "$@"; do
  d=$R/gpurun_out/probe; rm -rf $d; mkdir -p $d
  env $v timeout 300 rocprofv3 --kernel-trace -d $d --output-format csv -- python3 $R/tools/gba_run.py > $d/log.txt 2>&1
  echo "=== $v: $(grep 'LM it' $d/log.txt | tail -1)"
  python3 - <<PY
import csv, glob
f = glob.glob("$d/**/*kernel_trace.csv", recursive=True)[0]
v = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0 for r in csv.DictReader(open(f)) if "$K" in r["Kernel_Name"])
print("$K n=%d" % len(v), " ".join("%.0f" % x for x in v))
PY
  rm -rf $d
done
