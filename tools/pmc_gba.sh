#!/bin/bash
# Developer tool (GPU box): PMC counter groups over the config-4 global BA (tools/gba_run.py), per-kernel averages.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
filt=${1:-k_ba_pairs}
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  d=$R/gpurun_out/pmcg/g$i
  rm -rf $d; mkdir -p $d
  timeout 300 rocprofv3 --pmc $grp -d $d --output-format csv -- python3 $R/tools/gba_run.py > $d/log.txt 2>&1
  echo "== $grp"
  python3 $R/tools/pmc_summary.py $d $filt 2>&1 | head -6
done <<'GROUPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU
TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
GROUPS
