#!/bin/bash
# Developer tool (GPU box): rocprofv3 --pmc passes (SQ instruction mix, waits, LDS) over any python probe script,
# per-kernel averages by tools/pmc_summary.py.   tools/pmc_probe.sh <kernel-substring> <script.py> [args...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
filt=$1; shift
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  d=$R/gpurun_out/pmcp/g$i
  rm -rf $d; mkdir -p $d
  timeout 300 rocprofv3 --pmc $grp -d $d --output-format csv -- python3 "$@" > $d/log.txt 2>&1
  echo "== $grp"
  python3 $R/tools/pmc_summary.py $d $filt 2>&1 | head -6
done <<'GROUPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH
GROUPS
