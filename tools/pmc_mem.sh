#!/bin/bash
# Developer tool (GPU box): memory-side rocprofv3 --pmc passes (TA / TCP / TCC, HBM bytes, matrix-pipe busy) over any python
# probe script, per-kernel averages by tools/pmc_summary.py.   tools/pmc_mem.sh <kernel-substring> <script.py> [args...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
filt=$1; shift
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  d=$R/gpurun_out/pmcm/g$i
  rm -rf $d; mkdir -p $d
  VO_EXT_REPS=${VO_EXT_REPS:-3} timeout 150 rocprofv3 --pmc $grp -d $d --output-format csv -- python3 "$@" > $d/log.txt 2>&1
  echo "== $grp"
  python3 $R/tools/pmc_summary.py $d $filt 2>&1 | head -6
done <<'GROUPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
FETCH_SIZE
WRITE_SIZE
GROUPS
rm -rf $R/gpurun_out/pmcm
