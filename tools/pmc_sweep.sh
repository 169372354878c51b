#!/bin/bash
# Developer tool (GPU box): one rocprofv3 --pmc pass per counter group over a short bench run,
# per-kernel averages printed by tools/pmc_summary.py.   tools/pmc_sweep.sh [kernel-substring]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
filt=$1
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  d=$R/gpurun_out/pmcs/g$i
  rm -rf $d; mkdir -p $d
  timeout 300 rocprofv3 --pmc $grp -d $d --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-ba --no-cpu-baseline --no-bruteforce --no-single-stream > $d/log.txt 2>&1
  echo "== $grp"
  python3 $R/tools/pmc_summary.py $d $filt 2>&1 | grep -v "^at::\|elementwise\|vectorized\|rocclr\|fill_\|copy" | head -24
done <<'GROUPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
GROUPS
