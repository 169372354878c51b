#!/bin/bash
# Developer tool (GPU box): PMC counters of the extraction kernels (one rocprofv3 --pmc pass per counter group) for the
# product library and for every variant under vo_slam_test_amd/_variants/ -- tools/pmc_lp.sh [fused=1|0] [kernel-substring]
R="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
export VO_EXT_FUSED=${1:-1} VO_EXT_REPS=2
filt=${2:-k_}
for so in "" $R/vo_slam_test_amd/_variants/libvo_*.so; do
  [ -n "$so" ] && [ ! -e "$so" ] && continue
  if [ -n "$so" ]; then export VO_HIP_LIB=$so; else unset VO_HIP_LIB; fi
  echo "##### ${so:-product} (fused=$VO_EXT_FUSED)"
  i=0
  while read -r grp; do
    [ -z "$grp" ] && continue
    i=$((i+1))
    d=$R/gpurun_out/pmcs_lp/g$i
    rm -rf $d; mkdir -p $d
    timeout 300 rocprofv3 --pmc $grp -d $d --output-format csv -- python3 $R/tools/ext_stage_times.py > $d/log.txt 2>&1
    python3 $R/tools/pmc_summary.py $d $filt 2>&1 | grep -v "^at::\|elementwise\|vectorized\|rocclr\|fill_\|copy" | head -12
  done <<'GROUPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
GROUPS
done
