#!/bin/bash
# GPU box: what profiles/r06_* is made from.  Every rocprofv3 pass has its own timeout (a counter group the profiler does not
# like ended one gpurun call after 20 minutes this round).  Stages are selected by argument so that a call stays short:
#   tools/collect_r06.sh trace | traffic | pmc | mfma | gba | pose | bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ONE="--pipeline 1 --no-bruteforce --no-single-stream --no-ba --no-cpu-baseline"   # ONE regime per kernel
for what in "$@"; do case $what in
trace)
  # the tracked step, one batch in flight: every extraction / tracking kernel has one regime
  rm -rf $O/kt; timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt --output-format csv -- python3 $R/bench.py --steps 20 --warmup 2 $ONE > $O/kt.log 2>&1
  cp "$(find $O/kt -name '*kernel_stats.csv' | head -1)" $O/r06_kernel_stats.csv
  python3 $R/tools/ktrace_summary.py $O/kt > $O/r06_kernel_stats_summary.txt 2>&1
  grep '^{"metric"' $O/kt.log | tail -1 > $O/r06_bench_one_batch_in_flight.json
  # the extract + all-pairs matching leg (k_hamming_mfma) in the library's default order
  rm -rf $O/kte; timeout 300 rocprofv3 --kernel-trace --stats -d $O/kte --output-format csv -- python3 $R/tools/ebm_probe.py ham=0 blur=0 reps=2 stages=0 > $O/kte.log 2>&1
  python3 $R/tools/ktrace_summary.py $O/kte > $O/r06_kernel_stats_extract_match.txt 2>&1
  rm -rf $O/kt $O/kte ;;
traffic)
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/$c; timeout 400 rocprofv3 --pmc $c -d $O/$c --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 $ONE > $O/$c.log 2>&1
  done
  python3 $R/tools/make_traffic_json.py $O/FETCH_SIZE $O/WRITE_SIZE $O/r06_traffic_pmc.json > $O/traffic.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/h$c; VO_HAM_NOCHECK=1 timeout 200 rocprofv3 --pmc $c -d $O/h$c --output-format csv -- python3 $R/tools/ham_probe.py 1024 0 > $O/h$c.log 2>&1
    echo "== $c (tools/ham_probe.py 1024 0)"; python3 $R/tools/pmc_summary.py $O/h$c k_hamming
  done > $O/r06_traffic_hamming.txt 2>&1
  rm -rf $O/FETCH_SIZE $O/WRITE_SIZE $O/hFETCH_SIZE $O/hWRITE_SIZE ;;
pmc)
  bash $R/tools/pmc_sweep.sh > $O/r06_pmc_summary.txt 2>&1
  rm -rf $R/gpurun_out/pmcs ;;
mfma)
  # the matrix-pipe counters of the kernels that use it (VERDICT r5 #1b: record SQ_VALU_MFMA_BUSY_CYCLES): k_hamming_mfma, k_describe_win
  # (default extraction) and k_blur_mfma (VO_ORB_OPT_DESCRIBE_BLUR = 1)
  { for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
      rm -rf $O/pm; VO_HAM_NOCHECK=1 timeout 200 rocprofv3 --pmc $grp -d $O/pm --output-format csv -- python3 $R/tools/ham_probe.py 1024 0 > $O/pm.log 2>&1
      echo "== $grp"; python3 $R/tools/pmc_summary.py $O/pm k_hamming
      rm -rf $O/pm; VO_EXT_REPS=3 timeout 200 rocprofv3 --pmc $grp -d $O/pm --output-format csv -- python3 $R/tools/ext_stage_times.py > $O/pm.log 2>&1
      python3 $R/tools/pmc_summary.py $O/pm k_describe
      rm -rf $O/pm; VO_EXT_DESCBLUR=1 VO_EXT_REPS=3 timeout 200 rocprofv3 --pmc $grp -d $O/pm --output-format csv -- python3 $R/tools/ext_stage_times.py > $O/pm.log 2>&1
      python3 $R/tools/pmc_summary.py $O/pm k_blur
    done; } > $O/r06_pmc_mfma_kernels.txt 2>&1
  rm -rf $O/pm $R/gpurun_out/pmcs ;;
gba)
  for k in k_ba_pairs k_chol_tiles k_chol_back k_ba_backsub; do echo "##### $k"; bash $R/tools/pmc_gba.sh $k 2>&1 | grep -v "^W2\|^E2\|amdgpu.ids"; done > $O/r06_pmc_global_ba.txt 2>&1
  bash $R/tools/gba_ktrace.sh 2>&1 | grep -E "^k_|LM it" > $O/r06_global_ba_kernels.txt
  rm -rf $R/gpurun_out/pmcg $R/gpurun_out/gba_ktrace ;;
pose)
  bash $R/tools/pmc_probe.sh k_pose_only $R/tools/pose_probe.py 2>&1 | grep -v "^W2\|^E2\|amdgpu.ids" > $O/r06_pmc_pose_only.txt
  rm -rf $R/gpurun_out/pmcp ;;
bench)
  cd $R && python3 bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/r06_bench.json
  for f in orb match guided tracker ba chol pose_graph loop; do echo "== $f.hip"; fl=-ffp-contract=off; case $f in ba|chol|pose_graph) fl=-ffp-contract=fast;; esac; python3 $R/tools/kernel_resources.py $R/vo_slam_test_amd/csrc/$f.hip $fl -I$R/vo_slam_test_amd/csrc; done > $O/r06_kernel_resources.txt 2>&1 ;;
esac; done
ls -la $O
