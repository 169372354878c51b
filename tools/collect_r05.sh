#!/bin/bash
# GPU box: everything profiles/r05_* is made from (kernel stats, traffic, PMC groups for the tracked step, for the
# config-4 global BA kernels and for the pose-only solve).  Output: gpurun_out/refresh/.
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/refresh_profiles.sh 05 > /dev/null 2>&1
O=$R/gpurun_out/refresh
for k in k_ba_pairs k_chol_tiles k_chol_back k_ba_backsub; do
  echo "##### $k"; bash $R/tools/pmc_gba.sh $k 2>&1 | grep -v "^W2\|^E2\|amdgpu.ids"
done > $O/r05_pmc_global_ba.txt 2>&1
bash $R/tools/pmc_probe.sh k_pose_only $R/tools/pose_probe.py 2>&1 | grep -v "^W2\|^E2\|amdgpu.ids" > $O/r05_pmc_pose_only.txt
bash $R/tools/gba_ktrace.sh 2>&1 | grep -E "^k_|LM it" > $O/r05_global_ba_kernels.txt
for f in orb match guided tracker ba chol pose_graph loop; do echo "== $f.hip"; fl=-ffp-contract=off; case $f in ba|chol|pose_graph) fl=-ffp-contract=fast;; esac; python3 $R/tools/kernel_resources.py $R/vo_slam_test_amd/csrc/$f.hip $fl -I$R/vo_slam_test_amd/csrc; done > $O/r05_kernel_resources.txt 2>&1
python3 $R/tools/valu_rate_table.py $O/r05_pmc_summary.txt $O/r05_kernel_stats.csv $O/r05_bench.json > $O/r05_valu_rate_table.txt 2>&1
rm -rf $R/gpurun_out/pmcg $R/gpurun_out/pmcp $R/gpurun_out/gba_ktrace
ls -la $O
