#!/bin/bash
# GPU box: everything profiles/r03_* is made from (kernel stats, traffic, PMC groups for the tracked step, for the
# config-4 global BA kernels and for the pose-only solve).  Output: gpurun_out/refresh/.
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/refresh_profiles.sh 03 > /dev/null 2>&1
O=$R/gpurun_out/refresh
for k in k_ba_pairs k_chol_tiles k_chol_back k_ba_backsub; do
  echo "##### $k"; bash $R/tools/pmc_gba.sh $k 2>&1 | grep -v "^W2\|^E2\|amdgpu.ids"
done > $O/r03_pmc_global_ba.txt 2>&1
bash $R/tools/pmc_probe.sh k_pose_only $R/tools/pose_probe.py 2>&1 | grep -v "^W2\|^E2\|amdgpu.ids" > $O/r03_pmc_pose_only.txt
bash $R/tools/gba_ktrace.sh 2>&1 | grep -E "^k_|LM it" > $O/r03_global_ba_kernels.txt
rm -rf $R/gpurun_out/pmcg $R/gpurun_out/pmcp $R/gpurun_out/gba_ktrace
ls -la $O
