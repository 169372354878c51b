#!/bin/bash
# interleaved: config-4 global BA and config-3 local BA, product vs variant
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  for v in product bs4; do
    if [ $v = product ]; then unset VO_HIP_LIB; else export VO_HIP_LIB=$R/vo_slam_test_amd/_variants/libvo_$v.so; fi
    g=$(python $R/tools/gba_run.py 2>&1 | grep "^10 LM" | sed -e 's/.*= \([0-9.]*\) iters.*/\1/')
    l=$(python $R/tools/ba_run.py 2>&1 | tail -2 | tr '\n' ' ')
    echo "$v gba=$g it/s | lba: $l"
  done
done
