#!/bin/bash
# Developer tool (GPU box): per-stage extraction times of the product build and of every variant under
# vo_slam_test_amd/_variants/, interleaved and repeated so that box-to-box and clock drift cancel.
#   tools/build_variant.sh noref -DVO_FAST_REFINE=0 ; gpurun -- tools/ab_orb.sh
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  echo "product   $(python tools/ext_stage_times.py 2>&1 | tail -1)"
  for so in vo_slam_test_amd/_variants/libvo_*.so; do
    [ -e "$so" ] || continue
    n=$(basename $so .so)
    echo "$n $(VO_HIP_LIB=$so python tools/ext_stage_times.py 2>&1 | tail -1)"
  done
done
