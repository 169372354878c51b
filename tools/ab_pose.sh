#!/bin/bash
# Developer tool (GPU box): device-resident pose-only solve time of the product build and of every variant, interleaved.
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  echo "product   $(python tools/pose_probe.py 2>&1 | grep 'ms per launch')"
  for so in vo_slam_test_amd/_variants/libvo_*.so; do
    [ -e "$so" ] || continue
    echo "$(basename $so .so) $(VO_HIP_LIB=$so python tools/pose_probe.py 2>&1 | grep 'ms per launch')"
  done
done
