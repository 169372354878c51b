#!/bin/bash
# Developer tool (GPU box): the fused per-level pass against the three separate kernels, interleaved and repeated on ONE box
# (per-stage ms per 1024 frames; fused: `pyramid` carries the whole chain of level passes, `fast` and `blur` are empty).
# Variant libraries under vo_slam_test_amd/_variants/ are measured in fused mode as well.
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  echo "separate  $(VO_EXT_FUSED=0 python tools/ext_stage_times.py 2>&1 | tail -1)"
  echo "fused     $(VO_EXT_FUSED=1 python tools/ext_stage_times.py 2>&1 | tail -1)"
  for so in vo_slam_test_amd/_variants/libvo_*.so; do
    [ -e "$so" ] || continue
    echo "$(basename $so .so) $(VO_EXT_FUSED=1 VO_HIP_LIB=$so python tools/ext_stage_times.py 2>&1 | tail -1)"
  done
done
