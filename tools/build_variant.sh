#!/bin/bash
# Developer tool: build libvo_hip variants with extra -D flags for orb.hip (A/B timing on the GPU box).
#   tools/build_variant.sh NAME -DFOO=1 ...   ->  vo_slam_test_amd/_variants/libvo_NAME.so  (use with VO_HIP_LIB=...)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
exec tools/build_variant_src.sh "$name" orb vo_slam_test_amd/csrc/orb.hip "$@"
