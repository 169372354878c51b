#!/bin/bash
# Developer tool: build libvo_hip variants with extra -D flags for orb.hip (A/B timing on the GPU box).
#   tools/build_variant.sh NAME -DFOO=1 ...   ->  vo_slam_test_amd/_variants/libvo_NAME.so  (use with VO_HIP_LIB=...)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p vo_slam_test_amd/_variants
C="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $C -ffp-contract=off "$@" -c vo_slam_test_amd/csrc/orb.hip -o vo_slam_test_amd/_obj/orb_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o vo_slam_test_amd/_variants/libvo_$name.so \
  vo_slam_test_amd/_obj/vo_common.o vo_slam_test_amd/_obj/orb_$name.o vo_slam_test_amd/_obj/match.o \
  vo_slam_test_amd/_obj/ba.o vo_slam_test_amd/_obj/pose_graph.o vo_slam_test_amd/_obj/chol.o vo_slam_test_amd/_obj/guided.o vo_slam_test_amd/_obj/track.o vo_slam_test_amd/_obj/loop.o vo_slam_test_amd/_obj/dataset_io.o -lz
echo built vo_slam_test_amd/_variants/libvo_$name.so
