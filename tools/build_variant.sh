#!/bin/bash
# Developer tool: build libvo_hip variants with extra -D flags for orb.hip (A/B timing on the GPU box).
#   tools/build_variant.sh NAME -DFOO=1 ...   ->  vo_slam_test_amd/_variants/libvo_NAME.so  (use with VO_HIP_LIB=...)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p vo_slam_test_amd/_variants
C="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $C -ffp-contract=off "$@" -c vo_slam_test_amd/csrc/orb.hip -o vo_slam_test_amd/_obj/orb_$name.o
others=$(ls vo_slam_test_amd/_obj/*.o | grep -v '/orb' )
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o vo_slam_test_amd/_variants/libvo_$name.so $others vo_slam_test_amd/_obj/orb_$name.o -lz
echo built vo_slam_test_amd/_variants/libvo_$name.so
