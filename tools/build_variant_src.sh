#!/bin/bash
# Developer tool: a libvo_hip variant in which ONE translation unit is replaced by another version of its source
# (or the same source with extra -D flags).
#   tools/build_variant_src.sh NAME ba /path/to/other/ba.hip [-DFOO=1 ...]  ->  vo_slam_test_amd/_variants/libvo_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; unit=$2; srcfile=$3; shift 3
mkdir -p vo_slam_test_amd/_variants vo_slam_test_amd/_obj/variants
C="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -Ivo_slam_test_amd/csrc"
contract=off
case $unit in ba|pose_graph|chol) contract=fast;; esac
/opt/rocm/bin/hipcc $C -ffp-contract=$contract "$@" -c $srcfile -o vo_slam_test_amd/_obj/variants/${unit}_$name.o 2>/dev/null
others=$(ls vo_slam_test_amd/_obj/*.o | grep -v "/${unit}\.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o vo_slam_test_amd/_variants/libvo_$name.so $others vo_slam_test_amd/_obj/variants/${unit}_$name.o -lz
echo built vo_slam_test_amd/_variants/libvo_$name.so
