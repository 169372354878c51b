#!/bin/bash
# Developer tool: a libvo_hip variant in which ONE translation unit is replaced by another version of its source
# (or the same source with extra -D flags).
#   tools/build_variant_src.sh NAME ba /path/to/other/ba.hip [-DFOO=1 ...]  ->  vo_slam_test_amd/_variants/libvo_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; unit=$2; srcfile=$3; shift 3
python -m vo_slam_test_amd.build >/dev/null   # the other units' objects must be current
mkdir -p vo_slam_test_amd/_variants vo_slam_test_amd/_obj/variants
C="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -Wno-pass-failed -Ivo_slam_test_amd/csrc"
contract=off
case $unit in ba|pose_graph|chol) contract=fast;; esac
obj=vo_slam_test_amd/_obj/variants/${unit}_$name.o
if ! /opt/rocm/bin/hipcc $C -ffp-contract=$contract "$@" -c $srcfile -o $obj 2>/tmp/variant_$name.err; then
  cat /tmp/variant_$name.err >&2; echo "variant $name FAILED to compile" >&2; exit 1
fi
# the link list comes from build.py's SOURCES (not a glob of _obj/: stale objects of earlier variants live there)
others=$(python - <<PY
from vo_slam_test_amd import build as b
print(" ".join(str(b.PKG / "_obj" / (s.rsplit(".", 1)[0] + ".o")) for s, _ in b.SOURCES if s.rsplit(".", 1)[0] != "$unit"))
PY
)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o vo_slam_test_amd/_variants/libvo_$name.so $others $obj -lz
echo built vo_slam_test_amd/_variants/libvo_$name.so
