#!/bin/bash
# Developer tool (GPU box): examples/rccl_sharded_ba.cpp as ONE rank whose every collective is a real ncclAllReduce.
cd "$(dirname "$0")/.."
hipcc -O2 -std=c++17 --offload-arch=gfx950 examples/rccl_sharded_ba.cpp -Iinclude -Lvo_slam_test_amd -lvo_hip -L/opt/rocm/lib -lrccl -Wl,-rpath,$PWD/vo_slam_test_amd -o /tmp/rccl_sharded_ba 2>&1 | tail -2
python tools/dump_ba_problem.py /tmp/p_local.bin local > /dev/null
python tools/dump_ba_problem.py /tmp/p_global.bin global > /dev/null
for p in local global; do
  for seg in "" "--segments"; do
    echo "== $p problem, flags: --collectives-at-one-rank $seg"
    RANK=0 WORLD_SIZE=1 VO_NCCL_ID_FILE=/tmp/id_$p /tmp/rccl_sharded_ba /tmp/p_$p.bin --collectives-at-one-rank $seg 2>&1 | grep -v amdgpu.ids | tail -4
  done
done
