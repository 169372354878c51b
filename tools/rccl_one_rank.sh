cd ${GRAFT_REPO_ROOT:-/root/repo}
hipcc -O2 -std=c++17 --offload-arch=gfx950 examples/rccl_sharded_ba.cpp -Iinclude -Lvo_slam_test_amd -lvo_hip -L/opt/rocm/lib -lrccl -Wl,-rpath,$PWD/vo_slam_test_amd -o /tmp/rccl_sharded_ba 2>&1 | tail -2
python tools/dump_ba_problem.py /tmp/p_local.bin local > /dev/null
python tools/dump_ba_problem.py /tmp/p_global.bin global > /dev/null
for p in local global; do
  for seg in 0 1; do
    echo "== $p problem, VO_BA_SEGMENTS=$seg"
    RANK=0 WORLD_SIZE=1 VO_NCCL_ID_FILE=/tmp/id_$p VO_BA_COLLECTIVES_AT_ONE_RANK=1 VO_BA_SEGMENTS=$seg /tmp/rccl_sharded_ba /tmp/p_$p.bin 2>&1 | grep -v amdgpu.ids | tail -4
  done
done
