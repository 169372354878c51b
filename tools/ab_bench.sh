#!/bin/bash
# Developer tool (GPU box): the tracked-frame step of the product build and of every variant under vo_slam_test_amd/_variants/,
# interleaved and repeated: frames/s, ms per step, and the one-batch-in-flight stage times of the searches and solves.
cd "$(dirname "$0")/.."
one() {
  python bench.py --steps 20 --warmup 3 --no-ba --no-bruteforce --no-single-stream --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['one_batch_in_flight']['stage_ms_per_launch']
print(d['value'], d['ms_per_step'], 'alone:', {k: s[k] for k in ('frame_post','match_last_frame','pose_only_1','match_local_map','pose_only_2')})"
}
for rep in 1 2 3; do
  echo "product   $(one)"
  for so in vo_slam_test_amd/_variants/libvo_*.so; do
    [ -e "$so" ] || continue
    echo "$(basename $so .so) $(VO_HIP_LIB=$so one)"
  done
done
