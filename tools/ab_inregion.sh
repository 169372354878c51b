#!/bin/bash
cd "$(dirname "$0")/.."
one() {
  python bench.py --steps 20 --warmup 3 --no-ba --no-bruteforce --no-single-stream --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms_per_launch']
print(d['value'], d['ms_per_step'], 'in-region:', {k: s[k] for k in ('extract','fast','pose_only_1','pose_only_2')})"
}
for rep in 1 2 3; do
  echo "product   $(one)"
  for so in vo_slam_test_amd/_variants/libvo_*.so; do
    echo "$(basename $so .so) $(VO_HIP_LIB=$so one)"
  done
done
