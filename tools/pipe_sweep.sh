#!/bin/bash
# Developer tool (GPU box): tracked frames/s against the batches in flight, the stream priorities (extraction, tail) and
# shared / own extraction streams.
cd "$(dirname "$0")/.."
for rep in 1 2; do
for cfg in "2 0,-1 " "2 0,-1 1" "3 0,-1 1" "2 0,0 1" "3 0,-1 "; do
  set -- $cfg
  v=$(VO_BENCH_OWN_EXT_STREAMS=$3 VO_BENCH_PRIO=$2 python bench.py --steps 20 --warmup 3 --pipeline $1 --no-ba --no-bruteforce --no-single-stream --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "pipeline $1 prio $2 own-extraction-streams '$3' -> $v"
done
done
