#!/bin/bash
# GPU box: re-collect everything under profiles/ for round NN (default 02).  Run from the repo root via gpurun;
# results land in gpurun_out/refresh/ and are copied into profiles/ by the caller (gpurun_out/ is what travels back).
#   tools/refresh_profiles.sh [NN]
R=${GRAFT_REPO_ROOT:-/root/repo}
NN=${1:-02}
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace + stats of the bench command
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt --output-format csv -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-single-stream > $O/kt.log 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1)
cp "$f" $O/r${NN}_kernel_stats.csv
python3 $R/tools/prof_summary.py $O/kt 40 > $O/r${NN}_kernel_stats_summary.txt 2>&1
# 2. HBM-side traffic: separate passes per counter (never combined with trace domains)
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-ba --no-cpu-baseline --no-bruteforce --no-single-stream > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-ba --no-cpu-baseline --no-bruteforce --no-single-stream > $O/write.log 2>&1
python3 $R/tools/make_traffic_json.py $O/fetch $O/write $O/r${NN}_traffic_pmc.json > $O/traffic.log 2>&1
# 3. instruction-mix / LDS / texture-path / L2 counters per kernel (one --pmc pass per group)
bash $R/tools/pmc_sweep.sh > $O/r${NN}_pmc_summary.txt 2>&1
# 4. the bench line itself (with the fresh traffic figure in place)
cp $O/r${NN}_traffic_pmc.json $R/profiles/traffic.json 2>/dev/null
cd $R && python3 bench.py > $O/bench.log 2>&1
tail -1 $O/bench.log > $O/r${NN}_bench.json
rm -rf $O/kt $O/fetch $O/write $R/gpurun_out/pmcs
ls -la $O
