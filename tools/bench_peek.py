"""Developer tool: print selected keys of bench.py's JSON line.  usage: python tools/bench_peek.py file.json [key ...]"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
keys = sys.argv[2:] or ["value", "ms_per_step", "stage_ms_per_launch", "one_batch_in_flight", "single_stream", "ingest_inclusive",
                        "extract_match_frac_of_hbm_peak", "roofline", "cpu_baseline"]
for k in keys:
    print(k, "=", json.dumps(d.get(k)))
