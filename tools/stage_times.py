"""Developer tool: per-stage extraction times (instrumented mode) for the library named by VO_HIP_LIB."""
import json, pathlib, subprocess, sys
root = pathlib.Path(__file__).resolve().parent.parent
r = subprocess.run([sys.executable, str(root / "bench.py"), "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-ba"],
                   capture_output=True, text=True)
try:
    d = json.loads(r.stdout.strip().splitlines()[-1])
    print(round(d["value"]), d["ms_per_step"], d["stage_ms_per_launch"])
except Exception as e:  # noqa: BLE001
    print("failed", e, r.stdout[-500:], r.stderr[-1500:])
