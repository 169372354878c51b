#!/usr/bin/env python3
"""Per-kernel share of the calibrated VALU issue rate.

usage: tools/valu_rate_table.py profiles/rNN_pmc_summary.txt profiles/rNN_kernel_stats.csv [profiles/rNN_bench.json]

With the bench line given, the extraction kernels' durations are its `one_batch_in_flight` stage times (HIP events around
each stage with nothing else running); the kernel-trace averages mix launches of the timed region, where two batches
are in flight and every launch is stretched by the other batch's kernels.

For every kernel of the tracked step: VALU wave-instructions per launch (SQ_INSTS_VALU of the PMC pass), the launch
duration of the kernel-trace pass, and the time the issue of those instructions alone takes on 1024 SIMDs at the
calibrated rates of profiles/r03_valu_issue_calibration.txt: 4.2 cycles per wave64 instruction (integer VOP3 / packed /
SDWA / DPP / conversions / compares / all FP64), and 2.2 cycles for the classes that dual-rate (f32 add/mul/fma, 16-bit
VOP2, v_mov) -- the kernels here are almost entirely of the first kind, so the 4.2-cycle column is the one to read; the
2.2-cycle column is the floor if every instruction were of the fast kind.  Clock: 2.38 GHz (s_memtime / s_memrealtime
under load in the calibration runs)."""
import csv, re, sys, ast

pmc, stats = sys.argv[1], sys.argv[2]
alone = {}
if len(sys.argv) > 3:
    import json
    st = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])["one_batch_in_flight"]["stage_ms_per_launch"]
    # stage -> (kernel, launches per stage)
    for stage, (k, n) in {"pyramid": ("k_resize4", 7), "fast": ("k_fast_cell", 1), "octree": ("k_octree", 1), "blur": ("k_blur_mfma", 1),
                          "describe": ("k_describe", 1)}.items():
        alone[k] = st[stage] * 1e-3 / n
CLK, SIMDS = 2.38e9, 1024
valu, salu, lds = {}, {}, {}
for line in open(pmc):
    m = re.match(r"(?:void )?(k_\w+)(?:<[^>]*>)? (\{.*\}) n=", line)
    if not m or "SQ_INSTS_VALU" not in m.group(2):
        continue
    d = ast.literal_eval(m.group(2))
    if d["SQ_INSTS_VALU"] < valu.get(m.group(1), 0):
        continue  # several instantiations of one kernel (k_guided_cand<8> / <16>: the retry pass leaves every frame out): keep the working one
    valu[m.group(1)], salu[m.group(1)], lds[m.group(1)] = d["SQ_INSTS_VALU"], d.get("SQ_INSTS_SALU", 0), d.get("SQ_INSTS_LDS", 0)
dur = {}
for r in csv.DictReader(open(stats)):
    m = re.search(r"(k_\w+)", r["Name"])
    if m:
        dur.setdefault(m.group(1), float(r["AverageNs"]) * 1e-9)
print("%-18s %14s %12s %12s %12s %10s %10s" % ("kernel", "VALU instr", "SALU instr", "LDS instr", "launch ms", "% @4.2cy", "% @2.2cy"))
for k in sorted(valu, key=lambda k: -valu[k]):
    if k not in dur:
        continue
    if k in alone:
        dur[k] = alone[k]
    t42 = valu[k] * 4.2 / (SIMDS * CLK)
    t22 = valu[k] * 2.2 / (SIMDS * CLK)
    print("%-18s %14.0f %12.0f %12.0f %12.4f %9.0f%% %9.0f%%" % (k, valu[k], salu[k], lds[k], dur[k] * 1e3, 100 * t42 / dur[k], 100 * t22 / dur[k]))
