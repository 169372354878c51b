#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE) into profiles/traffic.json:
HBM-side bytes per launch for every bench stage.  Counter unit is KiB (x1024).  On gfx950
FETCH_SIZE under-reports wide streaming reads by 2x (MI355X_MICROARCH.md, HBM section); the
4-byte-per-lane reads used here are uncalibrated, so both the raw and the doubled read figure are
kept and `traffic` uses raw read + write (a lower bound)."""
import collections, csv, glob, json, sys
fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
STAGES = (("k_resize", "pyramid"), ("k_fast", "fast"), ("k_octree", "octree"), ("k_blur", "blur"),
          ("k_describe", "describe"), ("k_hamming", "hamming"), ("k_frame_", "frame_post"),
          ("k_guided_", "match"), ("k_track_", "match"), ("k_pose_only", "pose_only"))
def stage_of(kernel):  # template instances appear as "void k_fast_cell<48, false>"
    for key, st in STAGES:
        if key in kernel:
            return st
    return None
def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        acc[name].append(float(r["Counter_Value"]) * 1024.0)
    return acc
def full_size(acc):
    """drop the launches of bench.py's set-up (32 frames instead of the whole batch): anything under half the
    largest value of its kernel; the 7 pyramid levels differ by more than that, so k_resize keeps the launches of
    full-batch steps by position (the set-up's 7 come first)"""
    out = {}
    for k, v in acc.items():
        if "k_resize" in k:
            out[k] = v[7:] if len(v) > 7 else v
        elif any(t in k for t in ("k_fast", "k_octree", "k_blur", "k_describe", "k_frame_")):  # what the set-up runs
            m = max(v)
            out[k] = [x for x in v if x >= 0.5 * m]
        else:
            out[k] = v
    return out
fe, wr = full_size(load(fetch_dir, "FETCH_SIZE")), full_size(load(write_dir, "WRITE_SIZE"))
steps = max([len(v) for k, v in fe.items() if "k_describe" in k] + [1])
res = {"_note": "bytes per bench step (one launch of each stage over the whole batch); read figure raw",
       "_steps_profiled": steps}
detail = {}
for k in set(fe) | set(wr):
    st = stage_of(k)
    if st is None:
        continue
    rd = sum(fe.get(k, [0])) / steps
    w = sum(wr.get(k, [0])) / steps
    d = detail.setdefault(st, {"read": 0.0, "write": 0.0})
    d["read"] += rd
    d["write"] += w
for st, d in detail.items():
    res[st] = round(d["read"] + d["write"])
res["_detail"] = {k: {a: round(b) for a, b in v.items()} for k, v in detail.items()}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(res))
