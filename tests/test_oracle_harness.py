"""BASELINE config 0's loop (tracking + local BA) on the CPU oracle: tests/harness_ref.py -- the script that
examples/vo_run_hip.cpp runs on the device -- tracks a synthetic sequence with rotation against a non-empty local map, runs its
scripted local BA, and recovers the ground-truth camera path."""
import numpy as np

import harness_ref
import harness_seq
from vo_slam_test_amd import synth


def test_oracle_harness_recovers_the_ground_truth(orc):
    n = 12
    grays, raws = harness_seq.render(n)
    inv = np.float32(1.0) / np.float32(synth.DEPTH_SCALE)
    logs = []
    poses, info = harness_ref.run_sequence(orc, synth, list(grays), list(raws), synth.CAM.astype(np.float32), inv, log=logs.append)
    assert all(r["ok"] for r in info)
    assert all(r["n_local"] > 50 and r["tracked"] >= 200 for r in info[2:]), info  # the second search works on a populated local map
    assert len(logs) == 2 and "erased" in logs[0]                                    # local BA after frames 5 and 10
    harness_seq.check_against_truth(poses, tol_pos=0.02, tol_rot=2e-3)
