"""Batched, device-resident tracked-frame path: a thin Python shell over vo_tracker (csrc/tracker.hip, include/vo_hip.h),
which runs what VisualOdometry::trackWithMotion + trackLocalMap do per frame (reference src/visualOdometry.cpp:228-251,
286-300, 745-775, 864-886) behind the C-ABI.  Nothing here computes or sequences kernels; it stacks per-frame map arrays
into the [batch, n, ...] layout the C entry points take (round 2 kept the 21-launch sequence in this file)."""
from __future__ import annotations

import numpy as np

from . import _lib as vo


def stack_maps(maps, which, keys, n=None):
    """maps: list (one per frame) of tuples as synth.make_tracking_map returns; which: 2 = last frame, 3 = local map.
    -> dict key -> [B, n, ...] array (zero padded: flag 0 = no point)"""
    B = len(maps)
    n = n or max(len(m[which]["flags"]) for m in maps)
    out = {}
    for k in keys:
        a0 = np.asarray(maps[0][which][k])
        o = np.zeros((B, n) + a0.shape[1:], a0.dtype)
        if k == "link":
            o[:] = -1
        for f in range(B):
            a = np.asarray(maps[f][which][k])[:n]
            o[f, :len(a)] = a
        out[k] = o
    return out


def load_maps(trk: "vo.Tracker", maps, n_last=None, n_local=None):
    """hand the synthetic maps of synth.make_tracking_map (one per frame of the batch) to a tracker"""
    last = stack_maps(maps, 2, ("points", "flags", "octave", "angle", "desc"), n_last)
    local = stack_maps(maps, 3, ("points", "normals", "min_dist", "max_dist", "valid", "desc", "link"), n_local)
    trk.set_last_frame(np.stack([m[0] for m in maps]), last["points"], last["flags"], last["octave"], last["angle"], last["desc"])
    trk.set_local_map(local["points"], local["normals"], local["min_dist"], local["max_dist"], local["valid"], local["desc"],
                      link=local["link"])
    return last, local
