"""The script of examples/vo_run_hip.cpp on the CPU oracle (test infrastructure): BASELINE config 0's loop -- trackWithMotion,
the local map derived between the two stages, trackLocalMap, a local BA over the last frames on a fixed schedule -- with the
same scripted map (every frame a key-frame, unmatched features with depth create points, matched inliers become
observations, erased BA edges drop theirs).  Mirrors the C++ statement by statement: same orders, same float / double types."""
import numpy as np

from track_ref import local_map_stage, track_first

WINDOW, BA_EVERY, MAX_LAST, MAX_LOCAL = 10, 5, 2048, 16384
EYE = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], np.float64)


def _inv(T):
    R, t = T[:9].reshape(3, 3), T[9:]
    return np.concatenate([R.T.reshape(-1), -R.T @ t])


def _mul(A, B):
    Ra, ta, Rb, tb = A[:9].reshape(3, 3), A[9:], B[:9].reshape(3, 3), B[9:]
    return np.concatenate([(Ra @ Rb).reshape(-1), Ra @ tb + ta])


def _centre(T):
    return -T[:9].reshape(3, 3).T @ T[9:]


def run_sequence(orc, synth, grays, raws, cam5, inv_depth, log=None):
    """grays: list of uint8 [H, W]; raws: list of uint16 [H, W] -> (list of Tcw12 per frame as they stand at the end,
    per-frame dicts of counts)"""
    p = orc.orb_params()
    sf = np.array(list(p.scale)[:8], np.float32)
    H, W = grays[0].shape
    frames, mp_pos, mp_desc, mp_nsum, mp_ncnt, mp_ref, mp_level, mp_last, mp_obs = [], [], [], [], [], [], [], [], []
    Tcl = EYE.copy()
    info = []

    def add_normal(m, c):
        d = mp_pos[m] - c
        mp_nsum[m] = mp_nsum[m] + d / np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
        mp_ncnt[m] += 1

    for i in range(len(grays)):
        k, dsc, _ = orc.extract(p, grays[i])
        n = len(k)
        x, y = np.ascontiguousarray(k["x"]), np.ascontiguousarray(k["y"])
        dimg = np.zeros((H, W), np.float32)
        orc.lib().orc_depth_to_float(np.ascontiguousarray(raws[i]).reshape(-1), H * W, inv_depth, dimg.reshape(-1))
        ur, dep = np.zeros(n, np.float32), np.zeros(n, np.float32)
        orc.lib().orc_find_depth(n, x, y, x, dimg, W, H, W, float(cam5[4]), ur, dep)
        fr = dict(n=n, ux=x, uy=y, ur=ur, dep=dep, oct=k["octave"].astype(np.int32), angle=k["angle"].astype(np.float32), desc=dsc,
                  mp=np.full(n, -1, np.int64))
        Tpred, last_ids = EYE.copy(), []
        ok, st = True, dict(n_last=0, status=0)
        Tcw = Tpred
        rec = dict(n_last=0, n_local=0, inliers=0, tracked=0, status=0)
        if i > 0:
            L = frames[i - 1]
            Tpred = _mul(Tcl, L["Tcw"])
            last_ids = [int(m) for m in L["mp"] if m >= 0][:MAX_LAST]
            kk = [q for q in range(L["n"]) if L["mp"][q] >= 0][:MAX_LAST]
            last = dict(points=np.array([mp_pos[m] for m in last_ids]).reshape(-1, 3), flags=np.full(len(last_ids), 3, np.uint8),
                        octave=L["oct"][kk], angle=L["angle"][kk], desc=np.array([mp_desc[m] for m in last_ids], np.uint8).reshape(-1, 32))
            R, t = Tpred[:9].reshape(3, 3), Tpred[9:]
            st = track_first(orc, k, dsc, x, y, ur, Tpred, synth.se3_log(R, t), last, cam5, sf, W, H)
            ok = st["status"] == 0
            rec.update(n_last=st["n_last"], status=st["status"])
            local_ids = []
            if ok:
                pos_in_last = {m: q for q, m in enumerate(last_ids)}
                for m in range(len(mp_pos)):
                    if len(local_ids) >= MAX_LOCAL:
                        break
                    if not mp_obs[m] or mp_last[m] < i - WINDOW:
                        continue
                    local_ids.append(m)
                lp = np.array([mp_pos[m] for m in local_ids]).reshape(-1, 3)
                ln = np.array([mp_nsum[m] / float(mp_ncnt[m]) for m in local_ids]).reshape(-1, 3)
                d = lp - np.array([mp_ref[m] for m in local_ids]).reshape(-1, 3)
                dist = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]).astype(np.float32)
                lmax = (dist * sf[np.array([mp_level[m] for m in local_ids], np.int64)]).astype(np.float32)
                lmin = (lmax / sf[7]).astype(np.float32)
                local = dict(points=lp, normals=ln, min_dist=lmin, max_dist=lmax, valid=np.full(len(local_ids), 3, np.uint8),
                             link=np.array([pos_in_last.get(m, -1) for m in local_ids], np.int32),
                             desc=np.array([mp_desc[m] for m in local_ids], np.uint8).reshape(-1, 32))
                s2 = local_map_stage(orc, st["of"], k, x, y, ur, st["fpt"], st["has"].copy(), st["fobs"], st["pose_1"], st["assigned_last"],
                                     st["n_last_list"], local, cam5, sf, W, H, 3.0, 0.8, st["solve"])
                rec.update(n_local=s2["n_local"], inliers=s2["inliers_2"], tracked=s2["n_tracked"])
                ok = s2["n_tracked"] >= 30  # trackLocalMap's verdict (:304)
                R2, t2 = synth.se3_exp(s2["pose_2"])
                Tcw = np.concatenate([R2.reshape(-1), t2])
            if not ok:
                Tcw = Tpred
        fr["Tcw"] = Tcw
        Twc, centre = _inv(Tcw), _centre(Tcw)
        if ok and i > 0:
            a0, a1, has_end, foutl = st["assigned_last"], s2["assigned_local"], s2["has"], s2["feature_outlier"]
            for q in range(n):
                m = -1
                if a1[q] >= 0:
                    m = local_ids[a1[q]]
                elif a0[q] >= 0 and has_end[q]:
                    m = last_ids[a0[q]]
                if m < 0 or foutl[q]:
                    continue
                if any(o[0] == i for o in mp_obs[m]):
                    continue
                fr["mp"][q] = m
                mp_obs[m].append((i, q))
                mp_last[m] = i
                add_normal(m, centre)
        Rwc, twc = Twc[:9].reshape(3, 3), Twc[9:]
        for q in range(n):
            if fr["mp"][q] >= 0 or not (dep[q] > 0):
                continue
            z = float(dep[q])
            xc = (float(x[q]) - float(cam5[2])) * z / float(cam5[0])
            yc = (float(y[q]) - float(cam5[3])) * z / float(cam5[1])
            pw = np.array([Rwc[r, 0] * xc + Rwc[r, 1] * yc + Rwc[r, 2] * z + twc[r] for r in range(3)])
            m = len(mp_pos)
            mp_pos.append(pw), mp_desc.append(dsc[q].copy()), mp_nsum.append(np.zeros(3)), mp_ncnt.append(0)
            mp_ref.append(centre.copy()), mp_level.append(int(fr["oct"][q])), mp_last.append(i), mp_obs.append([(i, q)])
            add_normal(m, centre)
            fr["mp"][q] = m
        frames.append(fr)
        if i > 0 and i % BA_EVERY == 0:
            f0 = max(0, i - WINDOW + 1)
            nc = i - f0 + 1
            poses = np.array([synth.se3_log(frames[f0 + c]["Tcw"][:9].reshape(3, 3), frames[f0 + c]["Tcw"][9:]) for c in range(nc)])
            fixed = np.zeros(nc, np.uint8)
            fixed[0] = 1
            pid, pts, ecam, ept, eobs, eis, eref = [], [], [], [], [], [], []
            for m in range(len(mp_pos)):
                if sum(1 for o in mp_obs[m] if o[0] >= f0) < 2:
                    continue
                j = len(pid)
                pid.append(m), pts.append(mp_pos[m])
                for qi, o in enumerate(mp_obs[m]):
                    if o[0] < f0:
                        continue
                    F = frames[o[0]]
                    ecam.append(o[0] - f0), ept.append(j)
                    eobs.append([float(F["ux"][o[1]]), float(F["uy"][o[1]]), float(F["ur"][o[1]])])
                    eis.append(1.0 / float(sf[F["oct"][o[1]]]))
                    eref.append((m, qi))
            if pid:
                pr = dict(poses=poses, fixed=fixed, points=np.array(pts), e_cam=np.array(ecam, np.int32), e_pt=np.array(ept, np.int32),
                          e_obs=np.ascontiguousarray(np.array(eobs, np.float64)), e_inv_sigma=np.array(eis, np.float64),
                          cam=np.asarray(cam5, np.float64))
                nposes, npts, erase, sums, rc = orc.local_ba(pr)
                assert rc == 0
                for c in range(nc):
                    R, t = synth.se3_exp(nposes[c])
                    frames[f0 + c]["Tcw"] = np.concatenate([R.reshape(-1), t])
                for j, m in enumerate(pid):
                    mp_pos[m] = npts[j].copy()
                for e in range(len(ecam) - 1, -1, -1):
                    if not erase[e]:
                        continue
                    m, qi = eref[e]
                    o = mp_obs[m][qi]
                    frames[o[0]]["mp"][o[1]] = -1
                    del mp_obs[m][qi]
                if log:
                    log(f"oracle local BA after frame {i}: {nc} frames, {len(pid)} points, {len(ecam)} edges, "
                        f"{sums[0].iterations} + {sums[1].iterations} iterations, {int(np.sum(erase))} erased")
        if i > 0:
            Tcl = _mul(frames[i]["Tcw"], _inv(frames[i - 1]["Tcw"])) if ok else EYE.copy()
        rec["ok"] = ok
        rec["map"] = len(mp_pos)
        info.append(rec)
    return [f["Tcw"] for f in frames], info
