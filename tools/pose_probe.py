"""Developer tool: device-resident timing of the batched pose-only solve (1024 frames x 1000 observations, one launch)
and its parity with the oracle on the first problems.  usage: python tools/pose_probe.py [n_frames]"""
import ctypes, pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent / "tests"))
import numpy as np, torch  # noqa: E402
from vo_slam_test_amd import _lib as vo, synth  # noqa: E402

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
base = [synth.make_pose_problem(i) for i in range(64)]
probs = (base * ((nf + 63) // 64))[:nf]
offs = np.arange(len(probs) + 1, dtype=np.int32) * 1000
cat = lambda k: torch.from_numpy(np.ascontiguousarray(np.concatenate([pr[k] for pr in probs]))).cuda()
d_off, d_pts, d_obs, d_isg = torch.from_numpy(offs).cuda(), cat("pts"), cat("obs"), cat("inv_sigma")
d_cam = torch.from_numpy(np.ascontiguousarray(probs[0]["cam"], np.float64)).cuda()
pose0 = torch.from_numpy(np.stack([pr["pose0"] for pr in probs])).cuda()
d_pose, d_out = pose0.clone(), torch.zeros(len(probs) * 1000, dtype=torch.uint8, device="cuda")
d_inl = torch.zeros(len(probs), dtype=torch.int32, device="cuda")
cs = torch.cuda.current_stream()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ts = []
for rep in range(6):
    d_pose.copy_(pose0)
    ev[0].record(cs)
    vo.check(vo.lib().vo_pose_only_solve_dev(len(probs), vo._p(d_off), 1000, vo._p(d_pts), vo._p(d_obs), vo._p(d_isg),
                                             vo._p(d_cam), vo._p(d_pose), vo._p(d_out), vo._p(d_inl), None,
                                             ctypes.c_void_p(cs.cuda_stream)), "vo_pose_only_solve_dev")
    ev[1].record(cs)
    torch.cuda.synchronize()
    ts.append(ev[0].elapsed_time(ev[1]))
print("ms per launch (%d frames):" % nf, [round(t, 4) for t in ts], "median %.4f" % float(np.median(ts[1:])))
try:
    import oracle_lib as orc
    poses, outl, inl = d_pose.cpu().numpy(), d_out.cpu().numpy().reshape(len(probs), 1000), d_inl.cpu().numpy()
    worst = 0.0
    for i in range(min(16, len(base))):
        opose, ooutl, oninl, _, _ = orc.pose_only(base[i])
        worst = max(worst, float(np.abs(poses[i] - opose).max()))
        assert inl[i] == oninl and np.array_equal(outl[i], ooutl), ("mask / inlier mismatch", i)
    print("parity with the oracle on 16 problems: max |pose diff| %.3e, masks identical" % worst)
except ImportError as e:  # noqa: BLE001
    print("oracle not available:", e)
