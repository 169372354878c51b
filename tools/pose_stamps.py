"""Developer tool (GPU box): where the cycles of a pose-only LM iteration go.  Needs a -DVO_POSE_STAMPS build of csrc/ba.hip
(tools/build_variant_src.sh stamps ba vo_slam_test_amd/csrc/ba.hip -DVO_POSE_STAMPS; VO_HIP_LIB=.../libvo_stamps.so): the
summary fields then carry shader-clock cycles per phase, summed over the iterations of a round.
usage: VO_HIP_LIB=vo_slam_test_amd/_variants/libvo_stamps.so python tools/pose_stamps.py [n_frames]"""
import ctypes, pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np, torch  # noqa: E402
from vo_slam_test_amd import _lib as vo, synth  # noqa: E402

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 1
base = [synth.make_pose_problem(i) for i in range(min(nf, 64))]
probs = (base * ((nf + 63) // 64))[:nf]
offs = np.arange(len(probs) + 1, dtype=np.int32) * 1000
cat = lambda k: torch.from_numpy(np.ascontiguousarray(np.concatenate([pr[k] for pr in probs]))).cuda()
d_off, d_pts, d_obs, d_isg = torch.from_numpy(offs).cuda(), cat("pts"), cat("obs"), cat("inv_sigma")
d_cam = torch.from_numpy(np.ascontiguousarray(probs[0]["cam"], np.float64)).cuda()
pose0 = torch.from_numpy(np.stack([pr["pose0"] for pr in probs])).cuda()
d_pose, d_out = pose0.clone(), torch.zeros(len(probs) * 1000, dtype=torch.uint8, device="cuda")
d_inl = torch.zeros(len(probs), dtype=torch.int32, device="cuda")
d_sum = torch.zeros(2 * len(probs) * ctypes.sizeof(vo.LmSummary), dtype=torch.uint8, device="cuda")
cs = torch.cuda.current_stream()
for rep in range(3):
    d_pose.copy_(pose0)
    vo.check(vo.lib().vo_pose_only_solve_dev(len(probs), vo._p(d_off), 1000, vo._p(d_pts), vo._p(d_obs), vo._p(d_isg), vo._p(d_cam),
                                             vo._p(d_pose), vo._p(d_out), vo._p(d_inl), vo._p(d_sum), ctypes.c_void_p(cs.cuda_stream)))
    torch.cuda.synchronize()
sums = (vo.LmSummary * (2 * len(probs))).from_buffer_copy(np.ascontiguousarray(d_sum.cpu().numpy()).tobytes())
tot = np.zeros(5)
its = 0
for s in sums:
    tot += np.array([s.initial_cost, s.final_cost, s.final_radius, s.reserved, s.accepted], float)
    its += s.iterations
names = ["solve (read sums, scale, damp, 6x6 Cholesky, model)", "plus (exp, compose, log)", "pass over the observations",
         "28-sum reduction + read-back", "tests, radius update"]
print(f"{nf} frames x 1000 observations, {its} LM iterations in all; shader-clock cycles per iteration:")
for n, t in zip(names, tot):
    print(f"  {n:55s} {t / its:9.0f}")
print(f"  {'sum':55s} {tot.sum() / its:9.0f}")
