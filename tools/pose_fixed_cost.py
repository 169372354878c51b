"""Developer tool: ONE pose-only problem on the device with 1000 / 500 / 250 / 125 / 64 of its observations, per block width
(vo_set_option(VO_OPT_POSE_BLOCK, 64 / 128 / 256) in a child process): time per launch and LM iterations -- how much of
an LM iteration is the per-observation work and how much the fixed part (reduction, 6 x 6 solve, exp / log, tests)."""
import ctypes, os, pathlib, subprocess, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    from vo_slam_test_amd import _lib as vo, synth
    vo.set_option("pose_block", int(sys.argv[2]))
    pr = synth.make_pose_problem(3)
    cs = torch.cuda.current_stream()
    for n in (1000, 500, 250, 125, 64):
        offs = torch.tensor([0, n], dtype=torch.int32).cuda()
        d_pts, d_obs, d_isg = (torch.from_numpy(np.ascontiguousarray(pr[k][:n])).cuda() for k in ("pts", "obs", "inv_sigma"))
        d_cam = torch.from_numpy(np.ascontiguousarray(pr["cam"], np.float64)).cuda()
        pose0 = torch.from_numpy(pr["pose0"][None]).cuda()
        d_pose, d_out, d_inl = pose0.clone(), torch.zeros(n, dtype=torch.uint8, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
        d_sum = torch.zeros(2 * ctypes.sizeof(vo.LmSummary), dtype=torch.uint8, device="cuda")
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ts = []
        for rep in range(30):
            d_pose.copy_(pose0)
            ev[0].record(cs)
            vo.check(vo.lib().vo_pose_only_solve_dev(1, vo._p(offs), n, vo._p(d_pts), vo._p(d_obs), vo._p(d_isg), vo._p(d_cam), vo._p(d_pose),
                                                     vo._p(d_out), vo._p(d_inl), vo._p(d_sum), ctypes.c_void_p(cs.cuda_stream)))
            ev[1].record(cs)
            torch.cuda.synchronize()
            ts.append(ev[0].elapsed_time(ev[1]))
        raw = np.ascontiguousarray(d_sum.cpu().numpy())
        sums = (vo.LmSummary * 2).from_buffer_copy(raw.tobytes())
        its = sums[0].iterations + sums[1].iterations
        t = float(np.median(ts[5:])) * 1e3
        print(f"   n = {n:4d}: {t:7.1f} us per launch, {its} LM iterations -> {t / max(its, 1):5.2f} us per iteration")
    sys.exit(0)
for bw in (64, 128, 256):
    print("pose block =", bw)
    r = subprocess.run([sys.executable, __file__, "child", str(bw)], capture_output=True, text=True)
    print(r.stdout.rstrip() or r.stderr[-1500:])
