"""Developer tool: build ablated variants of k_resize4 (VO_RZ_ABLATE=1 no arithmetic, 2 no global loads,
3 no stores) into tools/_stamp/ and time the pyramid stage with each:  python tools/rz_ablate.py build|run"""
import json
import os
import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
OUT = ROOT / "tools" / "_stamp"


def build():
    OUT.mkdir(exist_ok=True)
    srcs = [ROOT / "vo_slam_test_amd" / "csrc" / n for n in ("vo_common.hip", "orb.hip", "match.hip", "ba.hip")]
    for k in (0, 1, 2, 3):
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", f"-DVO_RZ_ABLATE={k}",
               "-ffp-contract=off", "-Wno-unused-function", *map(str, srcs), "-o", str(OUT / f"libvo_rz{k}.so")]
        subprocess.run(cmd, check=True)
        print("built variant", k)


def run():
    for k in (0, 1, 2, 3):
        env = dict(os.environ, VO_HIP_LIB=str(OUT / f"libvo_rz{k}.so"))
        r = subprocess.run([sys.executable, __file__, "one"], env=env, capture_output=True, text=True)
        print("ablate", k, r.stdout.strip() or r.stderr[-400:])


def one():
    sys.path.insert(0, str(ROOT))
    import torch
    from vo_slam_test_amd import _lib as vo, synth
    B = 256
    frames = torch.from_numpy(synth.make_frames(8)).cuda().repeat(B // 8, 1, 1).contiguous()
    ext = vo.OrbExtractor(1000, 1.2, 8, 20, 7)
    cap = ext.max_keypoints()
    kps = torch.zeros((B, cap, 28), dtype=torch.uint8, device="cuda")
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    for _ in range(3):
        ext.extract_batch_dev(frames, kps, desc, cnt)
    ext.sync()
    ext.set_timing(True)
    for _ in range(10):
        ext.extract_batch_dev(frames, kps, desc, cnt)
    ext.sync()
    ms, n = ext.get_timing()
    print({k: round(v / n, 4) for k, v in ms.items()})


if __name__ == "__main__":
    {"build": build, "run": run, "one": one}[sys.argv[1]]()
