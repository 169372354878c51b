"""Developer tool: per-stage extraction times for the library named by VO_HIP_LIB, without bench.py's
sanity asserts (usable with ablated / early-exit developer builds)."""
import pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from vo_slam_test_amd import _lib as vo, synth  # noqa: E402

import os
B = int(os.environ.get('VO_EXT_B', '1024'))
stream = torch.cuda.Stream()
ext = vo.OrbExtractor(1000, 1.2, 8, 20, 7)
ext.set_stream(stream.cuda_stream)
if hasattr(ext, 'set_fused'):
    ext.set_fused(int(os.environ.get('VO_EXT_FUSED', '0')))  # 0: the three separate kernels per level
if hasattr(ext, 'set_describe_blur'):
    ext.set_describe_blur(int(os.environ.get('VO_EXT_DESCBLUR', '0')))  # 0: the descriptor kernel blurs its windows itself
cap = ext.max_keypoints()
with torch.cuda.stream(stream):
    frames = torch.from_numpy(synth.make_frames(32)).cuda().repeat(B // 32, 1, 1).contiguous()
    kps = torch.zeros((B, cap, 28), dtype=torch.uint8, device="cuda")
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    for _ in range(3):
        ext.extract_batch_dev(frames, kps, desc, cnt)
    torch.cuda.synchronize()
    ext.set_timing(True)
    for _ in range(int(os.environ.get('VO_EXT_REPS', '10'))):
        ext.extract_batch_dev(frames, kps, desc, cnt)
    torch.cuda.synchronize()
ms, n = ext.get_timing()
print({k: round(v / max(n, 1), 4) for k, v in ms.items()})
