"""Developer tool (GPU box): extract + all-pairs Hamming (BASELINE configs[1], 1024 frames per step) with the matching of batch
i launched on a SECOND stream from the extractor's stage hook of batch i + 1 -- i.e. fenced behind stage `after` of the next
extraction (0: behind its pyramid, next to its FAST; 1: behind FAST; ...; -1: no fence, launched before the next extraction) --
against the serial order.  Descriptors are double-buffered.  usage: python tools/ebm_fenced_probe.py [after=-1,0,1,2] [reps=3]"""
import pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np, torch  # noqa: E402
from vo_slam_test_amd import _lib as vo, synth  # noqa: E402

kw = dict(a.split("=") for a in sys.argv[1:])
afters = [int(x) for x in kw.get("after", "-1,0,1,2").split(",")]
B, NM = 1024, 1000
SB8 = 5123128 + 2064000
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ext = vo.OrbExtractor(1000, 1.2, 8, 20, 7)
ext.set_stream(s1.cuda_stream)
cap = ext.max_keypoints()
with torch.cuda.stream(s1):
    frames = torch.from_numpy(synth.make_frames(32)).cuda().repeat(B // 32, 1, 1).contiguous()
    kps = torch.zeros((B, cap, 28), dtype=torch.uint8, device="cuda")
    desc = [torch.zeros((B + 1, cap, 32), dtype=torch.uint8, device="cuda") for _ in range(2)]
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    dmat = torch.zeros((B, NM, NM), dtype=torch.int16, device="cuda")
torch.cuda.synchronize()


def serial(n):
    for i in range(n):
        d = desc[i & 1]
        with torch.cuda.stream(s1):
            ext.extract_batch_dev(frames, kps, d[:B], cnt)
            d[B].copy_(d[0])
            vo.hamming_matrix_batch_dev(d[:B, :NM], d[1:, :NM], dmat, stream=s1.cuda_stream)


def fenced(n, after):
    pending = {"d": None, "ev": None}
    done_prev = None  # event: the matching that read buffer (i & 1) two steps ago has finished

    def launch_match():
        d = pending["d"]
        if d is None:
            return
        pending["d"] = None
        vo.hamming_matrix_batch_dev(d[:B, :NM], d[1:, :NM], dmat, stream=s2.cuda_stream)

    def hook(stage, stream):
        if stage == after and pending["d"] is not None:
            e = torch.cuda.Event()
            e.record(s1)
            s2.wait_event(e)      # behind this stage of the running extraction ...
            s2.wait_event(pending["ev"])  # ... and behind the descriptors it reads
            launch_match()

    ext.set_stage_hook(hook if after >= 0 else None)
    match_done = [None, None]
    for i in range(n):
        d = desc[i & 1]
        if match_done[i & 1] is not None:
            s1.wait_event(match_done[i & 1])  # the buffer this extraction overwrites has been read
        if after < 0 and pending["d"] is not None:
            s2.wait_event(pending["ev"])
            launch_match()
        with torch.cuda.stream(s1):
            ext.extract_batch_dev(frames, kps, d[:B], cnt)
            d[B].copy_(d[0])
        if pending["d"] is not None:  # (a hook stage that did not fire)
            s2.wait_event(pending["ev"])
            launch_match()
        ev = torch.cuda.Event()
        ev.record(s1)
        pending["d"], pending["ev"] = d, ev
        md = torch.cuda.Event()
        match_done[i & 1] = md
        # the matching of THIS batch is launched during the next step; its completion event is recorded then
        def rec(md=md):
            md.record(s2)
        pending["rec"] = rec
        # record completion of the previously launched matching
    s2.wait_event(pending["ev"])
    launch_match()
    ext.set_stage_hook(None)


def fenced_simple(n, after):
    """matching of batch i on s2 from the hook of extraction i + 1; buffer reuse guarded by events"""
    state = {"d": None, "ready": None}
    free_ev = [None, None]

    def launch():
        d = state["d"]
        state["d"] = None
        s2.wait_event(state["ready"])
        vo.hamming_matrix_batch_dev(d[:B, :NM], d[1:, :NM], dmat, stream=s2.cuda_stream)
        e = torch.cuda.Event()
        e.record(s2)
        free_ev[state["slot"]] = e

    def hook(stage, stream):
        if stage == after and state["d"] is not None:
            e = torch.cuda.Event()
            e.record(s1)
            s2.wait_event(e)
            launch()

    ext.set_stage_hook(hook if after >= 0 else None)
    for i in range(n):
        slot = i & 1
        d = desc[slot]
        if free_ev[slot] is not None:
            s1.wait_event(free_ev[slot])
        if after < 0 and state["d"] is not None:
            launch()
        with torch.cuda.stream(s1):
            ext.extract_batch_dev(frames, kps, d[:B], cnt)
            d[B].copy_(d[0])
        if state["d"] is not None:
            launch()
        r = torch.cuda.Event()
        r.record(s1)
        state["d"], state["ready"], state["slot"] = d, r, slot
    launch()
    ext.set_stage_hook(None)


def timed(f, *a, n=10):
    f(3, *a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f(n, *a)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(int(kw.get("reps", "3"))):
    t = timed(serial)
    line = f"serial {t:.3f} ms = {SB8 * B / t / 1e6 / 8000 * 100:.1f} %"
    for a in afters:
        t = timed(fenced_simple, a)
        line += f"  after={a}: {t:.3f} ms = {SB8 * B / t / 1e6 / 8000 * 100:.1f} %"
    print(line, flush=True)
