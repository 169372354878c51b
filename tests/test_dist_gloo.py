"""World-size-2 test of the N>1 path on CPU (gloo): the sharding rule and the all-reduce payload
handling of tests/dist_ba.py, with the CPU oracle standing in for the per-shard HIP
linearisation (the oracle is the checker here, never the product path)."""
import os
import pathlib
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import torch
    import torch.distributed as dist
    import oracle_lib as orc
    import dist_ba
    from vo_slam_test_amd import synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pr = synth.make_lba_problem(12, n_kf=4, n_pts=120, n_fixed=1)
    mask = dist_ba.shard_edge_mask(pr["e_pt"], rank, world)
    S, b, cost, nf = orc.ba_schur(pr, active=mask, point_damping=1e-3)
    payload = torch.from_numpy(np.concatenate([S.ravel(), b, [cost]]))
    dist_ba.allreduce_sum_(payload, dist)
    # every rank updates only the points it owns; the merge restores the full array on all ranks
    pts = pr["points"].copy()
    own = dist_ba.shard_of_point(np.arange(len(pts)), world) == rank
    pts[own] += 1.0 + rank
    merged = dist_ba.merge_owned_points(
        pts, rank, world, lambda a: dist_ba.allreduce_sum_(torch.from_numpy(a.copy()), dist).numpy())
    q.put((rank, payload.numpy(), merged, int(mask.sum())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_gloo_allreduce_of_shard_partials(orc):
    import torch.multiprocessing as mp
    import dist_ba
    from vo_slam_test_amd import synth
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=150) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    pr = synth.make_lba_problem(12, n_kf=4, n_pts=120, n_fixed=1)
    S, b, cost, nf = orc.ba_schur(pr, point_damping=1e-3)
    ref = np.concatenate([S.ravel(), b, [cost]])
    assert np.array_equal(res[0][1], res[1][1])              # all ranks hold identical reduced systems
    assert np.abs(res[0][1] - ref).max() <= 1e-9 * np.abs(ref).max()
    assert res[0][3] + res[1][3] == len(pr["e_pt"])           # every edge belongs to exactly one shard
    expect = pr["points"].copy()
    expect[0::2] += 1.0
    expect[1::2] += 2.0
    assert np.array_equal(res[0][2], res[1][2]) and np.allclose(res[0][2], expect)


def test_sharding_rule():
    import dist_ba
    e_pt = np.arange(20) % 7
    masks = [dist_ba.shard_edge_mask(e_pt, r, 3) for r in range(3)]
    assert (np.sum(masks, 0) == 1).all()
    assert np.array_equal(dist_ba.shard_of_point(np.arange(6), 1), np.zeros(6, int))


def test_collective_watchdog_ends_a_stuck_rank(tmp_path):
    """a rank whose peer never arrives must END with a non-zero status, not hang.  Two gloo ranks are
    started, rank 1 exits before its first barrier; rank 0 waits in a barrier guarded by bench.py's OWN watchdog
    (vo_slam_test_amd.watchdog.CollectiveWatchdog -- the class bench.py arms around its barriers and around the all-reduce
    callback of the sharded LM loop), which gives up within the deadline."""
    script = tmp_path / "stuck.py"
    script.write_text(
        "import os, sys, datetime\n"
        f"sys.path.insert(0, {str(ROOT)!r})\n"
        "import torch.distributed as dist\n"
        "from vo_slam_test_amd.watchdog import CollectiveWatchdog\n"
        "rank = int(os.environ['RANK'])\n"
        "dist.init_process_group('gloo', timeout=datetime.timedelta(seconds=8))\n"
        "if rank == 1:\n"
        "    os._exit(0)\n"
        "wd = CollectiveWatchdog(10.0, rank, poll=0.5).start()\n"
        "wd.arm('barrier (waiting)')\n"
        "try:\n"
        "    dist.barrier()\n"
        "except Exception as e:\n"
        "    sys.stderr.write('barrier failed: %r\\n' % (e,))\n"
        "    os._exit(71)\n"
        "wd.disarm('barrier')\n"
        "os._exit(0)\n")
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    procs = []
    for rk in (0, 1):
        env = dict(os.environ, RANK=str(rk), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stderr=subprocess.PIPE))
    rcs = [p.wait(timeout=60) for p in procs]
    assert rcs[1] == 0 and rcs[0] in (70, 71), rcs  # the stuck rank ended by its deadline, with a non-zero status


def test_watchdog_fires_only_while_armed():
    """the deadline runs between arm() and disarm() only, mark() restarts it, and expiry calls the exit function with status 70
    (here a recording stand-in for os._exit)"""
    import time
    from vo_slam_test_amd.watchdog import CollectiveWatchdog
    fired = []
    wd = CollectiveWatchdog(0.3, rank=3, poll=0.05, exit_fn=fired.append).start()
    time.sleep(0.5)
    assert not fired and not wd.expired()          # never armed: idle time does not count
    wd.arm("all-reduce of 8 doubles (waiting)")
    time.sleep(0.15)
    wd.mark("still waiting")                        # progress restarts the clock
    time.sleep(0.2)
    assert not fired
    wd.disarm("all-reduce of 8 doubles")
    time.sleep(0.5)
    assert not fired                                # disarmed: a long compute phase between collectives is fine
    wd.arm("barrier (waiting)")
    time.sleep(0.6)
    assert fired == [70] and wd.what == "barrier (waiting)"


def test_bench_uses_the_importable_watchdog():
    """bench.py arms the watchdog around its barriers AND around the all-reduce callback of the sharded LM loop (the
    callback only marked progress, so a collective that hung inside vo_ba_solve was not ended by --collective-timeout)"""
    src = (ROOT / "bench.py").read_text()
    assert "from vo_slam_test_amd.watchdog import CollectiveWatchdog" in src
    cb = src[src.index("def _allreduce("):]
    cb = cb[:cb.index("return 0")]
    assert "wd.arm(" in cb and "wd.disarm(" in cb and cb.index("wd.arm(") < cb.index("dist.all_reduce") < cb.index("wd.disarm(")
