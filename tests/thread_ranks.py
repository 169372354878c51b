"""N shards of a sharded bundle adjustment as N threads of ONE process on one GPU, each with its own handle and stream;
the all-reduce callback meets at a barrier and sums the ranks' buffers in rank order (what RCCL / gloo do between
processes).  ctypes releases the GIL for the duration of vo_ba_solve and re-acquires it in the callback, so the ranks
really interleave.  Used by the GPU tests (world sizes a one-GPU box cannot host as processes) and by tools/gba_seg_run.py
(one process = something rocprofv3 can trace)."""
import threading

import numpy as np


class _DevView:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def run_ranks(vo, prob, world, solve, streams=None, serial=False, options=None):
    """solve(handle, rank) -> result, on `world` sharded handles of `prob`; returns (results, stats) with
    stats = dict(calls per rank, payload sizes in doubles per call of rank 0, and -- serial=True: the ranks take turns
    between collectives, so that a rank's GPU work runs alone -- per rank the wall time of every stretch between two
    collectives, launch overheads included: `stretch_ms`)"""
    import time
    import torch
    barrier = threading.Barrier(world)
    slots = [None] * world
    stats = {"calls": [0] * world, "sizes": [], "stretch_ms": [[] for _ in range(world)], "handshake": [False] * world}
    results, errors = [None] * world, []
    streams = streams or [torch.cuda.Stream() for _ in range(world)]
    arrived = [threading.Semaphore(0) for _ in range(world)]  # rank r has finished the stretch before its next collective
    released = [0.0] * world

    def worker(rank):
        try:
            def allreduce(ptr, n, stream):
                handshake = int(n) == 4 and stats["calls"][rank] == 0 and not stats["handshake"][rank]
                if handshake:   # the protocol handshake of a sharded handle's first use (4 doubles): not part of the LM loop's schedule
                    stats["handshake"][rank] = True
                else:
                    stats["calls"][rank] += 1
                    if rank == 0:
                        stats["sizes"].append(int(n))
                with torch.cuda.stream(streams[rank]):
                    t = torch.as_tensor(_DevView(ptr, n), device="cuda")
                    streams[rank].synchronize()
                    if serial:
                        stats["stretch_ms"][rank].append((time.perf_counter() - released[rank]) * 1e3)
                        arrived[rank].release()
                    slots[rank] = t
                    barrier.wait()
                    total = slots[0].clone()
                    for r in range(1, world):  # rank order: the same sum on every rank
                        total += slots[r]
                    streams[rank].synchronize()
                    barrier.wait()  # everybody has read every slot
                    t.copy_(total)
                    streams[rank].synchronize()
                if serial:
                    if rank > 0:
                        arrived[rank - 1].acquire()  # my turn once the rank before me is through its next stretch
                    released[rank] = time.perf_counter()
                return 0

            with torch.cuda.stream(streams[rank]):
                h = vo.BundleAdjuster(prob, shard=rank, n_shards=world, stream=streams[rank].cuda_stream, options=options)
                h.set_allreduce(allreduce)
                if serial:
                    if rank > 0:
                        arrived[rank - 1].acquire()
                    released[rank] = time.perf_counter()
                results[rank] = solve(h, rank)
                if serial:
                    arrived[rank].release()
                h.close()
        except BaseException as e:  # noqa: BLE001 -- a rank that dies must not leave the others at the barrier
            errors.append((rank, e))
            barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0][1]
    return results, stats
