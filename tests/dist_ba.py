"""TEST-ONLY helpers (tests/test_dist_gloo.py, tests/test_gpu_ba.py): the sharding rule and a Python driver of the
split-phase entry points, kept as an executable statement of the protocol.  The product path is the sharded LM loop
INSIDE the C-ABI (vo_ba_set_shard + vo_ba_set_allreduce, then vo_ba_solve / vo_ba_local_ba: tests/dist_worker.py,
examples/rccl_sharded_ba.cpp, bench.py); nothing in the library, the shims or bench.py imports this module.

Multi-GPU local BA: points (with all their edges) are sharded across ranks, cameras are
replicated, and each LM iteration exchanges exactly two sum all-reduces over torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests):

  1. the packed reduced camera system  [ Y W^T | Hpp | gp | cost | per-rank gmax slots ]
  2. six scalars of the candidate evaluation (cost, model-change parts, norms)

Every control decision of the trust-region loop is a function of all-reduced values only, so
all ranks take identical decisions without further communication.
"""
from __future__ import annotations

import numpy as np

HUBER_MONO = float(np.sqrt(np.float32(5.991)))    # optimizer_ceres.cpp:532
HUBER_STEREO = float(np.sqrt(np.float32(7.815)))  # :533


def shard_of_point(point_index, world_size: int):
    """Ownership rule shared by the C-ABI (vo_ba_set_shard) and the drivers: p % world."""
    return np.asarray(point_index) % world_size


def shard_edge_mask(e_pt: np.ndarray, rank: int, world_size: int) -> np.ndarray:
    return (shard_of_point(e_pt, world_size) == rank).astype(np.uint8)


def allreduce_sum_(tensor, dist=None):
    """In-place sum all-reduce; a no-op for a single process."""
    if dist is None:
        import torch.distributed as dist  # noqa: PLC0415
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return tensor


def merge_owned_points(points: np.ndarray, rank: int, world_size: int, allreduce):
    """Every rank contributes the points it owns; the sum is the full, updated point array."""
    own = shard_of_point(np.arange(len(points)), world_size) == rank
    masked = np.where(own[:, None], points, 0.0)
    return allreduce(masked)


class ShardedBundleAdjuster:
    """Drives one BundleAdjuster shard per rank through the split-phase C-ABI."""

    def __init__(self, prob, rank: int, world_size: int):
        import torch
        from vo_slam_test_amd import _lib
        self.torch = torch
        self.rank, self.world = rank, world_size
        self.stream = torch.cuda.current_stream()
        self.ba = _lib.BundleAdjuster(prob, shard=rank, n_shards=world_size, stream=self.stream.cuda_stream)
        _, n_sys = self.ba.reduced_system()
        _, n_cost = self.ba.reduced_cost()
        self.sys_t = torch.zeros(n_sys, dtype=torch.float64, device="cuda")
        self.cost_t = torch.zeros(n_cost, dtype=torch.float64, device="cuda")
        self.ba.set_reduce_buffers(self.sys_t, self.cost_t)

    def close(self):
        self.ba.close()

    def _iterate(self, max_iterations):
        for _ in range(max_iterations):
            self.ba.linearize()
            allreduce_sum_(self.sys_t)
            self.ba.step()
            allreduce_sum_(self.cost_t)
            self.ba.update()

    def solve(self, huber_mono=0.0, huber_stereo=0.0, max_iterations=10, edge_active=None):
        self.ba.lm_begin(huber_mono, huber_stereo, max_iterations, edge_active)
        self._iterate(max_iterations)
        return self.ba.lm_end()

    def _sync_points(self):
        poses, pts = self.ba.state()
        t = self.torch.from_numpy(np.where((shard_of_point(np.arange(len(pts)), self.world) == self.rank)[:, None],
                                           pts, 0.0)).cuda()
        allreduce_sum_(t)
        full = t.cpu().numpy()
        self.ba.set_state(None, full)
        return poses, full

    def local_ba(self):
        """Optimizer::solveLocalBAPoseAndPoint schedule (optimizer_ceres.cpp:597-755), sharded."""
        s1 = self.solve(HUBER_MONO, HUBER_STEREO, 5)
        self._sync_points()
        self.ba.classify(False)
        self.ba.lm_begin_inliers(0.0, 0.0, 10)
        self._iterate(10)
        s2 = self.ba.lm_end()
        poses, pts = self._sync_points()
        self.ba.classify(True)
        return poses, pts, self.ba.edge_outliers(), (s1, s2)
