"""Data-parallel glue: one process per GPU, samples sharded by contiguous ranges, no data-path collective.  The only
collective of the inference path is one tiny all-reduce of the metric sums ("C3") -- RCCL over
xGMI when the process group backend is "nccl" (== RCCL on ROCm), gloo in the CPU tests.

The reference wraps the whole model in DistributedDataParallel (scripts/eval.py:78-79 upstream) and evaluates on
rank 0 only; the head has no cross-sample operation, so sharding samples is exact."""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* (torch.distributed.run contract)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(n_items, rank, world):
    """Contiguous, balanced [lo, hi) slice of ``n_items`` for ``rank`` (first ``n_items % world`` ranks get one more)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_by_views(cam_view_num, rank, world):
    """Contiguous sample range whose summed view count is balanced across ranks (ragged batches: the sampling
    stage costs ~N_i, the decoder is per-sample constant).  Returns [lo, hi) over samples."""
    import numpy as np
    v = np.asarray(cam_view_num, dtype=np.int64)
    cost = v.astype(np.float64) * 0.05 + 1.0          # merge ~5 % of a sample per 8 views
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    bounds = [int(np.searchsorted(cum, cum[-1] * r / world, side="left")) for r in range(world)] + [len(v)]
    bounds[0] = 0
    for i in range(1, len(bounds)):
        bounds[i] = max(bounds[i], bounds[i - 1])
    return bounds[rank], bounds[rank + 1]


def all_reduce_sum_(t):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def all_reduce_max_(t):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
