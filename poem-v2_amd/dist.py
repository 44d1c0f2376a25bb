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


def force_init():
    """POEM_DIST_FORCE_INIT=1: build the process group even at WORLD_SIZE 1.  A one-rank RCCL communicator is legal, so a
    1-GPU box can execute the very branch an 8-GPU run takes (communicator creation with ``device_id=``, device-tensor
    all-reduces, barrier, destroy) -- tests/test_dist_gpu.py."""
    return os.environ.get("POEM_DIST_FORCE_INIT") == "1"


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* (torch.distributed.run contract): one process
    per GPU, communicator bound to the rank's device -- what the reference's ``setup_ddp`` does with
    ``init_process_group("nccl")`` + ``set_device(rank)`` + ``barrier()`` (scripts/eval.py:30-43 upstream)."""
    rank, local_rank, world = env_world()
    if (world > 1 or force_init()) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world > 1:
                raise RuntimeError("init_from_env: MASTER_PORT is not set (torch.distributed.run exports it); refusing to guess "
                                   "a rendezvous port for a multi-rank group")
            os.environ["MASTER_PORT"] = str(free_port())
        if backend == "nccl":
            if os.environ.get("POEM_SINGLE_DEVICE") == "1":
                local_rank = 0
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def free_port():
    """A TCP port free on 127.0.0.1 right now (the launchers and tests never hard-code one)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def active():
    """True when collectives run: a process group exists (any world size -- a forced one-rank group included)."""
    return dist.is_available() and dist.is_initialized()


def shard_range(n_items, rank, world):
    """Contiguous, balanced [lo, hi) slice of ``n_items`` for ``rank`` (first ``n_items % world`` ranks get one more)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_by_views(cam_view_num, rank, world):
    """Contiguous sample range whose cost is balanced across ranks (ragged batches).  Cost of a sample = 1 + 0.0127 N_i:
    the sampling stage is ~8 us per view next to ~630 us of per-sample decoder work (profiles/r04_step_timeline.txt).  Every
    boundary is the prefix sum NEAREST to r/world of the total (a first-not-below rule hands out 9 | 7 samples where 8 | 8 is
    closer: the slowest rank sets the step time).  Returns [lo, hi) over samples."""
    import numpy as np
    v = np.asarray(cam_view_num, dtype=np.int64)
    cost = v.astype(np.float64) * 0.0127 + 1.0
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    bounds = [0]
    for r in range(1, world):
        t = cum[-1] * r / world
        i = int(np.searchsorted(cum, t, side="left"))
        if i > 0 and abs(cum[i - 1] - t) <= abs(cum[min(i, len(v))] - t):
            i -= 1
        bounds.append(max(min(i, len(v)), bounds[-1]))
    bounds.append(len(v))
    return bounds[rank], bounds[rank + 1]


def all_reduce_sum_(t):
    if active():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def all_reduce_max_(t):
    if active():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def barrier():
    if active():
        dist.barrier()


def shutdown():
    """Tear the group down (``dist.destroy_process_group()``, scripts/eval.py:105 upstream); a no-op without one."""
    if active():
        dist.destroy_process_group()
