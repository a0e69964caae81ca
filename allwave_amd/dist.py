"""Multi-GPU plumbing: one process per GPU, pairs sharded with no collective on the data path
(SURVEY.md 8e); torch.distributed (RCCL on GPUs, gloo on CPUs) only carries the timing barrier and
the final reduction of counters."""
import os

import numpy as np


def env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_pairs(pairs, rank, world):
    """Strided shard of the pair list: rank r takes pairs r, r+world, ... (keeps per-rank cost even for a
    row-major all-pairs list; every pair lands on exactly one rank)."""
    return np.ascontiguousarray(pairs[rank::world])


def init(backend=None, device=None):
    rank, local_rank, world = env()
    if world <= 1:
        return None
    import torch
    import torch.distributed as dist
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank if device is None else device)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend=backend)
    return dist


def barrier(dist, local_rank=0):
    import torch
    if dist is not None:
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[local_rank])
        else:
            dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def reduce_max_sum(dist, elapsed, sums):
    """max over ranks of `elapsed`, sum over ranks of each value in `sums`."""
    import torch
    if dist is None:
        return float(elapsed), [float(s) for s in sums]
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(elapsed)], dtype=torch.float64, device=dev)
    s = torch.tensor([float(x) for x in sums], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return float(t[0]), [float(x) for x in s]
