"""Multi-GPU plumbing: one process per GPU, pairs sharded with no collective on the data path
(SURVEY.md 8e); torch.distributed (RCCL on GPUs, gloo on CPUs) only carries the timing barrier, the
final reduction of counters and -- optionally -- the gather of each rank's PAF text on rank 0."""
import os

import numpy as np


def env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_pairs(pairs, rank, world, lens=None, scores=None):
    """This rank's part of the pair list (every pair lands on exactly one rank, list order kept).
    With sequence lengths and scores: the cost-balanced LPT partition of the host library
    (planner::assign_shards_lpt -- what `allwave_hip --shard R/N` uses), which keeps the predicted cost
    of the shards even when pair costs span orders of magnitude (config 5).  Without: strided
    (r, r + world, ...), which is what LPT gives for equal-cost lists (configs 2 and 3)."""
    if world <= 1:
        return np.ascontiguousarray(pairs)
    if lens is None or scores is None:  # (the cost model needs both: without the penalties in use the strided shard is the fallback)
        return np.ascontiguousarray(pairs[rank::world])
    from . import host as H
    shard, _ = H.shard_assignment(pairs, lens, ",".join(str(int(v)) for v in scores), world)
    return np.ascontiguousarray(np.asarray(pairs)[shard == rank])


def init(backend=None, device=None):
    rank, local_rank, world = env()
    if world <= 1:
        return None
    import torch
    import torch.distributed as dist
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        dev = local_rank if device is None else int(device)
        torch.cuda.set_device(dev)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev))
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


def gather_floats(dist, value):
    """Every rank's `value`, in rank order, on every rank (per-rank kernel times of the bench line)."""
    if dist is None:
        return [float(value)]
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    out = torch.zeros(dist.get_world_size(), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out, torch.tensor([float(value)], dtype=torch.float64, device=dev))
    return [float(x) for x in out]


def gather_bytes(dist, payload, local_rank=0):
    """Optional result gather (BASELINE north_star: "RCCL over xGMI only for optional result gather"):
    every rank contributes a byte string (its PAF text); rank 0 gets the list in rank order, other
    ranks get None.  Two collectives: an all-gather of the sizes, then one of the padded payloads."""
    if dist is None:
        return [bytes(payload)]
    import torch
    dev = torch.device("cuda", local_rank) if dist.get_backend() == "nccl" else torch.device("cpu")
    world = dist.get_world_size()
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, torch.tensor([len(payload)], dtype=torch.int64, device=dev))
    cap = max(int(sizes.max()), 1)
    mine = torch.zeros(cap, dtype=torch.uint8, device=dev)
    if len(payload):
        mine[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(dev)
    allb = torch.empty(world * cap, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(allb, mine)
    if dist.get_rank() != 0:
        return None
    host = allb.cpu().numpy()
    return [host[r * cap:r * cap + int(sizes[r])].tobytes() for r in range(world)]
