"""Read sharding across the GPUs of one node.

Reads are independent, so there is NO collective on the hot path: each rank decodes its own shard.
The only exchange is one all-reduce of a few uint64 counters at the end of a run (RCCL when the
process group is `nccl`, gloo in the CPU tests).
"""
import numpy as np


def lpt_partition(lengths, world_size):
    """Longest-processing-time greedy: sort reads by event count (descending, stable) and give each
    to the least-loaded rank.  Returns a list of index arrays (one per rank, ascending read id)."""
    lengths = np.asarray(lengths, dtype=np.int64)
    order = np.argsort(-lengths, kind="stable")
    load = np.zeros(world_size, dtype=np.int64)
    shards = [[] for _ in range(world_size)]
    if len(lengths) and np.all(lengths == lengths[0]):
        # equal lengths (the C4 benchmark shape): contiguous slices, no sort needed
        bounds = np.linspace(0, len(lengths), world_size + 1).astype(np.int64)
        return [np.arange(bounds[r], bounds[r + 1], dtype=np.int64) for r in range(world_size)]
    for i in order:
        r = int(np.argmin(load))
        shards[r].append(int(i))
        load[r] += lengths[i]
    return [np.array(sorted(s), dtype=np.int64) for s in shards]


def gather_counters(local_counters, device=None):
    """Sum a small vector of uint64 counters over all ranks (one all-reduce).  Works without an
    initialised process group (returns the input)."""
    import torch
    import torch.distributed as dist

    t = torch.as_tensor(np.asarray(local_counters, dtype=np.int64))
    if not (dist.is_available() and dist.is_initialized()):
        return t.numpy().astype(np.uint64)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().astype(np.uint64)


def max_over_ranks(value, device=None):
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_per_rank(values, device=None):
    """One all-gather of a short float64 vector per rank -> array [world, len(values)] on every rank (row r = rank r's
    values).  Without an initialised process group: one row."""
    import torch
    import torch.distributed as dist

    t = torch.tensor([float(v) for v in values], dtype=torch.float64)
    if not (dist.is_available() and dist.is_initialized()):
        return t.numpy().reshape(1, -1)
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out).cpu().numpy()
