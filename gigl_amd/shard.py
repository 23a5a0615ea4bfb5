"""Ray-batch sharding for one process per GPU.

Rays are independent, so the path shards with no data-path collective: the tree is replicated,
rank r traces the contiguous ray range shard_range(n, world, r), and the only exchange is the
gather of the hit buffers to rank 0 (16 B per closest-hit ray, 1 bit per any-hit ray).  The
gather is issued chunk by chunk behind the traversal so the xGMI transfer of chunk i overlaps
the traversal of chunk i+1 (bench.py).
"""
import numpy as np


def shard_range(n, world, rank):
    """Contiguous [lo, hi) of rank `rank` of `world`; sizes differ by at most one."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, rem = divmod(int(n), world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def chunk_bounds(n, chunks):
    chunks = max(1, min(int(chunks), max(int(n), 1)))
    return [(i * int(n)) // chunks for i in range(chunks + 1)]


def gather_hits(local_hits, n_total, world, rank, dist, dst=0):
    """Gathers per-rank hit-record arrays (any shard sizes) into ray order on rank `dst`.

    local_hits: torch tensor [m, 4] int32 on the backend's device.  Returns the [n_total, 4]
    tensor on dst, None elsewhere.  Shards are padded to the largest shard for the collective.
    """
    import torch
    sizes = [shard_range(n_total, world, r) for r in range(world)]
    m_max = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((m_max, 4), dtype=local_hits.dtype, device=local_hits.device)
    pad[: local_hits.shape[0]] = local_hits
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    out = torch.empty((n_total, 4), dtype=local_hits.dtype, device=local_hits.device)
    for r, (lo, hi) in enumerate(sizes):
        out[lo:hi] = bufs[r][: hi - lo]
    return out
