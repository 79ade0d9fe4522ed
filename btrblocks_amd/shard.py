"""Sharding of a column's blocks over the GPUs of one node.

Blocks (<= 65536 rows each) are independent, so a column shards as contiguous block ranges
balanced by decoded bytes; no rank needs data from another one. The only exchange a consumer may
want is the per-shard row/byte offsets of the concatenated column, an all-gather of two integers.
"""
import numpy as np


def shard_ranges(decoded_bytes, world_size):
    """[(first_block, end_block)] per rank: contiguous ranges with near-equal decoded bytes."""
    sizes = np.asarray(decoded_bytes, dtype=np.int64)
    total = int(sizes.sum())
    cum = np.concatenate([[0], np.cumsum(sizes)])
    bounds = [0]
    for r in range(1, world_size):
        target = total * r // world_size
        bounds.append(int(np.searchsorted(cum, target, side="left")))
    bounds.append(len(sizes))
    bounds = [max(bounds[i], bounds[i - 1]) if i else bounds[i] for i in range(len(bounds))]
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


def exchange_offsets(rows, nbytes, group=None):
    """All-gathers (rows, bytes) of every rank's shard; returns this rank's (row offset, byte offset)
    in the concatenated column and the totals. Works on any torch.distributed backend."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    mine = torch.tensor([rows, nbytes], dtype=torch.int64, device=dev)
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    table = torch.stack(gathered).cpu().numpy()
    before = table[:rank].sum(axis=0) if rank else np.zeros(2, dtype=np.int64)
    return (int(before[0]), int(before[1])), (int(table[:, 0].sum()), int(table[:, 1].sum()))
