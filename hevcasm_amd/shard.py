"""Splitting a batch of independent blocks across ranks (one process per GPU).

The hot path has no exchange step: blocks are independent, so rank r of W simply owns the
contiguous slice [start, stop) of the block list and its results.  No collective touches the
data path; torch.distributed is used by callers only for the timing barrier / max-reduce.
"""


def shard_range(n_blocks, rank, world_size):
    """Contiguous, balanced split: the first (n_blocks % world_size) ranks get one extra block."""
    if not 0 <= rank < world_size:
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    q, r = divmod(n_blocks, world_size)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def shard_arrays(rank, world_size, n_blocks, per_block):
    """Slice per-block arrays (dict name -> (array, items_per_block)) to this rank's blocks."""
    a, b = shard_range(n_blocks, rank, world_size)
    return {k: v[a * m:b * m] for k, (v, m) in per_block.items()}, (a, b)
