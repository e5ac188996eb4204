"""Splitting the hot path's work across ranks (one process per GPU).

The hot path has no exchange step.  A batch of independent blocks (BASELINE.json configs #2-#4):
rank r of W owns the contiguous slice [start, stop) of the block list and its results.  A frame's
motion search (config #5): rank r owns a contiguous range of CTU ROWS, and holds those rows of the
current frame plus the same rows of the reference frame widened by the search range above and below
(the halo) -- what it needs and nothing more.  No collective touches the data path;
torch.distributed is used by callers only for the timing barrier / max-reduce and, after the
search, to collect the 12-byte answers.
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


def ctu_row_range(n_ctu_rows, rank, world_size):
    """Config #5: contiguous CTU rows per rank, balanced to within one row.  With more ranks than rows
    the last ranks get an empty range."""
    return shard_range(n_ctu_rows, rank, world_size)


def me_slab_rows(frame_h, ctu, search_range, row0, row1):
    """Pixel rows a rank must hold to search CTU rows [row0, row1): (y0, y1, r0, r1) -- current-frame rows [y0, y1) and
    reference rows [r0, r1) = [y0 - range, y1 + range) in FRAME coordinates (negative / beyond frame_h = rows of the
    reference frame's padding; the padded plane's row index is r + pad).  The halo is what neighbouring ranks' slabs share:
    2 * range rows of the reference on each seam, copied to both sides, read-only -- no exchange at run time."""
    if not 0 <= row0 <= row1 <= -(-frame_h // ctu):
        raise ValueError("CTU rows %d..%d outside a frame of %d rows" % (row0, row1, frame_h))
    y0, y1 = row0 * ctu, min(row1 * ctu, frame_h)
    return y0, y1, y0 - search_range, y1 + search_range


def ctu_range(n_ctus, rank, world_size):
    """Config #5, balanced to one CTU instead of one row: rank r owns CTUs [first, last) of the frame in raster order (34 rows of
    60 CTUs over 8 ranks: 255 each, where whole rows are 300 or 240).  The rank then holds every CTU row its run touches -- a row
    that two ranks share is held by both -- plus the halo (me_slab_rows on those rows)."""
    return shard_range(n_ctus, rank, world_size)


def rows_of_ctus(first, last, cols):
    """CTU rows [row0, row1) that the raster run [first, last) touches (empty run: (row, row))."""
    if last <= first:
        return first // cols, first // cols
    return first // cols, (last - 1) // cols + 1
