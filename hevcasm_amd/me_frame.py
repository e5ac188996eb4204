"""Full-search motion estimation over a frame's CTU grid, whole or one rank's rows of it
(BASELINE.json config #5).

The unit of work is a CTU (64x64, or what the frame's right / bottom edge leaves of it: a 2160-row
frame ends in a row of 64x48 CTUs -- both are sizes of the reference's partition table,
sad.c:231-232); its cost function is the reference's hevcasm_sad (sad.c:47-60) at every
displacement of a +-range window, its answer the raster-first minimum.  CTUs are independent, so a
frame shards by contiguous CTU ROWS (hevcasm_amd.shard.ctu_row_range): a rank holds the rows of the
current frame it owns and the same rows of the reference frame plus `range` rows above and below
(the halo), never the whole frame, and there is no exchange step -- the only thing that crosses
ranks is the 12-byte answer per CTU, after the search.

Host-side plumbing only (numpy geometry, torch for device memory); the search itself is
hevcasm_hip_me_full_search_frame_rows (include/hevcasm_hip.h).  Nothing here falls back to a CPU
search: `rows_fn` defaults to the HIP entry point, and the CPU rehearsal in tests/ passes its own.
"""
import numpy as np

from . import shard


def synthetic_frame_pair(w, h, pad, seed, ctu=64, max_motion=40):
    """A current frame and a reference frame padded by `pad` replicated pixels, video-like: a
    smooth texture plus noise, the current frame being the texture moved by one motion vector per
    ctu x ctu block (uniform in +-max_motion) plus fresh noise -- so the minima of the search are
    interior, non-zero and different from CTU to CTU."""
    rng = np.random.default_rng(seed)
    m = max_motion
    tw, th = w + 2 * m, h + 2 * m
    noise = rng.integers(0, 256, size=(th, tw)).astype(np.float32)
    # separable box blurs of two radii (cumulative sums): low frequencies that make neighbouring candidates similar
    def blur(a, r):
        k = 2 * r + 1
        c = np.cumsum(np.pad(a, ((0, 0), (r + 1, r)), mode="edge"), axis=1)
        a = (c[:, k:] - c[:, :-k]) / k
        c = np.cumsum(np.pad(a, ((r + 1, r), (0, 0)), mode="edge"), axis=0)
        return (c[k:, :] - c[:-k, :]) / k
    tex = 128 + 3.0 * (blur(noise, 6) - 127.5) + 1.5 * (blur(noise, 2) - 127.5) + 0.15 * (noise - 127.5)
    tex = np.clip(np.rint(tex), 0, 255).astype(np.uint8)
    ref = np.pad(tex[m:m + h, m:m + w], pad, mode="edge")
    cur = np.empty((h, w), dtype=np.uint8)
    for y in range(0, h, ctu):
        for x in range(0, w, ctu):
            dx, dy = (int(v) for v in rng.integers(-m, m + 1, 2))
            hh, ww = min(ctu, h - y), min(ctu, w - x)
            cur[y:y + hh, x:x + ww] = tex[m + y + dy: m + y + dy + hh, m + x + dx: m + x + dx + ww]
    cur = np.clip(cur.astype(np.int16) + rng.integers(-2, 3, size=cur.shape), 0, 255).astype(np.uint8)
    return dict(cur=np.ascontiguousarray(cur).reshape(-1), stride_cur=w, ref=np.ascontiguousarray(ref).reshape(-1), stride_ref=w + 2 * pad,
                pad=pad, w=w, h=h)


def ctu_jobs(frame_h, frame_w, ctu, stride_cur, stride_ref, pad, row0, row1, cur_row_base=0, ref_row_base=0):
    """CTU rows [row0, row1) as block lists for the list-mode entry point (hevcasm_hip_me_full_search_batch): one job per
    distinct CTU shape, in raster order within a job.  Offsets are relative to planes whose first stored row is frame row
    `cur_row_base` (current) / padded-frame row `ref_row_base` (reference; padded row = frame row + pad)."""
    cols = -(-frame_w // ctu)
    jobs = {}
    for r in range(row0, row1):
        hh = min(ctu, frame_h - r * ctu)
        for c in range(cols):
            ww = min(ctu, frame_w - c * ctu)
            j = jobs.setdefault((ww, hh), dict(w=ww, h=hh, cur_off=[], ref_off=[], index=[]))
            j["cur_off"].append((r * ctu - cur_row_base) * stride_cur + c * ctu)
            j["ref_off"].append((r * ctu + pad - ref_row_base) * stride_ref + c * ctu + pad)
            j["index"].append((r - row0) * cols + c)
    out = []
    for j in jobs.values():
        out.append(dict(w=j["w"], h=j["h"], cur_off=np.array(j["cur_off"], dtype=np.uint32), ref_off=np.array(j["ref_off"], dtype=np.uint32),
                        index=np.array(j["index"], dtype=np.int64)))
    return out


class Slab:
    """What one rank holds of a frame pair: current rows [y0, y1) and padded-reference rows [y0 + pad - R, y1 + pad + R)."""

    def __init__(self, frame, ctu, search_range, row0, row1):
        pad, w, h = frame["pad"], frame["w"], frame["h"]
        if pad < search_range:
            raise ValueError("reference frame padded by %d pixels, search range is %d" % (pad, search_range))
        self.frame_w, self.frame_h, self.ctu, self.range, self.row0, self.row1 = w, h, ctu, search_range, row0, row1
        self.stride_cur, self.stride_ref = frame["stride_cur"], frame["stride_ref"]
        y0, y1, r0, r1 = shard.me_slab_rows(h, ctu, search_range, row0, row1)
        self.y0, self.y1 = y0, y1
        self.ref_row_base = r0 + pad                                 # first padded-reference row held
        self.cur = frame["cur"][y0 * self.stride_cur: y1 * self.stride_cur]
        self.ref = frame["ref"][(r0 + pad) * self.stride_ref: (r1 + pad) * self.stride_ref]
        self.cur_byte0 = 0                                           # pixel (0, y0) of the current frame
        self.ref_byte0 = search_range * self.stride_ref + pad        # the co-located reference pixel: R halo rows down, pad columns in
        self.pad = pad

    @property
    def n_ctus(self):
        return (self.row1 - self.row0) * -(-self.frame_w // self.ctu)


class CtuRun:
    """A raster run of CTUs [first, last) and the slab that holds every row it touches."""

    def __init__(self, frame, ctu, search_range, first, last):
        cols = -(-frame["w"] // ctu)
        self.first, self.last, self.cols = first, last, cols
        row0, row1 = shard.rows_of_ctus(first, last, cols)
        self.slab = Slab(frame, ctu, search_range, row0, row1)

    @property
    def n_ctus(self):
        return self.last - self.first


def hip_ctus_fn(device=None):
    """run_fn backed by hevcasm_hip_me_full_search_frame_ctus (uploads the run's slab, searches the run)."""
    import torch
    from . import batch as B

    def fn(run):
        dev = device or torch.device("cuda", torch.cuda.current_device())
        sl = run.slab
        cur_d, ref_d = torch.from_numpy(sl.cur).to(dev), torch.from_numpy(sl.ref).to(dev)
        out = B.me_full_search_frame_ctus(cur_d, sl.cur_byte0, sl.stride_cur, ref_d, sl.ref_byte0, sl.stride_ref,
                                          sl.frame_w, sl.frame_h, sl.ctu, sl.range, run.first, run.n_ctus)
        torch.cuda.synchronize()
        return out.cpu().numpy()

    return fn


class CtuRunStack:
    """One rank's share of F consecutive frames (round 6): the same raster run of CTUs [first, last) of each, the F slabs stacked back to back in two arrays --
    what hevcasm_hip_me_full_search_frames_ctus takes in ONE launch (slab f lies f * pitch bytes after slab 0)."""

    def __init__(self, frames, ctu, search_range, first, last):
        self.runs = [CtuRun(f, ctu, search_range, first, last) for f in frames]
        sl = self.runs[0].slab
        self.first, self.last, self.n_frames = first, last, len(frames)
        self.pitch_cur, self.pitch_ref = sl.cur.size, sl.ref.size
        assert all(r.slab.cur.size == self.pitch_cur and r.slab.ref.size == self.pitch_ref for r in self.runs), "the frames of a stack share one geometry"
        self.cur = np.concatenate([r.slab.cur for r in self.runs])
        self.ref = np.concatenate([r.slab.ref for r in self.runs])
        self.slab = sl

    @property
    def n_ctus(self):
        return self.last - self.first


def hip_frames_fn(device=None):
    """stack_fn backed by hevcasm_hip_me_full_search_frames_ctus: uploads the stack, ONE launch for the F frames -> int32 [F, n_ctus, 3]."""
    import torch
    from . import batch as B

    def fn(stack):
        dev = device or torch.device("cuda", torch.cuda.current_device())
        sl = stack.slab
        cur_d, ref_d = torch.from_numpy(stack.cur).to(dev), torch.from_numpy(stack.ref).to(dev)
        out = B.me_full_search_frames_ctus(cur_d, sl.cur_byte0, sl.stride_cur, stack.pitch_cur, ref_d, sl.ref_byte0, sl.stride_ref, stack.pitch_ref, stack.n_frames,
                                           sl.frame_w, sl.frame_h, sl.ctu, sl.range, stack.first, stack.n_ctus)
        torch.cuda.synchronize()
        return out.cpu().numpy()

    return fn


def search_rank_ctus_frames(frames, ctu, search_range, rank, world, stack_fn):
    """One rank's share of F frames, split by CTUs: -> (first, last, int32[F, n, 3]).  stack_fn(CtuRunStack) -> [F, n, 3]."""
    rows, cols = -(-frames[0]["h"] // ctu), -(-frames[0]["w"] // ctu)
    first, last = shard.ctu_range(rows * cols, rank, world)
    if first == last:
        return first, last, np.zeros((len(frames), 0, 3), dtype=np.int32)
    return first, last, np.asarray(stack_fn(CtuRunStack(frames, ctu, search_range, first, last)), dtype=np.int32).reshape(len(frames), last - first, 3)


def search_rank_ctus(frame, ctu, search_range, rank, world, run_fn):
    """One rank's share of a frame when the split is by CTUs: -> (first, last, int32[n, 3])"""
    rows, cols = -(-frame["h"] // ctu), -(-frame["w"] // ctu)
    first, last = shard.ctu_range(rows * cols, rank, world)
    if first == last:
        return first, last, np.zeros((0, 3), dtype=np.int32)
    return first, last, np.asarray(run_fn(CtuRun(frame, ctu, search_range, first, last)), dtype=np.int32)


def assemble_ctus(parts, frame_w, frame_h, ctu):
    """[(first, last, results)] from every rank -> the frame's table; checks that the runs tile the frame exactly once."""
    rows, cols = -(-frame_h // ctu), -(-frame_w // ctu)
    out = np.full((rows * cols, 3), np.iinfo(np.int32).min, dtype=np.int32)
    seen = np.zeros(rows * cols, dtype=bool)
    for first, last, res in parts:
        assert not seen[first:last].any(), "CTUs %d..%d assigned twice" % (first, last)
        seen[first:last] = True
        out[first:last] = np.asarray(res).reshape(-1, 3)
    assert seen.all(), "CTUs without an owner"
    return out


def hip_rows_fn(device=None, stream_sync=True):
    """rows_fn backed by hevcasm_hip_me_full_search_frame_rows: uploads the slab it is given (the rank's H2D copy, halo
    included) and searches it, returning the answers as a numpy array.  Returns fn."""
    import torch
    from . import batch as B

    def fn(slab):
        dev = device or torch.device("cuda", torch.cuda.current_device())
        cur_d = torch.from_numpy(slab.cur).to(dev)
        ref_d = torch.from_numpy(slab.ref).to(dev)
        out = B.me_full_search_frame_rows(cur_d, slab.cur_byte0, slab.stride_cur, ref_d, slab.ref_byte0, slab.stride_ref,
                                          slab.frame_w, slab.frame_h, slab.ctu, slab.range, slab.row0, slab.row1)
        if stream_sync:
            torch.cuda.synchronize()
        return out.cpu().numpy()

    return fn


def search_rank(frame, ctu, search_range, rank, world, rows_fn):
    """One rank's share of a frame: its CTU rows, searched on its own slab.  -> (row0, row1, int32[n, 3])"""
    rows = -(-frame["h"] // ctu)
    row0, row1 = shard.ctu_row_range(rows, rank, world)
    if row0 == row1:
        return row0, row1, np.zeros((0, 3), dtype=np.int32)
    return row0, row1, np.asarray(rows_fn(Slab(frame, ctu, search_range, row0, row1)), dtype=np.int32)


def assemble(parts, frame_w, frame_h, ctu):
    """[(row0, row1, results)] from every rank -> the frame's (rows * cols, 3) table in raster order; checks that the
    rows tile the frame exactly once."""
    rows, cols = -(-frame_h // ctu), -(-frame_w // ctu)
    out = np.full((rows * cols, 3), np.iinfo(np.int32).min, dtype=np.int32)
    seen = np.zeros(rows, dtype=bool)
    for row0, row1, res in parts:
        assert not seen[row0:row1].any(), "CTU rows %d..%d assigned twice" % (row0, row1)
        seen[row0:row1] = True
        out[row0 * cols: row1 * cols] = np.asarray(res).reshape(-1, 3)
    assert seen.all(), "CTU rows without an owner"
    return out
