"""One inter-coded frame through the whole hot path, the way an encoder's inner loop strings the reference's functions together
(a codec fills its tables once and calls per block; here every stage is ONE batch call over the frame's blocks):

    full search (hevcasm_sad over a +-range window)  ->  quarter-pel refinement (hevcasm_pred_uni_8to8 + hevcasm_sad, 49 candidates)
    ->  hevcasm_pred_uni_8to8 at the winning vector  ->  residual  ->  hevcasm_transform  ->  hevcasm_quantize (levels, cbf)
    ->  hevcasm_quantize_inverse + hevcasm_inverse_transform_add (fused)  ->  reconstruction  ->  hevcasm_ssd per block, frame SSE / PSNR

`stages(frame, ops)` runs the chain on whatever `ops` provides: `HipOps` = the batch C-ABI on device tensors (hevcasm_amd.batch), or -- in tests -- an
oracle-backed object with the same methods on numpy arrays.  With HipOps NOTHING visits the host between the first stage and the last: the search
answers become the next stage's descriptors by a few tensor operations on the device, so the whole chain is a fixed sequence of launches on one stream
and can be captured into a HIP graph and replayed per frame (`HipOps.capture`; examples/frame_pipeline.py times both).  The two steps the reference's
library has no function for are the residual (current - prediction as int16) and turning vectors into offsets: a codec does those itself; here they are
torch / numpy arithmetic.  Host-side plumbing only; no arithmetic of the hot path lives here."""
import numpy as np


def block_grid(w, h, size, stride_cur, stride_ref, pad):
    by, bx = np.meshgrid(np.arange(h // size), np.arange(w // size), indexing="ij")
    cur_off = (by * size * stride_cur + bx * size).reshape(-1).astype(np.uint32)
    ref_off = ((by * size + pad) * stride_ref + bx * size + pad).reshape(-1).astype(np.uint32)
    return cur_off, ref_off


def stages(frame, ops, size=16, search_range=8, qscale=26214, qshift=21, qoffset=171 << 7, dqscale=160, dqshift=1):
    """-> dict of every intermediate, in the backend's array type (HipOps.to_numpy() converts): mv, qmv, pred, residual, coeffs, levels, cbf, recon,
    block_ssd, frame_sse (one element)."""
    w, h, pad = frame["w"], frame["h"], frame["pad"]
    sc, sr = frame["stride_cur"], frame["stride_ref"]
    log2n = size.bit_length() - 1
    cur, ref, cur_off, ref_base = ops.plan(frame, size)                                       # resident planes and descriptor arrays (uploaded once per frame / geometry)
    mv = ops.me_full_search(cur, sc, ref, sr, cur_off, ref_base, size, size, search_range)   # (dx, dy, cost) per block
    ref_int = ops.offsets_after_search(ref_base, mv, sr)
    qmv = ops.subpel_refine(cur, sc, ref, sr, cur_off, ref_int, size, size)                  # (qx, qy, cost), quarter pels
    ref_sub, frac = ops.subpel_descriptors(ref_int, qmv, sr)
    pred = ops.pred_uni(w * h, sc, ref, sr, cur_off, ref_sub, frac, 8, size, size)           # a plane laid out like the current frame
    residual = ops.residual(cur, pred)                                                       # int16 plane, stride sc
    coeffs = ops.transform(residual, sc, cur_off, 0, log2n)
    levels, cbf = ops.quantize(coeffs, qscale, qshift, qoffset, size * size)
    recon = ops.dequant_inverse_transform_add(w * h, sc, pred, sc, cur_off, cur_off, levels, dqscale, dqshift, 0, log2n)
    block_ssd = ops.ssd(cur, sc, recon, sc, cur_off, cur_off, size, size)
    frame_sse = ops.frame_sse(cur, sc, recon, sc, w, h)
    return dict(mv=mv, qmv=qmv, pred=pred, residual=residual, coeffs=coeffs, levels=levels, cbf=cbf, recon=recon, block_ssd=block_ssd, frame_sse=frame_sse)


def partition_grid(pus, stride_cur, stride_ref, pad):
    """A frame's prediction units [(x, y, w, h)] (hevcasm_amd.workload.inter_partition) sorted by size -> descriptor arrays in that order and the runs:
    (cur_off, ref_base, sizes = [(w, h, first, count)], order = the sort permutation)."""
    pus = np.asarray(pus, dtype=np.int64).reshape(-1, 4)
    order = np.lexsort((np.arange(len(pus)), pus[:, 3], pus[:, 2]))
    p = pus[order]
    cur_off = (p[:, 1] * stride_cur + p[:, 0]).astype(np.uint32)
    ref_base = ((p[:, 1] + pad) * stride_ref + p[:, 0] + pad).astype(np.uint32)
    sizes, start = [], 0
    for i in range(1, len(p) + 1):
        if i == len(p) or (p[i, 2:] != p[start, 2:]).any():
            sizes.append((int(p[start, 2]), int(p[start, 3]), start, i - start))
            start = i
    return cur_off, ref_base, sizes, order


def tu_grid(tus, stride, qp):
    """A plane's transform units [(x, y, log2, tr_type)] (hevcasm_amd.workload.tu_partition) sorted by (size, type) -> (offsets in that order, sizes =
    [(log2, tr_type, first, count, coeff_first, (q_scale, q_shift, q_offset, dq_scale, dq_shift))] with the parameters HEVC uses for the size at `qp`, order)."""
    from . import workload as wl
    t = np.asarray(tus, dtype=np.int64).reshape(-1, 4)
    order = np.lexsort((np.arange(len(t)), -t[:, 3], t[:, 2]))
    t = t[order]
    off = (t[:, 1] * stride + t[:, 0]).astype(np.uint32)
    q = wl.intra_qp(qp)
    sizes, start, coeff = [], 0, 0
    for i in range(1, len(t) + 1):
        if i == len(t) or (t[i, 2:] != t[start, 2:]).any():
            log2 = int(t[start, 2])
            sizes.append((log2, int(t[start, 3]), start, i - start, coeff, tuple(int(v) for v in q[log2 - 2])))
            coeff += (i - start) << (2 * log2)
            start = i
    return off, sizes, order


def stages_partition(frame, ops, pus, tu=16, search_range=8, qscale=26214, qshift=21, qoffset=171 << 7, dqscale=160, dqshift=1, tus=None, qp=27, fused=False):
    """The same chain on a REAL partition (round 6): the frame cut into prediction units of many sizes (`pus`, any of the 24 of
    pred_inter.c:438-444), transform units on a uniform tu x tu grid (HEVC's transform tree is independent of the PU boundaries inside a CU).
    Every stage is ONE launch for the whole frame: hevcasm_hip_me_full_search_runs, hevcasm_hip_subpel_refine_runs, hevcasm_hip_pred_uni_8to8_runs,
    hevcasm_hip_sad_runs (the PUs' prediction cost -- which must be the refinement's winning cost), the transform stages on the TU grid,
    hevcasm_hip_ssd_runs per PU.  With `tus` (a residual quad-tree over the four transform sizes, hevcasm_amd.workload.tu_partition) the transform stages run on
    THAT partition instead, one launch each as well (hevcasm_hip_transform_runs, _quantize_runs, _dequant_inverse_transform_add_runs), every size with the
    (de)quantiser parameters HEVC gives it at `qp`; coefficients and levels are then in run order (tu_grid).  `fused` (with `tus`): the residual loop as ONE launch
    from the current and predicted planes (hevcasm_hip_transform_quantize_reconstruct_runs); residual and coefficients then never exist (None in the result).
    -> dict of every intermediate; per-PU arrays are in size order (`order` maps them back to `pus`)."""
    w, h, pad = frame["w"], frame["h"], frame["pad"]
    sc, sr = frame["stride_cur"], frame["stride_ref"]
    log2n = tu.bit_length() - 1
    cur, ref, cur_off, ref_base, sizes, order, tu_off = ops.plan_partition(frame, pus, tu)
    mv = ops.me_full_search_runs(cur, sc, ref, sr, cur_off, ref_base, sizes, search_range)      # every PU size, one launch
    ref_int = ops.offsets_after_search(ref_base, mv, sr)
    qmv = ops.subpel_refine_runs(cur, sc, ref, sr, cur_off, ref_int, sizes)                      # one launch
    ref_sub, frac = ops.subpel_descriptors(ref_int, qmv, sr)
    pred = ops.pred_uni_runs(w * h, sc, ref, sr, cur_off, ref_sub, frac, sizes)                 # every PU size, one launch
    pu_sad = ops.cost_runs("sad", cur, sc, pred, sc, cur_off, cur_off, sizes)                   # one launch; == qmv[:, 2]
    residual = None if fused else ops.residual(cur, pred)
    if fused:
        tu_off, tu_sizes = ops.plan_tus(frame, tus, qp)
        coeffs = None
        recon, levels, cbf = ops.transform_quantize_reconstruct_runs(w * h, sc, cur, pred, tu_off, tu_sizes, len(tus))   # one launch for the whole residual loop
    elif tus is None:
        coeffs = ops.transform(residual, sc, tu_off, 0, log2n)
        levels, cbf = ops.quantize(coeffs, qscale, qshift, qoffset, tu * tu)
        recon = ops.dequant_inverse_transform_add(w * h, sc, pred, sc, tu_off, tu_off, levels, dqscale, dqshift, 0, log2n)
    else:
        tu_off, tu_sizes = ops.plan_tus(frame, tus, qp)
        coeffs = ops.transform_runs(residual, sc, tu_off, tu_sizes, w * h)                       # every TU size, one launch
        levels, cbf = ops.quantize_runs(coeffs, tu_sizes, len(tus))                              # one launch
        recon = ops.dequant_inverse_transform_add_runs(w * h, sc, pred, sc, tu_off, tu_sizes, levels)   # one launch
    pu_ssd = ops.cost_runs("ssd", cur, sc, recon, sc, cur_off, cur_off, sizes)                  # one launch
    frame_sse = ops.frame_sse(cur, sc, recon, sc, w, h)
    return dict(mv=mv, qmv=qmv, pred=pred, pu_sad=pu_sad, residual=residual, coeffs=coeffs, levels=levels, cbf=cbf, recon=recon, pu_ssd=pu_ssd, frame_sse=frame_sse, order=order)


def psnr(frame_sse, w, h):
    return float("inf") if frame_sse == 0 else 10.0 * np.log10(255.0 * 255.0 * w * h / float(frame_sse))


class HipOps:
    """The chain's stages on the GPU.  Everything is a device tensor from plan() on; no method synchronises or copies to the host."""

    def __init__(self, device="cuda"):
        import torch
        from . import batch
        self.t, self.B, self.device = torch, batch, device
        self._plans = {}
        self._partitions = {}

    MAX_PLANS = 4      # resident frames kept (the cache used to grow by one pair of device planes per frame dict, for ever)

    def plan(self, frame, size):
        """Device copies of the frame's planes and its block grid, made on first use and kept while the SAME frame dict (same arrays) comes back -- what
        capture() / replay need.  The host arrays are copied ONCE: to run the chain on new content, write into the resident planes (update()) or pass a new
        frame dict; at most MAX_PLANS frames stay resident, the oldest goes first."""
        if frame["w"] % size or frame["h"] % size:
            raise ValueError("the chain tiles the frame with %dx%d blocks: %dx%d is not a multiple (uncovered pixels would stay 0 in the reconstruction and count in the SSE)"
                             % (size, size, frame["w"], frame["h"]))
        key = (id(frame["cur"]), id(frame["ref"]), size)
        if key not in self._plans:
            cur_off, ref_base = block_grid(frame["w"], frame["h"], size, frame["stride_cur"], frame["stride_ref"], frame["pad"])
            up = lambda a: self.t.from_numpy(np.ascontiguousarray(a)).to(self.device)
            while len(self._plans) >= self.MAX_PLANS:
                self._plans.pop(next(iter(self._plans)))
            self._plans[key] = (frame, up(frame["cur"]), up(frame["ref"]), self.B.off32(cur_off), self.B.off32(ref_base),
                                self.B.off32(np.zeros(1, dtype=np.uint32)))
        p = self._plans[key]
        self._zero = p[5]
        return p[1], p[2], p[3], p[4]

    def plan_partition(self, frame, pus, tu):
        """plan() for stages_partition: the resident planes of plan() (TU grid = the uniform blocks) + the PUs' descriptor arrays and runs, cached with them."""
        cur, ref, tu_off, _ = self.plan(frame, tu)
        key = (id(frame["cur"]), id(frame["ref"]), tu, "pus", id(pus))
        if key not in self._partitions:
            cur_off, ref_base, sizes, order = partition_grid(pus, frame["stride_cur"], frame["stride_ref"], frame["pad"])
            from . import workload as wl
            while len(self._partitions) >= self.MAX_PLANS:
                self._partitions.pop(next(iter(self._partitions)))
            self._partitions[key] = (pus, self.B.off32(cur_off), self.B.off32(ref_base), sizes, order,
                                     np.array([(f, n, 8, pw, ph, 0) for (pw, ph, f, n) in sizes], dtype=wl.PRED_RUN_DTYPE),
                                     np.array([(f, n, (pw << 8) | ph) for (pw, ph, f, n) in sizes], dtype=wl.COST_RUN_DTYPE))
        p = self._partitions[key]
        self._pred_runs, self._cost_runs = p[5], p[6]
        return cur, ref, p[1], p[2], p[3], p[4], tu_off

    def plan_tus(self, frame, tus, qp):
        """The TU partition's descriptor array (device) and run tables (host), cached with the frame's planes."""
        key = (id(frame["cur"]), id(frame["ref"]), "tus", id(tus), qp)
        if key not in self._partitions:
            from . import workload as wl
            off, sizes, _ = tu_grid(tus, frame["stride_cur"], qp)
            while len(self._partitions) >= self.MAX_PLANS:
                self._partitions.pop(next(iter(self._partitions)))
            self._partitions[key] = (tus, self.B.off32(off), sizes,
                                     np.array([(f, n, cf, l2, tr, 0, q[0], q[1], q[2]) for (l2, tr, f, n, cf, q) in sizes], dtype=wl.TU_RUN_DTYPE),
                                     np.array([(f, n, cf, l2, tr, 0, q[3], q[4], 0) for (l2, tr, f, n, cf, q) in sizes], dtype=wl.TU_RUN_DTYPE))
        p = self._partitions[key]
        self._tu_runs_q, self._tu_runs_dq = p[3], p[4]
        return p[1], p[2]

    def transform_runs(self, residual, stride, off, tu_sizes, n_coeffs):
        return self.B.transform_runs(residual, stride, off, self._tu_runs_q, n_coeffs)

    def quantize_runs(self, coeffs, tu_sizes, n_tus):
        return self.B.quantize_runs(coeffs, self._tu_runs_q, n_tus)

    def dequant_inverse_transform_add_runs(self, dst_bytes, sd, pred, sp, off, tu_sizes, levels):
        dst = self.t.zeros(dst_bytes, dtype=self.t.uint8, device=self.device)
        return self.B.inverse_transform_add_runs(dst, sd, pred, sp, off, off, levels, self._tu_runs_dq, dequant=True)

    def transform_quantize_reconstruct_runs(self, n_bytes, stride, cur, pred, off, tu_sizes, n_tus):
        rec = self.t.zeros(n_bytes, dtype=self.t.uint8, device=self.device)
        return self.B.transform_quantize_reconstruct_runs(rec, stride, cur, stride, pred, stride, off, off, off, self._tu_runs_q, self._tu_runs_dq, n_bytes, n_tus)

    def me_full_search_runs(self, cur, sc, ref, sr, cur_off, ref_off, sizes, search_range):
        return self.B.me_full_search_runs(cur, sc, ref, sr, cur_off, ref_off, self._cost_runs, search_range)

    def subpel_refine_runs(self, cur, sc, ref, sr, cur_off, ref_off, sizes):
        return self.B.subpel_refine_runs(cur, sc, ref, sr, cur_off, ref_off, self._cost_runs)

    def pred_uni_runs(self, dst_bytes, sd, ref, sr, dst_off, ref_off, frac, sizes):
        dst = self.t.zeros(dst_bytes, dtype=self.t.uint8, device=self.device)
        self.B.pred_runs([(dst, sd, ref, None, sr)], self._pred_runs, dst_off, ref_off, frac)
        return dst

    def cost_runs(self, kind, a, sa, b, sb, a_off, b_off, sizes):
        return self.B.cost_runs(kind, a, sa, b, sb, a_off, b_off, self._cost_runs)

    def update(self, frame, size, cur=None, ref=None):
        """New content for a resident frame: copied INTO the device planes plan() made (so a captured graph replays on it)."""
        cur_d, ref_d, _, _ = self.plan(frame, size)
        if cur is not None:
            cur_d.copy_(self.t.from_numpy(np.ascontiguousarray(cur)))
        if ref is not None:
            ref_d.copy_(self.t.from_numpy(np.ascontiguousarray(ref)))

    def me_full_search(self, cur, sc, ref, sr, cur_off, ref_off, w, h, search_range):
        return self.B.me_full_search(cur, sc, ref, sr, cur_off, ref_off, w, h, search_range)

    def offsets_after_search(self, ref_base, mv, sr):
        return ref_base + mv[:, 1] * sr + mv[:, 0]                      # int32 arithmetic = the u32 offsets' bits

    def subpel_refine(self, cur, sc, ref, sr, cur_off, ref_off, w, h):
        return self.B.subpel_refine(cur, sc, ref, sr, cur_off, ref_off, w, h)

    def subpel_descriptors(self, ref_int, qmv, sr):
        ref_sub = ref_int + (qmv[:, 1] >> 2) * sr + (qmv[:, 0] >> 2)
        frac = ((qmv[:, 0] & 3) | ((qmv[:, 1] & 3) << 4)).to(self.t.uint8)
        return ref_sub, frac

    def pred_uni(self, dst_bytes, sd, ref, sr, dst_off, ref_off, frac, taps, w, h):
        dst = self.t.zeros(dst_bytes, dtype=self.t.uint8, device=self.device)
        return self.B.pred_uni(dst, sd, ref, sr, dst_off, ref_off, frac, taps, w, h)

    def residual(self, cur, pred):
        return cur.to(self.t.int16) - pred.to(self.t.int16)

    def transform(self, residual, stride, off, tr_type, log2n):
        return self.B.transform(residual, stride, off, tr_type, log2n)

    def quantize(self, coeffs, scale, shift, offset, n_per_block):
        return self.B.quantize(coeffs, scale, shift, offset, n_per_block)

    def dequant_inverse_transform_add(self, dst_bytes, sd, pred, sp, dst_off, pred_off, levels, scale, shift, tr_type, log2n):
        dst = self.t.zeros(dst_bytes, dtype=self.t.uint8, device=self.device)
        return self.B.dequant_inverse_transform_add(dst, sd, pred, sp, dst_off, pred_off, levels, scale, shift, tr_type, log2n)

    def ssd(self, a, sa, b, sb, a_off, b_off, w, h):
        return self.B.ssd(a, sa, b, sb, a_off, b_off, w, h)

    def frame_sse(self, a, sa, b, sb, w, h):
        return self.B.frame_sse(a, sa, b, sb, self._zero, self._zero, w, h)

    def capture(self, frame, pus=None, **kw):
        """The whole chain of `frame` as ONE HIP graph (torch.cuda.CUDAGraph: stream capture of the launches stages() -- with `pus`: stages_partition() -- issues).
        -> (graph, results): graph.replay() re-runs the chain on the frame's resident planes, results are the tensors it writes."""
        t = self.t
        run = (lambda: stages(frame, self, **kw)) if pus is None else (lambda: stages_partition(frame, self, pus, **kw))
        side = t.cuda.Stream()
        side.wait_stream(t.cuda.current_stream())
        with t.cuda.stream(side):                  # warm-up on a side stream: uploads, allocator, first launches
            run()
        t.cuda.current_stream().wait_stream(side)
        t.cuda.synchronize()
        graph = t.cuda.CUDAGraph()
        with t.cuda.graph(graph):
            results = run()
        return graph, results

    @staticmethod
    def to_numpy(results):
        import torch
        torch.cuda.synchronize()
        return {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in results.items()}
