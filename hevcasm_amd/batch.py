"""Batched hot-path calls on torch CUDA tensors.

torch is used only as plumbing: it owns the device memory and the stream.  Every function
passes raw device pointers to the C-ABI (include/hevcasm_hip.h) and raises ``HevcasmError``
on a non-zero return code -- there is no eager/PyTorch fallback for any of these.
"""
import numpy as np
import torch

from . import capi
from .capi import check, lib, rect


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(x, dtype):
    """numpy array or tensor -> contiguous CUDA tensor of ``dtype`` (uint32 travels as int32 bits)."""
    if isinstance(x, np.ndarray):
        if x.dtype == np.uint32:
            x = x.view(np.int32)
        x = torch.from_numpy(np.ascontiguousarray(x))
    if not x.is_cuda:
        x = x.cuda()
    assert x.dtype == dtype, (x.dtype, dtype)
    return x.contiguous()


def u8(x):
    return _dev(x, torch.uint8)


def i16(x):
    return _dev(x, torch.int16)


def off32(x):
    return _dev(x, torch.int32)


def sad(src, stride_src, ref, stride_ref, src_off, ref_off, w, h, out=None):
    n = src_off.numel()
    out = torch.empty(n, dtype=torch.int32, device=src.device) if out is None else out
    check(lib().hevcasm_hip_sad_batch(src.data_ptr(), stride_src, ref.data_ptr(), stride_ref, src_off.data_ptr(), ref_off.data_ptr(),
                                      out.data_ptr(), rect(w, h), n, _stream()), "sad_batch")
    return out


def sad_multiref(src, stride_src, ref, stride_ref, src_off, ref_off4, w, h, out=None):
    n = src_off.numel()
    assert ref_off4.numel() == 4 * n
    out = torch.empty((n, 4), dtype=torch.int32, device=src.device) if out is None else out
    check(lib().hevcasm_hip_sad_multiref_batch(src.data_ptr(), stride_src, ref.data_ptr(), stride_ref, src_off.data_ptr(),
                                               ref_off4.data_ptr(), out.data_ptr(), rect(w, h), n, _stream()), "sad_multiref_batch")
    return out


def cost_runs(kind, src, stride_src, ref, stride_ref, src_off, ref_off, runs, out=None):
    """hevcasm_hip_sad_runs / _sad_multiref_runs / _ssd_runs (kind "sad" | "sad4" | "ssd"): ONE launch for the block costs of a frame's whole partition.
    runs: numpy array of workload.COST_RUN_DTYPE (host); ref_off holds 4 offsets per block for "sad4", and the result is [n, 4]."""
    n = src_off.numel()
    ways = 4 if kind == "sad4" else 1
    assert ref_off.numel() == ways * n
    runs = np.ascontiguousarray(runs)
    assert runs.dtype.itemsize == 12
    if out is None:
        out = torch.zeros((n, 4) if ways == 4 else (n,), dtype=torch.int32, device=src.device)
    fn = {"sad": lib().hevcasm_hip_sad_runs, "sad4": lib().hevcasm_hip_sad_multiref_runs, "ssd": lib().hevcasm_hip_ssd_runs}[kind]
    check(fn(src.data_ptr(), stride_src, ref.data_ptr(), stride_ref, src_off.data_ptr(), ref_off.data_ptr(), out.data_ptr(), runs.ctypes.data, len(runs), _stream()), kind + "_runs")
    return out


def ssd(a, stride_a, b, stride_b, a_off, b_off, w, h, out=None):
    n = a_off.numel()
    out = torch.empty(n, dtype=torch.int32, device=a.device) if out is None else out
    check(lib().hevcasm_hip_ssd_batch(a.data_ptr(), stride_a, b.data_ptr(), stride_b, a_off.data_ptr(), b_off.data_ptr(),
                                      out.data_ptr(), w, h, n, _stream()), "ssd_batch")
    return out


def hadamard_satd(a, stride_a, b, stride_b, a_off, b_off, log2n, out=None):
    n = a_off.numel()
    out = torch.empty(n, dtype=torch.int32, device=a.device) if out is None else out
    check(lib().hevcasm_hip_hadamard_satd_batch(a.data_ptr(), stride_a, b.data_ptr(), stride_b, a_off.data_ptr(), b_off.data_ptr(),
                                                out.data_ptr(), log2n, n, _stream()), "hadamard_satd_batch")
    return out


def pred_uni(dst, stride_dst, ref, stride_ref, dst_off, ref_off, frac, taps, w, h):
    n = dst_off.numel()
    check(lib().hevcasm_hip_pred_uni_8to8_batch(dst.data_ptr(), stride_dst, ref.data_ptr(), stride_ref, dst_off.data_ptr(),
                                                ref_off.data_ptr(), frac.data_ptr(), taps, w, h, n, _stream()), "pred_uni_8to8_batch")
    return dst


def pred_bi(dst, stride_dst, ref0, ref1, stride_ref, dst_off, ref0_off, ref1_off, frac0, frac1, taps, w, h):
    n = dst_off.numel()
    check(lib().hevcasm_hip_pred_bi_8to8_batch(dst.data_ptr(), stride_dst, ref0.data_ptr(), ref1.data_ptr(), stride_ref,
                                               dst_off.data_ptr(), ref0_off.data_ptr(), ref1_off.data_ptr(),
                                               frac0.data_ptr(), frac1.data_ptr(), taps, w, h, n, _stream()), "pred_bi_8to8_batch")
    return dst


def _same_plane_shape(name, a, b):
    """The second plane of a two-plane call is addressed with the first one's offsets and strides: it must be a tensor of the same kind."""
    assert a.dtype == torch.uint8 and b.dtype == torch.uint8, "%s planes must be uint8" % name
    assert a.is_contiguous() and b.is_contiguous(), "%s planes must be contiguous" % name
    assert b.numel() >= a.numel(), "the second %s plane is shorter than the first (%d < %d bytes)" % (name, b.numel(), a.numel())
    assert a.device == b.device, "the two %s planes are on different devices" % name


def pred_uni_2planes(dst_a, dst_b, stride_dst, ref_a, ref_b, stride_ref, dst_off, ref_off, frac, taps, w, h):
    """Cb and Cr (one PU list, two planes) in one launch."""
    _same_plane_shape("dst", dst_a, dst_b)
    _same_plane_shape("ref", ref_a, ref_b)
    check(lib().hevcasm_hip_pred_uni_8to8_batch_2planes(dst_a.data_ptr(), dst_b.data_ptr(), stride_dst, ref_a.data_ptr(), ref_b.data_ptr(), stride_ref,
                                                        dst_off.data_ptr(), ref_off.data_ptr(), frac.data_ptr(), taps, w, h, dst_off.numel(), _stream()),
          "pred_uni_8to8_batch_2planes")
    return dst_a, dst_b


def pred_bi_2planes(dst_a, dst_b, stride_dst, ref0_a, ref0_b, ref1_a, ref1_b, stride_ref, dst_off, ref0_off, ref1_off, frac0, frac1, taps, w, h):
    _same_plane_shape("dst", dst_a, dst_b)
    _same_plane_shape("ref0", ref0_a, ref0_b)
    _same_plane_shape("ref1", ref1_a, ref1_b)
    check(lib().hevcasm_hip_pred_bi_8to8_batch_2planes(dst_a.data_ptr(), dst_b.data_ptr(), stride_dst, ref0_a.data_ptr(), ref0_b.data_ptr(),
                                                       ref1_a.data_ptr(), ref1_b.data_ptr(), stride_ref, dst_off.data_ptr(), ref0_off.data_ptr(),
                                                       ref1_off.data_ptr(), frac0.data_ptr(), frac1.data_ptr(), taps, w, h, dst_off.numel(), _stream()),
          "pred_bi_8to8_batch_2planes")
    return dst_a, dst_b


def inverse_transform_add(dst, stride_dst, pred, stride_pred, dst_off, pred_off, coeffs, tr_type, log2n):
    n = dst_off.numel()
    check(lib().hevcasm_hip_inverse_transform_add_batch(dst.data_ptr(), stride_dst, pred.data_ptr(), stride_pred, dst_off.data_ptr(),
                                                        pred_off.data_ptr(), coeffs.data_ptr(), tr_type, log2n, n, _stream()),
          "inverse_transform_add_batch")
    return dst


def dequant_inverse_transform_add(dst, stride_dst, pred, stride_pred, dst_off, pred_off, levels, scale, shift, tr_type, log2n):
    n = dst_off.numel()
    check(lib().hevcasm_hip_dequant_inverse_transform_add_batch(dst.data_ptr(), stride_dst, pred.data_ptr(), stride_pred,
                                                                dst_off.data_ptr(), pred_off.data_ptr(), levels.data_ptr(),
                                                                scale, shift, tr_type, log2n, n, _stream()),
          "dequant_inverse_transform_add_batch")
    return dst


def transform(src, src_stride, src_off, tr_type, log2n, out=None):
    n = src_off.numel()
    out = torch.empty(n << (2 * log2n), dtype=torch.int16, device=src.device) if out is None else out
    check(lib().hevcasm_hip_transform_batch(out.data_ptr(), src.data_ptr(), src_stride, src_off.data_ptr(), tr_type, log2n, n, _stream()),
          "transform_batch")
    return out


def quantize_inverse(src, scale, shift, out=None):
    out = torch.empty_like(src) if out is None else out
    check(lib().hevcasm_hip_quantize_inverse(out.data_ptr(), src.data_ptr(), scale, shift, src.numel(), _stream()), "quantize_inverse")
    return out


def quantize(src, scale, shift, offset, n_per_block, out=None, cbf=None):
    nb = src.numel() // n_per_block
    out = torch.empty_like(src) if out is None else out
    cbf = torch.empty(nb, dtype=torch.int32, device=src.device) if cbf is None else cbf
    check(lib().hevcasm_hip_quantize_batch(out.data_ptr(), src.data_ptr(), scale, shift, offset, n_per_block, nb, cbf.data_ptr(), _stream()),
          "quantize_batch")
    return out, cbf


def quantize_reconstruct(rec, stride_rec, pred, stride_pred, rec_off, pred_off, res, ncbs):
    n = rec_off.numel()
    check(lib().hevcasm_hip_quantize_reconstruct_batch(rec.data_ptr(), stride_rec, pred.data_ptr(), stride_pred, rec_off.data_ptr(),
                                                       pred_off.data_ptr(), res.data_ptr(), ncbs, n, _stream()), "quantize_reconstruct_batch")
    return rec


def _tu_runs(runs):
    runs = np.ascontiguousarray(runs)
    assert runs.dtype.itemsize == 28, "runs must be workload.TU_RUN_DTYPE (hevcasm_hip_tu_run, 28 bytes)"
    return runs


def transform_runs(src, src_stride, src_off, runs, n_coeffs, out=None):
    """hevcasm_hip_transform_runs: the forward transforms of a frame's TUs (runs: numpy workload.TU_RUN_DTYPE, host) in ONE launch -> int16 [n_coeffs]."""
    runs = _tu_runs(runs)
    out = torch.empty(n_coeffs, dtype=torch.int16, device=src.device) if out is None else out
    check(lib().hevcasm_hip_transform_runs(out.data_ptr(), src.data_ptr(), src_stride, src_off.data_ptr(), runs.ctypes.data, len(runs), _stream()), "transform_runs")
    return out


def quantize_runs(src, runs, n_tus, out=None, cbf=None):
    """hevcasm_hip_quantize_runs: the quantiser over a frame's TUs with each run's (scale, shift, offset) in ONE launch -> (levels, cbf per TU)."""
    runs = _tu_runs(runs)
    out = torch.empty_like(src) if out is None else out
    cbf = torch.empty(n_tus, dtype=torch.int32, device=src.device) if cbf is None else cbf
    check(lib().hevcasm_hip_quantize_runs(out.data_ptr(), src.data_ptr(), cbf.data_ptr(), runs.ctypes.data, len(runs), _stream()), "quantize_runs")
    return out, cbf


def inverse_transform_add_runs(dst, stride_dst, pred, stride_pred, dst_off, pred_off, coeffs, runs, dequant=False):
    """hevcasm_hip_inverse_transform_add_runs / (dequant) hevcasm_hip_dequant_inverse_transform_add_runs: ONE launch for a frame's TUs."""
    runs = _tu_runs(runs)
    fn = lib().hevcasm_hip_dequant_inverse_transform_add_runs if dequant else lib().hevcasm_hip_inverse_transform_add_runs
    check(fn(dst.data_ptr(), stride_dst, pred.data_ptr(), stride_pred, dst_off.data_ptr(), pred_off.data_ptr(), coeffs.data_ptr(), runs.ctypes.data, len(runs), _stream()),
          "dequant_inverse_transform_add_runs" if dequant else "inverse_transform_add_runs")
    return dst


def quantize_reconstruct_runs(rec, stride_rec, pred, stride_pred, rec_off, pred_off, res, runs):
    """hevcasm_hip_quantize_reconstruct_runs: rec = clip(pred + res) for units of every size in ONE launch."""
    runs = _tu_runs(runs)
    check(lib().hevcasm_hip_quantize_reconstruct_runs(rec.data_ptr(), stride_rec, pred.data_ptr(), stride_pred, rec_off.data_ptr(), pred_off.data_ptr(), res.data_ptr(),
                                                      runs.ctypes.data, len(runs), _stream()), "quantize_reconstruct_runs")
    return rec


def transform_quantize_reconstruct_runs(rec, stride_rec, src, stride_src, pred, stride_pred, rec_off, src_off, pred_off, runs_q, runs_dq, n_coeffs, n_tus, levels=None, cbf=None):
    """hevcasm_hip_transform_quantize_reconstruct_runs: src - pred -> transform -> quantise -> dequantise -> inverse + add for a frame's TUs in ONE launch
    -> (rec, levels, cbf)."""
    runs_q, runs_dq = _tu_runs(runs_q), _tu_runs(runs_dq)
    assert len(runs_q) == len(runs_dq)
    levels = torch.empty(n_coeffs, dtype=torch.int16, device=src.device) if levels is None else levels
    cbf = torch.empty(n_tus, dtype=torch.int32, device=src.device) if cbf is None else cbf
    check(lib().hevcasm_hip_transform_quantize_reconstruct_runs(rec.data_ptr(), stride_rec, src.data_ptr(), stride_src, pred.data_ptr(), stride_pred, rec_off.data_ptr(), src_off.data_ptr(),
                                                                pred_off.data_ptr(), levels.data_ptr(), cbf.data_ptr(), runs_q.ctypes.data, runs_dq.ctypes.data, len(runs_q), _stream()),
          "transform_quantize_reconstruct_runs")
    return rec, levels, cbf


def transform_quantize_reconstruct_planes(planes, rec_off, src_off, pred_off, runs_q, runs_dq, levels, cbf):
    """hevcasm_hip_transform_quantize_reconstruct_planes: the residual loop over the planes of a frame (luma, Cb, Cr: own pointers and strides) in ONE launch.
    planes: list of (rec tensor, stride_rec, src tensor, stride_src, pred tensor, stride_pred); a run's `plane` field picks its plane."""
    import ctypes as C
    runs_q, runs_dq = _tu_runs(runs_q), _tu_runs(runs_dq)
    assert len(runs_q) == len(runs_dq)
    table = (C.c_int64 * (6 * len(planes)))()      # hevcasm_hip_tu_plane = { rec, src, pred, stride_rec, stride_src, stride_pred }: six 8-byte members
    for i, (rec, srec, src, ssrc, pred, spred) in enumerate(planes):
        table[6 * i:6 * i + 6] = [rec.data_ptr(), src.data_ptr(), pred.data_ptr(), srec, ssrc, spred]
    check(lib().hevcasm_hip_transform_quantize_reconstruct_planes(table, len(planes), rec_off.data_ptr(), src_off.data_ptr(), pred_off.data_ptr(), levels.data_ptr(), cbf.data_ptr(),
                                                                  runs_q.ctypes.data, runs_dq.ctypes.data, len(runs_q), _stream()), "transform_quantize_reconstruct_planes")
    return [p[0] for p in planes], levels, cbf


def _result_rows(out, n, like):
    """The (dx, dy, cost) table of n answers: allocated, or the caller's, checked -- the entry points write 12 n bytes through a raw pointer
    (and, for small launches, memset them first), so a short, strided, int64 or other-device `out` would be overrun or silently misread."""
    if out is None:
        return torch.empty((n, 3), dtype=torch.int32, device=like.device)
    assert out.dtype == torch.int32, "out must be int32, not %s" % out.dtype
    assert out.is_contiguous(), "out must be contiguous"
    assert out.numel() >= 3 * n, "out holds %d ints, %d answers need %d" % (out.numel(), n, 3 * n)
    assert out.device == like.device, "out is on %s, the planes on %s" % (out.device, like.device)
    return out


def me_full_search(src, stride_src, ref, stride_ref, src_off, ref_off, w, h, search_range, out=None):
    n = src_off.numel()
    out = _result_rows(out, n, src)
    check(lib().hevcasm_hip_me_full_search_batch(src.data_ptr(), stride_src, ref.data_ptr(), stride_ref, src_off.data_ptr(),
                                                 ref_off.data_ptr(), out.data_ptr(), rect(w, h), search_range, n, _stream()),
          "me_full_search_batch")
    return out


def me_full_search_frame_rows(cur, cur_byte0, stride_cur, ref, ref_byte0, stride_ref, frame_w, frame_h, ctu, search_range, row0, row1, out=None):
    """Full search over CTU rows [row0, row1) of a frame.  ``cur`` / ``ref`` are device tensors holding (a slab of) the
    planes; ``cur_byte0`` / ``ref_byte0`` are the byte positions in them of the pixel in column 0 of CTU row ``row0``."""
    cols = -(-frame_w // ctu)
    out = _result_rows(out, (row1 - row0) * cols, cur)
    check(lib().hevcasm_hip_me_full_search_frame_rows(cur.data_ptr() + cur_byte0, stride_cur, ref.data_ptr() + ref_byte0, stride_ref,
                                                      frame_w, frame_h, ctu, search_range, row0, row1, out.data_ptr(), _stream()),
          "me_full_search_frame_rows")
    return out


def me_full_search_frame_ctus(cur, cur_byte0, stride_cur, ref, ref_byte0, stride_ref, frame_w, frame_h, ctu, search_range, first_ctu, n_ctus, out=None):
    """Full search over CTUs [first_ctu, first_ctu + n_ctus) of a frame in raster order; ``cur_byte0`` / ``ref_byte0`` are the byte
    positions of the pixel in column 0 of the CTU row that holds ``first_ctu``."""
    out = _result_rows(out, n_ctus, cur)
    check(lib().hevcasm_hip_me_full_search_frame_ctus(cur.data_ptr() + cur_byte0, stride_cur, ref.data_ptr() + ref_byte0, stride_ref,
                                                      frame_w, frame_h, ctu, search_range, first_ctu, n_ctus, out.data_ptr(), _stream()),
          "me_full_search_frame_ctus")
    return out


def me_full_search_frames_ctus(cur, cur_byte0, stride_cur, pitch_cur, ref, ref_byte0, stride_ref, pitch_ref, n_frames, frame_w, frame_h, ctu, search_range, first_ctu, n_ctus, out=None):
    """The same run of CTUs in n_frames slabs (pitch bytes apart in both tensors) in ONE launch -> int32 [n_frames, n_ctus, 3]."""
    out = torch.empty((n_frames, n_ctus, 3), dtype=torch.int32, device=cur.device) if out is None else out
    check(lib().hevcasm_hip_me_full_search_frames_ctus(cur.data_ptr() + cur_byte0, stride_cur, pitch_cur, ref.data_ptr() + ref_byte0, stride_ref, pitch_ref, n_frames,
                                                       frame_w, frame_h, ctu, search_range, first_ctu, n_ctus, out.data_ptr(), _stream()),
          "me_full_search_frames_ctus")
    return out


def subpel_refine(src, stride_src, ref, stride_ref, src_off, ref_off, w, h, metric=0, out=None):
    """Fused quarter-pel refinement: out[i] = (qx, qy, cost) of the best of the 49 displacements."""
    n = src_off.numel()
    out = _result_rows(out, n, src)
    check(lib().hevcasm_hip_subpel_refine_batch(src.data_ptr(), stride_src, ref.data_ptr(), stride_ref, src_off.data_ptr(),
                                                ref_off.data_ptr(), out.data_ptr(), rect(w, h), metric, n, _stream()),
          "subpel_refine_batch")
    return out


def me_full_search_runs(src, stride_src, ref, stride_ref, src_off, ref_off, runs, search_range, out=None):
    """hevcasm_hip_me_full_search_runs: the full search of a frame's whole partition (runs: numpy COST_RUN_DTYPE, host) in ONE launch -> int32 [n, 3]."""
    n = src_off.numel()
    out = _result_rows(out, n, src)
    runs = np.ascontiguousarray(runs)
    assert runs.dtype.itemsize == 12
    check(lib().hevcasm_hip_me_full_search_runs(src.data_ptr(), stride_src, ref.data_ptr(), stride_ref, src_off.data_ptr(), ref_off.data_ptr(), out.data_ptr(),
                                                runs.ctypes.data, len(runs), search_range, _stream()), "me_full_search_runs")
    return out


def subpel_refine_runs(src, stride_src, ref, stride_ref, src_off, ref_off, runs, metric=0, out=None):
    """hevcasm_hip_subpel_refine_runs: the quarter-pel refinement of a frame's whole partition in ONE launch -> int32 [n, 3]."""
    n = src_off.numel()
    out = _result_rows(out, n, src)
    runs = np.ascontiguousarray(runs)
    assert runs.dtype.itemsize == 12
    check(lib().hevcasm_hip_subpel_refine_runs(src.data_ptr(), stride_src, ref.data_ptr(), stride_ref, src_off.data_ptr(), ref_off.data_ptr(), out.data_ptr(),
                                               runs.ctypes.data, len(runs), metric, _stream()), "subpel_refine_runs")
    return out


def pred_intra(dst, neighbours, neighbour_stride, mode, packed, n):
    """Batched hevcasm_pred_intra, one mode for the batch: block i reads neighbours + i * neighbour_stride, writes dst[i * N * N ...].
    packed | INTRA_COMPACT: compact records (see pred_intra_modes)."""
    check(lib().hevcasm_hip_pred_intra_batch(dst.data_ptr(), _intra_base(neighbours, packed, (packed >> 1) & 0x7f), neighbour_stride, mode, packed, n, _stream()), "pred_intra_batch")
    return dst


INTRA_COMPACT = 0x10000     # HEVCASM_HIP_INTRA_COMPACT: the corner sample at [64 + 2N], records of 4N + 1 contiguous bytes


def _intra_base(neighbours, packed_or_log2, log2n):
    """The `neighbours` argument: the tensor's first byte is record 0's byte 0 -- or, for compact records, record 0's first SAMPLE (byte 64 - 2N of the
    record), in which case the pointer handed over lies 64 - 2N bytes before the tensor (never dereferenced there)."""
    return neighbours.data_ptr() - ((64 - (2 << log2n)) if packed_or_log2 & INTRA_COMPACT else 0)


def pred_intra_modes(dst, neighbours, neighbour_stride, modes, packed):
    """One intra mode per block (modes: uint8 tensor, 0..34).  Mode 1 is pinned to the reference; the others follow H.265 8.4.4.2.4 / .6 (unpinned).
    packed | INTRA_COMPACT: `neighbours` holds compact records (4N + 1 samples each, neighbour_stride apart, the first at the tensor's first byte)."""
    assert modes.dtype == torch.uint8 and modes.is_contiguous()
    check(lib().hevcasm_hip_pred_intra_modes_batch(dst.data_ptr(), _intra_base(neighbours, packed, (packed >> 1) & 0x7f), neighbour_stride, modes.data_ptr(), packed, modes.numel(),
                                                   _stream()), "pred_intra_modes_batch")
    return dst


def intra_filter_neighbours(neighbours, neighbour_stride, log2n, strong, n, out=None, out_stride=129):
    """H.265 8.4.4.2.3 (unpinned): the smoothed copies of n neighbour records, out_stride bytes apart."""
    out = torch.zeros(out_stride * n, dtype=torch.uint8, device=neighbours.device) if out is None else out
    assert out.dtype == torch.uint8 and out.is_contiguous() and out.numel() >= out_stride * (n - 1) + ((4 << (log2n & 0xff)) + 1 if log2n & INTRA_COMPACT else 129)
    check(lib().hevcasm_hip_intra_filter_neighbours_batch(_intra_base(out, log2n, log2n & 0xff), out_stride, _intra_base(neighbours, log2n, log2n & 0xff), neighbour_stride, log2n, int(strong), n,
                                                          _stream()), "intra_filter_neighbours_batch")
    return out


def ssd_linear(src0, src1, off0, off1, size, out=None):
    n = off0.numel()
    out = torch.empty(n, dtype=torch.int32, device=src0.device) if out is None else out
    check(lib().hevcasm_hip_ssd_linear_batch(src0.data_ptr(), src1.data_ptr(), off0.data_ptr(), off1.data_ptr(), out.data_ptr(), size, n, _stream()),
          "ssd_linear_batch")
    return out


def frame_sse(a, stride_a, b, stride_b, a_off, b_off, width, height, out=None):
    """Whole-frame SSE, 64-bit: out[i] = sum over rows of hevcasm_ssd_linear(row) for frame i (int64 tensor; the values are < 2^63)."""
    n = a_off.numel()
    if out is None:
        out = torch.empty(n, dtype=torch.int64, device=a.device)
    assert out.dtype == torch.int64 and out.is_contiguous() and out.numel() >= n and out.device == a.device
    check(lib().hevcasm_hip_frame_sse_batch(a.data_ptr(), stride_a, b.data_ptr(), stride_b, a_off.data_ptr(), b_off.data_ptr(), out.data_ptr(),
                                            width, height, n, _stream()), "frame_sse_batch")
    return out


class PredPlane(capi.C.Structure):
    """= hevcasm_hip_pred_plane"""
    _fields_ = [("dst", capi.C.c_void_p), ("stride_dst", capi.C.c_ssize_t), ("ref0", capi.C.c_void_p), ("ref1", capi.C.c_void_p), ("stride_ref", capi.C.c_ssize_t)]


def pred_runs(planes, runs, dst_off, ref0_off, frac0, ref1_off=None, frac1=None):
    """hevcasm_hip_pred_uni_8to8_runs (ref1_off is None) / hevcasm_hip_pred_bi_8to8_runs: ONE launch for a frame's whole partition.
    planes: list of (dst tensor, stride_dst, ref0 tensor, ref1 tensor or None, stride_ref); runs: numpy array of workload.PRED_RUN_DTYPE (host)."""
    table = (PredPlane * len(planes))()
    for i, (dst, sd, r0, r1, sr) in enumerate(planes):
        table[i] = PredPlane(dst.data_ptr(), sd, r0.data_ptr(), r1.data_ptr() if r1 is not None else None, sr)
    runs = np.ascontiguousarray(runs)
    assert runs.dtype.itemsize == 12
    L = lib()
    if ref1_off is None:
        check(L.hevcasm_hip_pred_uni_8to8_runs(table, len(planes), runs.ctypes.data, len(runs), dst_off.data_ptr(), ref0_off.data_ptr(), frac0.data_ptr(), _stream()), "pred_uni_runs")
    else:
        check(L.hevcasm_hip_pred_bi_8to8_runs(table, len(planes), runs.ctypes.data, len(runs), dst_off.data_ptr(), ref0_off.data_ptr(), ref1_off.data_ptr(),
                                              frac0.data_ptr(), frac1.data_ptr(), _stream()), "pred_bi_runs")


class IntraFrames:
    """hevcasm_hip_intra_reconstruct_frames on torch tensors: n_frames planes of frame_w x frame_h (row stride `stride`, frames `stride * frame_h` apart),
    a TU list per frame (hevcasm_amd.workload.intra_tu_map: tus + ctu_first), quantiser parameters int32 [4][5].  Device buffers and the workspace are made
    once; run() only enqueues (memset + two launches), so it can be captured into a HIP graph.  encode: src -> levels, cbf, recon; decode: levels -> recon."""

    def __init__(self, frame_w, frame_h, ctu, tus_per_frame, firsts_per_frame, qp, c_idx=0, strong=0, stride=None, device="cuda", want_pred=False):
        self.w, self.h, self.ctu, self.c_idx, self.strong = frame_w, frame_h, ctu, c_idx, strong
        self.stride = stride or frame_w
        self.n_frames = len(tus_per_frame)
        self.cols, self.rows = -(-frame_w // ctu), -(-frame_h // ctu)
        n_ctus = self.cols * self.rows
        base, firsts = 0, []
        for t, f in zip(tus_per_frame, firsts_per_frame):
            assert len(f) == n_ctus + 1 and int(f[-1]) == len(t)
            firsts.append(np.asarray(f[:-1], dtype=np.int64) + base)
            base += len(t)
        firsts.append(np.array([base], dtype=np.int64))
        self.n_tus = base
        tus = np.concatenate([np.ascontiguousarray(t).view(np.uint8).reshape(-1, 8) for t in tus_per_frame])
        self.tus = torch.from_numpy(tus).to(device)
        self.ctu_first = torch.from_numpy(np.concatenate(firsts).astype(np.int32)).to(device)
        self.qp = np.ascontiguousarray(qp, dtype=np.int32).reshape(4, 5)
        plane = self.stride * frame_h
        self.recon = torch.zeros(self.n_frames * plane, dtype=torch.uint8, device=device)
        self.pred = torch.zeros(self.n_frames * plane, dtype=torch.uint8, device=device) if want_pred else None
        self.levels = torch.zeros(self.n_frames * n_ctus * ctu * ctu, dtype=torch.int16, device=device)
        self.cbf = torch.zeros(self.n_tus, dtype=torch.int32, device=device)
        nbytes = lib().hevcasm_hip_intra_frames_workspace_bytes(frame_w, frame_h, ctu, self.n_frames)
        if nbytes == 0:
            raise capi.HevcasmError("intra frames: " + capi.last_error())
        self.workspace = torch.zeros(nbytes, dtype=torch.uint8, device=device)

    def run(self, src=None, stride_src=None):
        """src: uint8 tensor holding the n_frames source planes (encoder) or None (decoder: self.levels is the input)."""
        plane = self.stride * self.h
        s_stride = stride_src or self.stride
        check(lib().hevcasm_hip_intra_reconstruct_frames(
            self.recon.data_ptr(), self.stride, plane,
            src.data_ptr() if src is not None else None, s_stride, s_stride * self.h,
            self.levels.data_ptr(), self.cbf.data_ptr() if src is not None else None, self.pred.data_ptr() if self.pred is not None else None,
            self.tus.data_ptr(), self.ctu_first.data_ptr(), self.qp.ctypes.data,
            self.w, self.h, self.ctu, self.c_idx, 1 if self.strong else 0, self.n_frames, self.workspace.data_ptr(), _stream()), "intra_reconstruct_frames")

    def status(self):
        check(lib().hevcasm_hip_intra_frames_status(self.workspace.data_ptr(), _stream()), "intra_frames_status")
