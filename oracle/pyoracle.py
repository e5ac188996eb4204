"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE ONLY).

Every wrapper takes/returns numpy arrays and mirrors one ``orc_*_batch``
driver of ``hevc_oracle.h``: a base plane plus per-block u32 offsets -- the
same descriptors the HIP C-ABI (include/hevcasm_hip.h) takes, so a parity test
feeds both sides identical arguments.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile liboracle.so (and _ref/refdrv when /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        # The drivers are OpenMP loops.  A GPU box shows the process every core of its host but grants a share of them (a CPU quota): one thread per VISIBLE
        # core then costs a scheduler period per parallel region -- 0.12 s per call whatever its size, which was 500 of the GPU suite's 550 seconds.  Eight
        # threads are plenty for test-sized inputs; an explicit OMP_NUM_THREADS wins.  (dlsym through the library's own handle finds its OpenMP runtime.)
        if "OMP_NUM_THREADS" not in os.environ:
            try:
                _LIB.omp_set_num_threads(C.c_int(min(8, os.cpu_count() or 8)))
            except AttributeError:
                pass
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _i16(a):
    return np.ascontiguousarray(a, dtype=np.int16)


def rect(w, h):
    return (w << 8) | h


def sad_batch(src, stride_src, ref, stride_ref, src_off, ref_off, w, h):
    src, ref, src_off, ref_off = _u8(src), _u8(ref), _u32(src_off), _u32(ref_off)
    n = src_off.size
    out = np.empty(n, dtype=np.int32)
    lib().orc_sad_batch(_p(src), C.c_ssize_t(stride_src), _p(ref), C.c_ssize_t(stride_ref),
                        _p(src_off), _p(ref_off), _p(out), C.c_uint32(rect(w, h)), C.c_size_t(n))
    return out


def sad_multiref_batch(src, stride_src, ref, stride_ref, src_off, ref_off4, w, h):
    src, ref, src_off, ref_off4 = _u8(src), _u8(ref), _u32(src_off), _u32(ref_off4)
    n = src_off.size
    assert ref_off4.size == 4 * n
    out = np.empty((n, 4), dtype=np.int32)
    lib().orc_sad_multiref_batch(_p(src), C.c_ssize_t(stride_src), _p(ref), C.c_ssize_t(stride_ref),
                                 _p(src_off), _p(ref_off4), _p(out), C.c_uint32(rect(w, h)), C.c_size_t(n))
    return out


def ssd_batch(a, stride_a, b, stride_b, a_off, b_off, w, h):
    a, b, a_off, b_off = _u8(a), _u8(b), _u32(a_off), _u32(b_off)
    n = a_off.size
    out = np.empty(n, dtype=np.int32)
    lib().orc_ssd_batch(_p(a), C.c_ssize_t(stride_a), _p(b), C.c_ssize_t(stride_b),
                        _p(a_off), _p(b_off), _p(out), C.c_int(w), C.c_int(h), C.c_size_t(n))
    return out


def hadamard_satd_batch(n, a, stride_a, b, stride_b, a_off, b_off):
    a, b, a_off, b_off = _u8(a), _u8(b), _u32(a_off), _u32(b_off)
    out = np.empty(a_off.size, dtype=np.int32)
    lib().orc_hadamard_satd_batch(C.c_int(n), _p(a), C.c_ssize_t(stride_a), _p(b), C.c_ssize_t(stride_b),
                                  _p(a_off), _p(b_off), _p(out), C.c_size_t(a_off.size))
    return out


def pred_uni_batch(taps, dst_bytes, stride_dst, ref, stride_ref, dst_off, ref_off, frac, w, h):
    ref, dst_off, ref_off, frac = _u8(ref), _u32(dst_off), _u32(ref_off), _u8(frac)
    n = dst_off.size
    dst = np.zeros(dst_bytes, dtype=np.uint8)
    lib().orc_pred_uni_batch(C.c_int(taps), _p(dst), C.c_ssize_t(stride_dst), _p(ref), C.c_ssize_t(stride_ref),
                             _p(dst_off), _p(ref_off), _p(frac), C.c_int(w), C.c_int(h), C.c_size_t(n))
    return dst


def pred_bi_batch(taps, dst_bytes, stride_dst, ref0, ref1, stride_ref, dst_off, ref0_off, ref1_off, frac0, frac1, w, h):
    ref0, ref1 = _u8(ref0), _u8(ref1)
    dst_off, ref0_off, ref1_off = _u32(dst_off), _u32(ref0_off), _u32(ref1_off)
    frac0, frac1 = _u8(frac0), _u8(frac1)
    n = dst_off.size
    dst = np.zeros(dst_bytes, dtype=np.uint8)
    lib().orc_pred_bi_batch(C.c_int(taps), _p(dst), C.c_ssize_t(stride_dst), _p(ref0), _p(ref1), C.c_ssize_t(stride_ref),
                            _p(dst_off), _p(ref0_off), _p(ref1_off), _p(frac0), _p(frac1),
                            C.c_int(w), C.c_int(h), C.c_size_t(n))
    return dst


def inverse_transform_add_batch(tr_type, log2n, dst_bytes, stride_dst, pred, stride_pred, dst_off, pred_off, coeffs):
    pred, dst_off, pred_off, coeffs = _u8(pred), _u32(dst_off), _u32(pred_off), _i16(coeffs)
    n = dst_off.size
    assert coeffs.size == n << (2 * log2n)
    dst = np.zeros(dst_bytes, dtype=np.uint8)
    lib().orc_inverse_transform_add_batch(C.c_int(tr_type), C.c_int(log2n), _p(dst), C.c_ssize_t(stride_dst),
                                          _p(pred), C.c_ssize_t(stride_pred), _p(dst_off), _p(pred_off),
                                          _p(coeffs), C.c_size_t(n))
    return dst


def dequant_inverse_transform_add_batch(tr_type, log2n, dst_bytes, stride_dst, pred, stride_pred, dst_off, pred_off,
                                        levels, scale, shift):
    pred, dst_off, pred_off, levels = _u8(pred), _u32(dst_off), _u32(pred_off), _i16(levels)
    n = dst_off.size
    assert levels.size == n << (2 * log2n)
    dst = np.zeros(dst_bytes, dtype=np.uint8)
    lib().orc_dequant_inverse_transform_add_batch(C.c_int(tr_type), C.c_int(log2n), _p(dst), C.c_ssize_t(stride_dst),
                                                  _p(pred), C.c_ssize_t(stride_pred), _p(dst_off), _p(pred_off),
                                                  _p(levels), C.c_int(scale), C.c_int(shift), C.c_size_t(n))
    return dst


def transform_batch(tr_type, log2n, src, src_stride, src_off):
    src, src_off = _i16(src), _u32(src_off)
    n = src_off.size
    out = np.empty(n << (2 * log2n), dtype=np.int16)
    lib().orc_transform_batch(C.c_int(tr_type), C.c_int(log2n), _p(out), _p(src), C.c_ssize_t(src_stride),
                              _p(src_off), C.c_size_t(n))
    return out


def quantize_inverse(src, scale, shift):
    src = _i16(src)
    out = np.empty(src.size, dtype=np.int16)
    lib().orc_quantize_inverse(_p(out), _p(src), C.c_int(scale), C.c_int(shift), C.c_int(src.size))
    return out


def quantize_batch(src, scale, shift, offset, n_per_block):
    src = _i16(src)
    nb = src.size // n_per_block
    out = np.empty(src.size, dtype=np.int16)
    cbf = np.empty(nb, dtype=np.int32)
    lib().orc_quantize_batch(_p(out), _p(src), C.c_int(scale), C.c_int(shift), C.c_int(offset), C.c_int(n_per_block),
                             _p(cbf), C.c_size_t(nb))
    return out, cbf


def quantize_reconstruct_batch(rec_bytes, stride_rec, pred, stride_pred, rec_off, pred_off, res, ncbs):
    pred, rec_off, pred_off, res = _u8(pred), _u32(rec_off), _u32(pred_off), _i16(res)
    n = rec_off.size
    rec = np.zeros(rec_bytes, dtype=np.uint8)
    lib().orc_quantize_reconstruct_batch(_p(rec), C.c_ssize_t(stride_rec), _p(pred), C.c_ssize_t(stride_pred),
                                         _p(rec_off), _p(pred_off), _p(res), C.c_int(ncbs), C.c_size_t(n))
    return rec


def me_full_search(src, stride_src, ref, stride_ref, src_off, ref_off, w, h, search_range):
    src, ref, src_off, ref_off = _u8(src), _u8(ref), _u32(src_off), _u32(ref_off)
    n = src_off.size
    out = np.empty((n, 3), dtype=np.int32)
    lib().orc_me_full_search(_p(src), C.c_ssize_t(stride_src), _p(ref), C.c_ssize_t(stride_ref),
                             _p(src_off), _p(ref_off), _p(out), C.c_uint32(rect(w, h)), C.c_int(search_range), C.c_size_t(n))
    return out


def subpel_refine(src, stride_src, ref, stride_ref, src_off, ref_off, w, h, metric=0):
    src, ref, src_off, ref_off = _u8(src), _u8(ref), _u32(src_off), _u32(ref_off)
    n = src_off.size
    out = np.empty((n, 3), dtype=np.int32)
    lib().orc_subpel_refine(_p(src), C.c_ssize_t(stride_src), _p(ref), C.c_ssize_t(stride_ref),
                            _p(src_off), _p(ref_off), _p(out), C.c_uint32(rect(w, h)), C.c_int(metric), C.c_size_t(n))
    return out


def pred_intra_batch(neighbours, neighbour_stride, mode, packed, n):
    neighbours = _u8(neighbours)
    size = 1 << ((packed >> 1) & 0x7f)
    dst = np.zeros(n * size * size, dtype=np.uint8)
    lib().orc_pred_intra_batch(_p(dst), _p(neighbours), C.c_ssize_t(neighbour_stride), C.c_int(mode), C.c_int(packed), C.c_size_t(n))
    return dst


def pred_intra_modes_batch(neighbours, neighbour_stride, modes, packed):
    """UNPINNED beyond mode 1: H.265 8.4.4.2.4-6 restated (oracle/hevc_oracle.c)."""
    neighbours, modes = _u8(neighbours), _u8(modes)
    size = 1 << ((packed >> 1) & 0x7f)
    dst = np.zeros(modes.size * size * size, dtype=np.uint8)
    lib().orc_pred_intra_modes_batch(_p(dst), _p(neighbours), C.c_ssize_t(neighbour_stride), _p(modes), C.c_int(packed), C.c_size_t(modes.size))
    return dst


def intra_filter_neighbours_batch(neighbours, neighbour_stride, log2, strong, n, out_stride=129):
    """UNPINNED: H.265 8.4.4.2.3 restated."""
    neighbours = _u8(neighbours)
    out = np.zeros(out_stride * n, dtype=np.uint8)
    lib().orc_intra_filter_neighbours_batch(_p(out), C.c_ssize_t(out_stride), _p(neighbours), C.c_ssize_t(neighbour_stride), C.c_int(log2), C.c_int(strong), C.c_size_t(n))
    return out


def ssd_linear_batch(src0, src1, off0, off1, size):
    src0, src1, off0, off1 = _u8(src0), _u8(src1), _u32(off0), _u32(off1)
    out = np.empty(off0.size, dtype=np.int32)
    lib().orc_ssd_linear_batch(_p(src0), _p(src1), _p(off0), _p(off1), _p(out), C.c_int(size), C.c_size_t(off0.size))
    return out


def frame_sse_batch(a, stride_a, b, stride_b, a_off, b_off, width, height):
    a, b, a_off, b_off = _u8(a), _u8(b), _u32(a_off), _u32(b_off)
    out = np.empty(a_off.size, dtype=np.uint64)
    lib().orc_frame_sse_batch(_p(a), C.c_ssize_t(stride_a), _p(b), C.c_ssize_t(stride_b), _p(a_off), _p(b_off), _p(out), C.c_int(width), C.c_int(height),
                              C.c_size_t(a_off.size))
    return out.astype(np.int64)


def dct_matrix(n):
    f = lib().orc_dct_matrix
    return np.array([[f(n, k, i) for i in range(n)] for k in range(n)], dtype=np.int32)


def intra_reconstruct_frame(src, stride_src, levels_in, tus, qp, w, h, stride_recon, ctu, cidx, strong, want_pred=False):
    """One intra frame, TU after TU (orc_intra_reconstruct_frame).  src = None: decoder, levels_in is input.  tus: structured array / (n, 8) uint8 of
    hevcasm_hip_intra_tu; qp: int32 [4][5].  -> (recon [h * stride_recon], levels, cbf[, pred]).  PINNED only for all-DC maps."""
    tus = np.ascontiguousarray(tus).view(np.uint8).reshape(-1, 8)
    qp = np.ascontiguousarray(qp, dtype=np.int32).reshape(4, 5)
    cols, rows = -(-w // ctu), -(-h // ctu)
    levels = np.zeros(cols * rows * ctu * ctu, dtype=np.int16) if src is not None else _i16(levels_in).copy()
    recon = np.zeros(h * stride_recon, dtype=np.uint8)
    pred = np.zeros(h * stride_recon, dtype=np.uint8) if want_pred else None
    cbf = np.zeros(len(tus), dtype=np.int32)
    s = _u8(src) if src is not None else None
    f = lib().orc_intra_reconstruct_frame
    f.restype = C.c_int
    rc = f(_p(recon), C.c_ssize_t(stride_recon), _p(s) if s is not None else None, C.c_ssize_t(stride_src), _p(levels), _p(cbf),
           _p(pred) if want_pred else None, _p(tus), C.c_size_t(len(tus)), _p(qp), C.c_int(w), C.c_int(h), C.c_int(ctu), C.c_int(cidx), C.c_int(strong))
    if rc != 0:
        raise ValueError("orc_intra_reconstruct_frame: the TU list is not a decodable tiling of the %d x %d plane" % (w, h))
    return (recon, levels, cbf, pred) if want_pred else (recon, levels, cbf)


def intra_build_neighbours(recon, stride, w, h, order, cur, x, y, n):
    """The 129-byte record (layout: left [63 - y], top [64 + x], corner [128]) of one block: H.265 8.4.4.2.2 (orc_intra_build_neighbours)."""
    rec = np.zeros(160, dtype=np.uint8)
    order = np.ascontiguousarray(order, dtype=np.int32)
    r = _u8(recon)
    lib().orc_intra_build_neighbours(C.c_void_p(rec.ctypes.data), _p(r), C.c_ssize_t(stride), C.c_int(w), C.c_int(h), _p(order), C.c_int(cur), C.c_int(x), C.c_int(y), C.c_int(n))
    return rec
