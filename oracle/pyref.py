"""Client for oracle/_ref/refdrv -- the REFERENCE's own C path (TEST INFRASTRUCTURE ONLY).

refdrv is the reference's library C files compiled in place from
/root/reference by oracle/Makefile, plus our protocol driver refdrv.c.  It is
used (a) to pin the oracle restatement, (b) to generate tests/golden/, and
(c) as bench.py's ``cpu_baseline`` with ``kind: "reference"``.
The binary travels to the GPU box with the snapshot; /root/reference does not.
"""
import os
import struct
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
BINARY = os.path.join(_HERE, "_ref", "refdrv")

OP_SAD, OP_SAD4, OP_SSD, OP_PRED_UNI, OP_PRED_BI, OP_INV, OP_FWD = 1, 2, 3, 4, 5, 6, 7
OP_DEQUANT, OP_QUANT, OP_RECON, OP_DEQ_INV, OP_ME, OP_SATD, OP_SUBPEL, OP_INTRA, OP_SSD_LINEAR, OP_FRAME_SSE = 8, 9, 10, 11, 12, 13, 14, 15, 16, 17
OP_INTRA_FRAME = 18
OP_SELFTEST, OP_BENCH_SAD, OP_BENCH_PRED, OP_BENCH_INV = 20, 30, 31, 32


def available():
    return os.path.exists(BINARY) and os.access(BINARY, os.X_OK)


class RefProc:
    """One refdrv process; call() issues a request and returns the output buffers as bytes."""

    def __init__(self):
        if not available():
            raise RuntimeError("oracle/_ref/refdrv not built (run `make -C oracle` where /root/reference exists)")
        self.p = subprocess.Popen([BINARY], stdin=subprocess.PIPE, stdout=subprocess.PIPE)

    def call(self, op, params, bufs):
        msg = [struct.pack("<III", 0x51525648, op, len(params))]
        msg.append(struct.pack("<%dq" % len(params), *[int(x) for x in params]))
        msg.append(struct.pack("<I", len(bufs)))
        for b in bufs:
            raw = b if isinstance(b, (bytes, bytearray)) else np.ascontiguousarray(b).tobytes()
            msg.append(struct.pack("<Q", len(raw)))
            msg.append(raw)
        self.p.stdin.write(b"".join(msg))
        self.p.stdin.flush()
        hdr = self._read(12)
        magic, status, nout = struct.unpack("<III", hdr)
        if magic != 0x53525648 or status != 0:
            raise RuntimeError("refdrv: bad response (magic %x status %d)" % (magic, status))
        outs = []
        for _ in range(nout):
            (nbytes,) = struct.unpack("<Q", self._read(8))
            outs.append(self._read(nbytes))
        return outs

    def _read(self, n):
        chunks = []
        while n:
            c = self.p.stdout.read(n)
            if not c:
                raise RuntimeError("refdrv died (exit %s)" % self.p.poll())
            chunks.append(c)
            n -= len(c)
        return b"".join(chunks)

    def close(self):
        if self.p and self.p.poll() is None:
            self.p.stdin.close()
            self.p.wait(timeout=30)
        self.p = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- typed wrappers, argument-for-argument the same as oracle.pyoracle ----

    def sad_batch(self, src, stride_src, ref, stride_ref, src_off, ref_off, w, h):
        o = self.call(OP_SAD, [stride_src, stride_ref, (w << 8) | h, len(src_off)],
                      [_u8(src), _u8(ref), _u32(src_off), _u32(ref_off)])
        return np.frombuffer(o[0], dtype=np.int32).copy()

    def sad_multiref_batch(self, src, stride_src, ref, stride_ref, src_off, ref_off4, w, h):
        o = self.call(OP_SAD4, [stride_src, stride_ref, (w << 8) | h, len(src_off)],
                      [_u8(src), _u8(ref), _u32(src_off), _u32(ref_off4)])
        return np.frombuffer(o[0], dtype=np.int32).reshape(-1, 4).copy()

    def ssd_batch(self, a, stride_a, b, stride_b, a_off, b_off, w, h):
        o = self.call(OP_SSD, [stride_a, stride_b, w, h, len(a_off)], [_u8(a), _u8(b), _u32(a_off), _u32(b_off)])
        return np.frombuffer(o[0], dtype=np.int32).copy()

    def hadamard_satd_batch(self, n, a, stride_a, b, stride_b, a_off, b_off):
        o = self.call(OP_SATD, [stride_a, stride_b, n, len(a_off)], [_u8(a), _u8(b), _u32(a_off), _u32(b_off)])
        return np.frombuffer(o[0], dtype=np.int32).copy()

    def pred_uni_batch(self, taps, dst_bytes, stride_dst, ref, stride_ref, dst_off, ref_off, frac, w, h):
        o = self.call(OP_PRED_UNI, [stride_dst, stride_ref, taps, w, h, len(dst_off), dst_bytes],
                      [_u8(ref), _u32(dst_off), _u32(ref_off), _u8(frac)])
        return np.frombuffer(o[0], dtype=np.uint8).copy()

    def pred_bi_batch(self, taps, dst_bytes, stride_dst, ref0, ref1, stride_ref, dst_off, ref0_off, ref1_off, frac0, frac1, w, h):
        o = self.call(OP_PRED_BI, [stride_dst, stride_ref, taps, w, h, len(dst_off), dst_bytes],
                      [_u8(ref0), _u8(ref1), _u32(dst_off), _u32(ref0_off), _u32(ref1_off), _u8(frac0), _u8(frac1)])
        return np.frombuffer(o[0], dtype=np.uint8).copy()

    def inverse_transform_add_batch(self, tr_type, log2n, dst_bytes, stride_dst, pred, stride_pred, dst_off, pred_off, coeffs):
        o = self.call(OP_INV, [stride_dst, stride_pred, tr_type, log2n, len(dst_off), dst_bytes],
                      [_u8(pred), _u32(dst_off), _u32(pred_off), _i16(coeffs)])
        return np.frombuffer(o[0], dtype=np.uint8).copy()

    def dequant_inverse_transform_add_batch(self, tr_type, log2n, dst_bytes, stride_dst, pred, stride_pred, dst_off, pred_off,
                                            levels, scale, shift):
        o = self.call(OP_DEQ_INV, [stride_dst, stride_pred, tr_type, log2n, len(dst_off), dst_bytes, scale, shift],
                      [_u8(pred), _u32(dst_off), _u32(pred_off), _i16(levels)])
        return np.frombuffer(o[0], dtype=np.uint8).copy()

    def transform_batch(self, tr_type, log2n, src, src_stride, src_off):
        o = self.call(OP_FWD, [src_stride, tr_type, log2n, len(src_off)], [_i16(src), _u32(src_off)])
        return np.frombuffer(o[0], dtype=np.int16).copy()

    def quantize_inverse(self, src, scale, shift):
        src = _i16(src)
        o = self.call(OP_DEQUANT, [scale, shift, src.size], [src])
        return np.frombuffer(o[0], dtype=np.int16).copy()

    def quantize_batch(self, src, scale, shift, offset, n_per_block):
        src = _i16(src)
        o = self.call(OP_QUANT, [scale, shift, offset, n_per_block, src.size // n_per_block], [src])
        return np.frombuffer(o[0], dtype=np.int16).copy(), np.frombuffer(o[1], dtype=np.int32).copy()

    def quantize_reconstruct_batch(self, rec_bytes, stride_rec, pred, stride_pred, rec_off, pred_off, res, ncbs):
        o = self.call(OP_RECON, [stride_rec, stride_pred, ncbs, len(rec_off), rec_bytes],
                      [_u8(pred), _u32(rec_off), _u32(pred_off), _i16(res)])
        return np.frombuffer(o[0], dtype=np.uint8).copy()

    def me_full_search(self, src, stride_src, ref, stride_ref, src_off, ref_off, w, h, search_range):
        o = self.call(OP_ME, [stride_src, stride_ref, (w << 8) | h, search_range, len(src_off)],
                      [_u8(src), _u8(ref), _u32(src_off), _u32(ref_off)])
        return np.frombuffer(o[0], dtype=np.int32).reshape(-1, 3).copy()

    def subpel_refine(self, src, stride_src, ref, stride_ref, src_off, ref_off, w, h, metric=0):
        o = self.call(OP_SUBPEL, [stride_src, stride_ref, (w << 8) | h, len(src_off), metric],
                      [_u8(src), _u8(ref), _u32(src_off), _u32(ref_off)])
        return np.frombuffer(o[0], dtype=np.int32).reshape(-1, 3).copy()

    def pred_intra_batch(self, neighbours, neighbour_stride, mode, packed, n):
        o = self.call(OP_INTRA, [neighbour_stride, mode, packed, n], [_u8(neighbours)])
        return np.frombuffer(o[0], dtype=np.uint8).copy()

    def ssd_linear_batch(self, src0, src1, off0, off1, size):
        o = self.call(OP_SSD_LINEAR, [size, len(off0)], [_u8(src0), _u8(src1), _u32(off0), _u32(off1)])
        return np.frombuffer(o[0], dtype=np.int32).copy()

    def frame_sse_batch(self, a, stride_a, b, stride_b, a_off, b_off, width, height):
        o = self.call(OP_FRAME_SSE, [stride_a, stride_b, width, height, len(a_off)], [_u8(a), _u8(b), _u32(a_off), _u32(b_off)])
        return np.frombuffer(o[0], dtype=np.uint64).astype(np.int64)

    def intra_reconstruct_frame(self, src, stride_src, levels_in, tus, qp, w, h, stride_recon, ctu, cidx, strong=0, want_pred=False):
        """The reference's own functions composed over a frame (refdrv OP_INTRA_FRAME); only modes its table holds (INTRA_DC) -- any other raises."""
        assert not want_pred
        tus = np.ascontiguousarray(tus).view(np.uint8).reshape(-1, 8)
        qp = np.ascontiguousarray(qp, dtype=np.int32).reshape(-1)
        encode = src is not None
        o = self.call(OP_INTRA_FRAME, [stride_recon, stride_src, w, h, ctu, cidx, len(tus), int(encode)] + [int(v) for v in qp],
                      [_u8(src) if encode else _i16(levels_in), tus])
        return np.frombuffer(o[0], dtype=np.uint8).copy(), np.frombuffer(o[1], dtype=np.int16).copy(), np.frombuffer(o[2], dtype=np.int32).copy()

    def selftest(self, family_mask, out_path):
        o = self.call(OP_SELFTEST, [family_mask], [out_path.encode()])
        return struct.unpack("<i", o[0])[0]

    def bench_sad(self, src, stride_src, ref, stride_ref, src_off, ref_off, w, h, ways, min_ms):
        o = self.call(OP_BENCH_SAD, [stride_src, stride_ref, (w << 8) | h, len(src_off), ways, min_ms],
                      [_u8(src), _u8(ref), _u32(src_off), _u32(ref_off)])
        secs, blocks = struct.unpack("<dd", o[0][:16])
        return secs, blocks

    def bench_pred(self, taps, dst_bytes, stride_dst, refs, stride_ref, dst_off, ref_offs, fracs, w, h, min_ms):
        """Times the reference's pred_uni (one reference) / pred_bi (two) table entries over the PU list -> (seconds, PUs)."""
        bi = len(refs) == 2
        bufs = [_u8(r) for r in refs] + [_u32(dst_off)] + [_u32(o) for o in ref_offs] + [_u8(f) for f in fracs]
        o = self.call(OP_BENCH_PRED, [stride_dst, stride_ref, taps, w, h, len(dst_off), dst_bytes, int(bi), min_ms], bufs)
        return struct.unpack("<dd", o[0][:16])

    def bench_inverse(self, tr_type, log2n, dst_bytes, stride_dst, pred, stride_pred, dst_off, pred_off, coeffs, min_ms, dequant=None):
        """Times the reference's inverse_transform_add entry (dequant = (scale, shift): hevcasm_quantize_inverse in front) -> (seconds, blocks)."""
        scale, shift = dequant if dequant else (0, 0)
        o = self.call(OP_BENCH_INV, [stride_dst, stride_pred, tr_type, log2n, len(dst_off), dst_bytes, int(dequant is not None), scale, shift, min_ms],
                      [_u8(pred), _u32(dst_off), _u32(pred_off), _i16(coeffs)])
        return struct.unpack("<dd", o[0][:16])


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _i16(a):
    return np.ascontiguousarray(a, dtype=np.int16)
