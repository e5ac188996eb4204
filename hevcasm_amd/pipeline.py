"""One inter-coded frame through the whole hot path, the way an encoder's inner loop strings the reference's functions together
(a codec fills its tables once and calls per block; here every stage is ONE batch call over the frame's blocks):

    full search (hevcasm_sad over a +-range window)  ->  quarter-pel refinement (hevcasm_pred_uni_8to8 + hevcasm_sad, 49 candidates)
    ->  hevcasm_pred_uni_8to8 at the winning vector  ->  residual  ->  hevcasm_transform  ->  hevcasm_quantize (levels, cbf)
    ->  hevcasm_quantize_inverse + hevcasm_inverse_transform_add (fused)  ->  reconstruction  ->  hevcasm_ssd per block, frame SSE / PSNR

`stages(frame, ops)` runs the chain on whatever `ops` provides: `HipOps` = the batch C-ABI on device tensors (hevcasm_amd.batch; frames stay in HBM
from the first stage to the last, only the 12-byte search answers are looked at on the host to build the next stage's descriptors), or -- in tests --
an oracle-backed object with the same methods.  The one step the reference's library has no function for is the residual (current - prediction as
int16): a codec does that itself; here it is a torch / numpy subtraction.  Host-side plumbing only; no arithmetic of the hot path lives here."""
import numpy as np


def block_grid(w, h, size, stride_cur, stride_ref, pad):
    by, bx = np.meshgrid(np.arange(h // size), np.arange(w // size), indexing="ij")
    cur_off = (by * size * stride_cur + bx * size).reshape(-1).astype(np.uint32)
    ref_off = ((by * size + pad) * stride_ref + bx * size + pad).reshape(-1).astype(np.int64)
    return cur_off, ref_off


def stages(frame, ops, size=16, search_range=8, qscale=26214, qshift=21, qoffset=171 << 7, dqscale=160, dqshift=1):
    """-> dict of every intermediate (numpy): mv, qmv, pred, residual, coeffs, levels, cbf, recon, block_ssd, frame_sse, psnr."""
    w, h, pad = frame["w"], frame["h"], frame["pad"]
    sc, sr = frame["stride_cur"], frame["stride_ref"]
    log2n = size.bit_length() - 1
    cur_off, ref_base = block_grid(w, h, size, sc, sr, pad)
    n = cur_off.size
    mv = ops.me_full_search(frame["cur"], sc, frame["ref"], sr, cur_off, ref_base.astype(np.uint32), size, size, search_range)          # (dx, dy, cost)
    ref_int = ref_base + mv[:, 1].astype(np.int64) * sr + mv[:, 0]
    qmv = ops.subpel_refine(frame["cur"], sc, frame["ref"], sr, cur_off, ref_int.astype(np.uint32), size, size)                            # (qx, qy, cost), quarter pels
    ref_sub = ref_int + (qmv[:, 1] >> 2).astype(np.int64) * sr + (qmv[:, 0] >> 2)
    frac = ((qmv[:, 0] & 3) | ((qmv[:, 1] & 3) << 4)).astype(np.uint8)
    pred = ops.pred_uni(w * h, sc, frame["ref"], sr, cur_off, ref_sub.astype(np.uint32), frac, 8, size, size)                              # a plane laid out like the current frame
    residual = ops.residual(frame["cur"], pred)                                                                                             # int16 plane, stride sc
    coeffs = ops.transform(residual, sc, cur_off, 0, log2n)
    levels, cbf = ops.quantize(coeffs, qscale, qshift, qoffset, size * size)
    recon = ops.dequant_inverse_transform_add(w * h, sc, pred, sc, cur_off, cur_off, levels, dqscale, dqshift, 0, log2n)
    block_ssd = ops.ssd(frame["cur"], sc, recon, sc, cur_off, cur_off, size, size)
    frame_sse = ops.frame_sse(frame["cur"], sc, recon, sc, w, h)
    psnr = float("inf") if frame_sse == 0 else 10.0 * np.log10(255.0 * 255.0 * w * h / float(frame_sse))
    return dict(mv=mv, qmv=qmv, pred=pred, residual=residual, coeffs=coeffs, levels=levels, cbf=cbf, recon=recon, block_ssd=block_ssd,
                frame_sse=int(frame_sse), psnr=psnr, blocks=n)


class HipOps:
    """The chain's stages on the GPU: device tensors in, numpy out only where the host needs to look (the search answers) or at the very end."""

    def __init__(self, device="cuda"):
        import torch
        from . import batch
        self.t, self.B, self.device = torch, batch, device
        self._planes = {}

    def _dev(self, a, dtype):
        """Upload once per array: the frames stay resident across the stages."""
        key = id(a)
        if key not in self._planes:
            self._planes[key] = (a, self.t.from_numpy(np.ascontiguousarray(a)).to(self.device))
        return self._planes[key][1]

    def _off(self, o):
        return self.B.off32(np.ascontiguousarray(o, dtype=np.uint32))

    def me_full_search(self, cur, sc, ref, sr, cur_off, ref_off, w, h, search_range):
        out = self.B.me_full_search(self._dev(cur, None), sc, self._dev(ref, None), sr, self._off(cur_off), self._off(ref_off), w, h, search_range)
        self.t.cuda.synchronize()
        return out.cpu().numpy()

    def subpel_refine(self, cur, sc, ref, sr, cur_off, ref_off, w, h):
        out = self.B.subpel_refine(self._dev(cur, None), sc, self._dev(ref, None), sr, self._off(cur_off), self._off(ref_off), w, h)
        self.t.cuda.synchronize()
        return out.cpu().numpy()

    def pred_uni(self, dst_bytes, sd, ref, sr, dst_off, ref_off, frac, taps, w, h):
        dst = self.t.zeros(dst_bytes, dtype=self.t.uint8, device=self.device)
        return self.B.pred_uni(dst, sd, self._dev(ref, None), sr, self._off(dst_off), self._off(ref_off), self.B.u8(frac), taps, w, h)

    def residual(self, cur, pred):
        return self._dev(cur, None).to(self.t.int16) - pred.to(self.t.int16)

    def transform(self, residual, stride, off, tr_type, log2n):
        return self.B.transform(residual, stride, self._off(off), tr_type, log2n)

    def quantize(self, coeffs, scale, shift, offset, n_per_block):
        return self.B.quantize(coeffs, scale, shift, offset, n_per_block)

    def dequant_inverse_transform_add(self, dst_bytes, sd, pred, sp, dst_off, pred_off, levels, scale, shift, tr_type, log2n):
        dst = self.t.zeros(dst_bytes, dtype=self.t.uint8, device=self.device)
        return self.B.dequant_inverse_transform_add(dst, sd, pred, sp, self._off(dst_off), self._off(pred_off), levels, scale, shift, tr_type, log2n)

    def ssd(self, a, sa, b, sb, a_off, b_off, w, h):
        return self.B.ssd(self._dev(a, None), sa, b, sb, self._off(a_off), self._off(b_off), w, h)

    def frame_sse(self, a, sa, b, sb, w, h):
        zero = self._off(np.zeros(1, dtype=np.uint32))
        out = self.B.frame_sse(self._dev(a, None), sa, b, sb, zero, zero, w, h)
        self.t.cuda.synchronize()
        return int(out[0])

    @staticmethod
    def to_numpy(results):
        import torch
        torch.cuda.synchronize()
        return {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in results.items()}
