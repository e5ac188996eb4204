/*
 * hevc_oracle.h -- CPU restatement of the reference's C path for the hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * library.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this; the product (hevcasm_amd/, libhevcasm_hip.so) never
 * links, imports or calls it.
 *
 * Parity status: PINNED.  Every function here is checked bit-for-bit against
 * (1) oracle/_ref/refdrv, which is the reference's own C files compiled where
 * they lie under /root/reference (recipe: oracle/Makefile), and (2) the
 * committed fixtures under tests/golden/ that were generated from that binary
 * (tests/golden/make_golden.py).  The reference has no golden vectors of its
 * own: its tests are differential against HEVCASM_C_REF, which
 * is exactly the path refdrv runs.
 *
 * All citations are file:line into /root/reference/src/lib.
 */
#ifndef HEVC_ORACLE_H
#define HEVC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- per-block functions (same argument meaning as the reference typedefs) ---- */

/* sad.h:50, sad.c:47-60 */
int orc_sad(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref, uint32_t rect);
/* sad.h:95, sad.c:101-121 */
void orc_sad_multiref_4(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref[], ptrdiff_t stride_ref, int sad[], uint32_t rect);
/* ssd.h:53, ssd.c:43-55 */
int orc_ssd(const uint8_t *pA, ptrdiff_t strideA, const uint8_t *pB, ptrdiff_t strideB, int w, int h);

/* pred_inter.c:53-83 */
int orc_pred_coefficient(int n, int fractionalPosition, int k);
/* pred_inter.h:56; dispatch as pred_inter.c:231-283 (C branch), bodies pred_inter.c:141-228 */
void orc_pred_uni_8to8(int taps, uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *ref, ptrdiff_t stride_ref,
		int nPbW, int nPbH, int xFrac, int yFrac);
/* pred_inter.h:76; pred_inter.c:490-530 */
void orc_pred_bi_8to8(int taps, uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *ref0, const uint8_t *ref1, ptrdiff_t stride_ref,
		int nPbW, int nPbH, int xFrac0, int yFrac0, int xFrac1, int yFrac1);

/* residual_decode.h:54; residual_decode.c:69-413 */
void orc_inverse_transform_add(int trType, int log2TrafoSize, uint8_t *dst, ptrdiff_t stride_dst,
		const uint8_t *pred, ptrdiff_t stride_pred, const int16_t *coeffs);
/* residual_decode.h:82; residual_decode.c:592-892 */
void orc_transform(int trType, int log2TrafoSize, int16_t *coeffs, const int16_t *src, ptrdiff_t src_stride);

/* quantize.h:57; quantize.c:53-62 */
void orc_quantize_inverse(int16_t *dst, const int16_t *src, int scale, int shift, int n);
/* quantize.h:78; quantize.c:160-186 */
int orc_quantize(int16_t *dst, const int16_t *src, int scale, int shift, int offset, int n);
/* quantize.h:99; quantize.c:292-302 */
void orc_quantize_reconstruct(uint8_t *rec, ptrdiff_t stride_rec, const uint8_t *pred, ptrdiff_t stride_pred, const int16_t *res, int n);

/* hadamard.h:53; hadamard.c:45-111: n = 2, 4 or 8 */
int orc_hadamard_satd(int n, const uint8_t *pA, ptrdiff_t strideA, const uint8_t *pB, ptrdiff_t strideB);

/* the transform matrix the oracle generates (checked against the literals at
 * residual_decode.c:254-288 through the refdrv parity tests) */
int orc_dct_matrix(int n, int k, int i);

/* ---- batch drivers: loop the per-block function over a block list ----
 * Offsets are byte offsets (element offsets for int16 planes) from the base
 * pointer, the same descriptor the HIP C-ABI takes (include/hevcasm_hip.h). */

void orc_sad_batch(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref,
		const uint32_t *src_off, const uint32_t *ref_off, int *sad, uint32_t rect, size_t n);
void orc_sad_multiref_batch(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref,
		const uint32_t *src_off, const uint32_t *ref_off4, int *sad4, uint32_t rect, size_t n);
void orc_ssd_batch(const uint8_t *a, ptrdiff_t stride_a, const uint8_t *b, ptrdiff_t stride_b,
		const uint32_t *a_off, const uint32_t *b_off, int *ssd, int w, int h, size_t n);
void orc_pred_uni_batch(int taps, uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *ref, ptrdiff_t stride_ref,
		const uint32_t *dst_off, const uint32_t *ref_off, const uint8_t *frac, int w, int h, size_t n);
void orc_pred_bi_batch(int taps, uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *ref0, const uint8_t *ref1, ptrdiff_t stride_ref,
		const uint32_t *dst_off, const uint32_t *ref0_off, const uint32_t *ref1_off,
		const uint8_t *frac0, const uint8_t *frac1, int w, int h, size_t n);
void orc_inverse_transform_add_batch(int trType, int log2TrafoSize, uint8_t *dst, ptrdiff_t stride_dst,
		const uint8_t *pred, ptrdiff_t stride_pred, const uint32_t *dst_off, const uint32_t *pred_off,
		const int16_t *coeffs, size_t n);
/* a10 o a8: dequantise each block's levels, then inverse transform + add */
void orc_dequant_inverse_transform_add_batch(int trType, int log2TrafoSize, uint8_t *dst, ptrdiff_t stride_dst,
		const uint8_t *pred, ptrdiff_t stride_pred, const uint32_t *dst_off, const uint32_t *pred_off,
		const int16_t *levels, int scale, int shift, size_t n);
void orc_transform_batch(int trType, int log2TrafoSize, int16_t *coeffs, const int16_t *src, ptrdiff_t src_stride,
		const uint32_t *src_off, size_t n);
void orc_quantize_batch(int16_t *dst, const int16_t *src, int scale, int shift, int offset, int n_per_block, int *cbf, size_t n_blocks);
void orc_quantize_reconstruct_batch(uint8_t *rec, ptrdiff_t stride_rec, const uint8_t *pred, ptrdiff_t stride_pred,
		const uint32_t *rec_off, const uint32_t *pred_off, const int16_t *res, int nCbS, size_t n);

void orc_hadamard_satd_batch(int n, const uint8_t *a, ptrdiff_t stride_a, const uint8_t *b, ptrdiff_t stride_b,
		const uint32_t *a_off, const uint32_t *b_off, int *satd, size_t count);

/* Full search: for each block, SAD (sad.c:47-60) at every
 * displacement (dx,dy) in [-range,range]^2 around ref_off; best = minimum cost, ties broken by
 * the first in raster order of (dy,dx).  out[3*i] = dx, dy, cost. */
void orc_me_full_search(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref,
		const uint32_t *src_off, const uint32_t *ref_off, int *out3, uint32_t rect, int range, size_t n);

/* Quarter-pel refinement: a composition of the luma interpolation
 * (pred_inter.h:56, bodies pred_inter.c:141-228) and SAD (sad.c:47-60).  For each block, every
 * quarter-pel displacement (qx,qy) in [-3,3]^2 around ref_off is predicted with the 8-tap filter at
 * integer offset (qx >> 2, qy >> 2), fraction (qx & 3, qy & 3), and costed against src: metric 0 = SAD,
 * metric 1 = Hadamard SATD (hadamard.c:76-111) summed over 8x8 tiles (4x4 tiles unless both dimensions are
 * multiples of 8).
 * out[3*i] = qx, qy, cost of the minimum; ties go to the first in raster order of (qy,qx). */
void orc_subpel_refine(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref,
		const uint32_t *src_off, const uint32_t *ref_off, int *out3, uint32_t rect, int metric, size_t n);

/* pred_intra.h:55; INTRA_DC body pred_intra.c:61-101 (sample addressing pred_intra.c:52-58) */
void orc_pred_intra_dc(uint8_t *dst, const uint8_t *neighbours, int packed);
void orc_pred_intra_batch(uint8_t *dst, const uint8_t *neighbours, ptrdiff_t neighbour_stride, int intraPredMode, int packed, size_t n);
/* parity UNPINNED beyond mode 1 (H.265 8.4.4.2.3-6 restated; the reference has these modes only as assembly): see hevc_oracle.c */
void orc_pred_intra(uint8_t *dst, const uint8_t *nb, int mode, int packed);
void orc_pred_intra_modes_batch(uint8_t *dst, const uint8_t *neighbours, ptrdiff_t neighbour_stride, const uint8_t *modes, int packed, size_t n);
void orc_intra_filter_neighbours_batch(uint8_t *out, ptrdiff_t out_stride, const uint8_t *neighbours, ptrdiff_t neighbour_stride, int log2, int strong, size_t n);

/* diff.h:52; diff.c:45-54 */
int orc_ssd_linear(const uint8_t *p1, const uint8_t *p2, int n);
void orc_ssd_linear_batch(const uint8_t *src0, const uint8_t *src1, const uint32_t *off0, const uint32_t *off1, int *ssd, int size, size_t n);
void orc_frame_sse_batch(const uint8_t *a, ptrdiff_t stride_a, const uint8_t *b, ptrdiff_t stride_b, const uint32_t *a_off, const uint32_t *b_off,
                         uint64_t *sse, int width, int height, size_t n);

/* ---- a whole intra frame: neighbour construction + substitution (H.265 8.4.4.2.2), smoothing decision (8.4.4.2.3),
 * prediction, and the reconstruction loop residual -> hevcasm_transform (residual_decode.h:82) -> hevcasm_quantize (quantize.h:78) ->
 * hevcasm_quantize_inverse (quantize.h:57) -> hevcasm_inverse_transform_add (residual_decode.h:54), TU after TU in decoding order.
 * The reference has none of the frame-level steps (its self-test feeds hevcasm_pred_intra random neighbours, pred_intra.c:196-227): they follow
 * the text.  Pinned part: a frame whose TUs all use INTRA_DC -- every arithmetic step is then one of the reference's own C functions, composed by
 * oracle/refdrv.c OP_INTRA_FRAME.  Frames with other modes: UNPINNED, like orc_pred_intra beyond mode 1.
 * Same descriptors as include/hevcasm_hip.h (hevcasm_hip_intra_tu / _qp). */
typedef struct { uint16_t x, y; uint8_t log2, mode; uint16_t reserved; } orc_intra_tu;
typedef struct { int q_scale, q_shift, q_offset, dq_scale, dq_shift; } orc_intra_qp;

/* the 4 nTbS + 1 neighbouring samples of the nTbS x nTbS block at (x, y) of `recon`, in the reference's record layout around nb (left column
 * upwards ending at nb[63], top row from nb[64], corner at nb[128]); `order` holds, per 4 x 4 cell of the frame (row-major, (frame_w + 3) / 4 per row),
 * the decoding index of the TU that covers it or -1; a sample is available when it lies inside the picture in a cell whose index is below `cur`. */
void orc_intra_build_neighbours(uint8_t *nb, const uint8_t *recon, ptrdiff_t stride, int frame_w, int frame_h, const int32_t *order, int cur, int x, int y, int n);
/* 8.4.4.2.3's filterFlag for a luma block (0 for chroma planes of 4:2:0) */
int orc_intra_filter_flag(int cIdx, int mode, int n);
/* offset (in coefficients) of the TU at (x, y) in the frame's level buffer: CTUs in raster order, ctu * ctu coefficients each, 4 x 4 cells in z-scan order inside */
size_t orc_intra_level_offset(int x, int y, int frame_w, int ctu);
/* returns 0, or -1 when the TU list is not a decodable tiling (a TU outside the picture / misaligned / overlapping another, or cells left uncovered).
 * src == NULL: decoder -- `levels` is input, cbf is not written.  pred_out (optional): the prediction of every TU, same geometry as recon. */
int orc_intra_reconstruct_frame(uint8_t *recon, ptrdiff_t stride_recon, const uint8_t *src, ptrdiff_t stride_src, int16_t *levels, int *cbf,
		uint8_t *pred_out, const orc_intra_tu *tus, size_t n_tus, const orc_intra_qp *qp /* [4], by log2 - 2 */,
		int frame_w, int frame_h, int ctu, int cIdx, int strong);

#ifdef __cplusplus
}
#endif
#endif
