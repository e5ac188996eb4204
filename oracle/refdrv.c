/*
 * refdrv.c -- request/response driver around the REFERENCE's own C path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is ours; it is compiled together with
 * the reference's library C files where they lie under /root/reference/src/lib
 * (recipe: oracle/Makefile, output only into oracle/_ref/).  Every kernel call
 * below goes through the reference's public API exactly as a codec would use
 * it: hevcasm_populate_*(&table, HEVCASM_C_REF) then
 * (*hevcasm_get_*(...))(...).  The mask is always HEVCASM_C_REF, so the x86
 * asm entry points (which cannot be assembled in this image: no yasm/nasm)
 * are never selected or called; the link leaves them unresolved
 * (-Wl,--unresolved-symbols=ignore-all) rather than faking them.
 *
 * Protocol (little endian, stdin -> stdout), one or more requests per process:
 *   request : u32 magic 'HVRQ', u32 op, u32 nparams, i64 params[nparams],
 *             u32 nbufs, { u64 bytes, data } * nbufs
 *   response: u32 magic 'HVRS', u32 status, u32 nbufs, { u64 bytes, data } * nbufs
 * The batch ops take the same descriptors as oracle/hevc_oracle.h's batch
 * drivers and include/hevcasm_hip.h (base pointer + per-block u32 offsets).
 */
#include "hevcasm.h"
#include "sad.h"
#include "ssd.h"
#include "pred_inter.h"
#include "residual_decode.h"
#include "quantize.h"
#include "hadamard.h"
#include "pred_intra.h"
#include "diff.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define MAXBUF 8
#define MAXPAR 40

typedef struct { uint64_t bytes; uint8_t *p; } buf_t;

static int rd(void *p, size_t n) { return n == 0 || fread(p, 1, n, stdin) == n; }
static void wr(const void *p, size_t n) { if (n) fwrite(p, 1, n, stdout); }

static hevcasm_table_sad t_sad;
static hevcasm_table_sad_multiref t_sad4;
static hevcasm_table_ssd t_ssd;
static hevcasm_table_pred_uni_8to8 t_uni;
static hevcasm_table_pred_bi_8to8 t_bi;
static hevcasm_table_inverse_transform_add t_inv;
static hevcasm_table_transform t_fwd;
static hevcasm_table_quantize_inverse t_deq;
static hevcasm_table_quantize t_q;
static hevcasm_table_quantize_reconstruct t_rec;
static hevcasm_table_hadamard_satd t_satd;
static hevcasm_table_pred_intra t_intra;

static void populate_all(void)
{
	const hevcasm_instruction_set m = HEVCASM_C_REF;
	hevcasm_populate_sad(&t_sad, m);
	hevcasm_populate_sad_multiref(&t_sad4, m);
	hevcasm_populate_ssd(&t_ssd, m);
	hevcasm_populate_pred_uni_8to8(&t_uni, m);
	hevcasm_populate_pred_bi_8to8(&t_bi, m);
	hevcasm_populate_inverse_transform_add(&t_inv, m, 0);
	hevcasm_populate_transform(&t_fwd, m);
	hevcasm_populate_quantize_inverse(&t_deq, m);
	hevcasm_populate_quantize(&t_q, m);
	hevcasm_populate_quantize_reconstruct(&t_rec, m);
	hevcasm_populate_hadamard_satd(&t_satd, m);
	hevcasm_populate_pred_intra(&t_intra, m);
}

static int ilog2(int n) { int l = 0; while ((1 << l) < n) ++l; return l; }

static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

enum {
	OP_SAD = 1, OP_SAD4 = 2, OP_SSD = 3, OP_PRED_UNI = 4, OP_PRED_BI = 5, OP_INV = 6, OP_FWD = 7,
	OP_DEQUANT = 8, OP_QUANT = 9, OP_RECON = 10, OP_DEQ_INV = 11, OP_ME = 12, OP_SATD = 13, OP_SUBPEL = 14, OP_INTRA = 15, OP_SSD_LINEAR = 16, OP_FRAME_SSE = 17,
	OP_INTRA_FRAME = 18,
	OP_SELFTEST = 20, OP_BENCH_SAD = 30, OP_BENCH_PRED = 31, OP_BENCH_INV = 32
};

/* OP_INTRA_FRAME's own plumbing (ours, not the reference's: the reference never builds a neighbour record -- its self-test feeds random bytes,
 * pred_intra.c:196-227).  H.265 8.4.4.2.2 by the letter: a sample is available when it lies in the picture in a 4x4 cell that has been reconstructed;
 * if none is, everything is 1 << (BitDepth - 1); else the scan runs from p[-1][2N-1] up the left column, over the corner and along the top row, and every
 * unavailable sample takes the value of the one before it (the first one: of the first available sample found by that scan). */
static uint8_t frame_sample_or_flag(const uint8_t *recon, ptrdiff_t stride, const uint8_t *decoded, int cells_w, int w, int h, int px, int py, int *available)
{
	*available = px >= 0 && py >= 0 && px < w && py < h && decoded[(py / 4) * cells_w + px / 4];
	return *available ? recon[px + py * stride] : 0;
}

static void frame_neighbours(uint8_t *nb, const uint8_t *recon, ptrdiff_t stride, const uint8_t *decoded, int cells_w, int w, int h, int x0, int y0, int n)
{
	int found = 0, available;
	uint8_t last = 0;
	/* pass 1: is anything available, and what does the scan meet first */
	for (int y = 2 * n - 1; y >= -1 && !found; --y) { last = frame_sample_or_flag(recon, stride, decoded, cells_w, w, h, x0 - 1, y0 + y, &available); found = available; }
	for (int x = 0; x < 2 * n && !found; ++x) { last = frame_sample_or_flag(recon, stride, decoded, cells_w, w, h, x0 + x, y0 - 1, &available); found = available; }
	if (!found) last = 128;
	/* pass 2: the scan itself */
	for (int y = 2 * n - 1; y >= -1; --y)
	{
		const uint8_t v = frame_sample_or_flag(recon, stride, decoded, cells_w, w, h, x0 - 1, y0 + y, &available);
		if (available) last = v;
		nb[y < 0 ? 128 : 63 - y] = last;        /* pred_intra.c:59-65: left at [63 - y]; the corner p[-1][-1] at [128] (f265/intra.asm:126-131) */
	}
	for (int x = 0; x < 2 * n; ++x)
	{
		const uint8_t v = frame_sample_or_flag(recon, stride, decoded, cells_w, w, h, x0 + x, y0 - 1, &available);
		if (available) last = v;
		nb[64 + x] = last;
	}
}

int main(void)
{
	populate_all();

	for (;;)
	{
		uint32_t magic, op, npar, nbuf;
		int64_t par[MAXPAR] = { 0 };
		buf_t in[MAXBUF] = { { 0, 0 } };
		buf_t out[3] = { { 0, 0 }, { 0, 0 }, { 0, 0 } };
		uint32_t nout = 0, status = 0;

		if (!rd(&magic, 4)) return 0; /* EOF: done */
		if (magic != 0x51525648u) return 2;
		if (!rd(&op, 4) || !rd(&npar, 4) || npar > MAXPAR || !rd(par, 8 * npar)) return 2;
		if (!rd(&nbuf, 4) || nbuf > MAXBUF) return 2;
		for (uint32_t i = 0; i < nbuf; ++i)
		{
			if (!rd(&in[i].bytes, 8)) return 2;
			in[i].p = (uint8_t *)malloc(in[i].bytes + 64);
			if (!in[i].p || !rd(in[i].p, in[i].bytes)) return 2;
		}

		switch (op)
		{
		case OP_SAD: /* par: stride_src, stride_ref, rect, n; bufs: src, ref, src_off, ref_off */
		{
			const size_t n = (size_t)par[3];
			const uint32_t *so = (const uint32_t *)in[2].p, *ro = (const uint32_t *)in[3].p;
			const int w = (int)(par[2] >> 8), h = (int)(par[2] & 0xff);
			hevcasm_sad *f = *hevcasm_get_sad(&t_sad, w, h);
			int *r = (int *)malloc(4 * n + 4);
			for (size_t i = 0; i < n; ++i)
				r[i] = f(in[0].p + so[i], par[0], in[1].p + ro[i], par[1], (uint32_t)par[2]);
			out[0].p = (uint8_t *)r; out[0].bytes = 4 * n; nout = 1;
			break;
		}
		case OP_SAD4: /* as OP_SAD with ref_off[4n] -> sad[4n] */
		{
			const size_t n = (size_t)par[3];
			const uint32_t *so = (const uint32_t *)in[2].p, *ro = (const uint32_t *)in[3].p;
			const int w = (int)(par[2] >> 8), h = (int)(par[2] & 0xff);
			hevcasm_sad_multiref *f = *hevcasm_get_sad_multiref(&t_sad4, 4, w, h);
			int *r = (int *)malloc(16 * n + 4);
			for (size_t i = 0; i < n; ++i)
			{
				const uint8_t *refs[4];
				for (int k = 0; k < 4; ++k) refs[k] = in[1].p + ro[4 * i + k];
				f(in[0].p + so[i], par[0], refs, par[1], r + 4 * i, (uint32_t)par[2]);
			}
			out[0].p = (uint8_t *)r; out[0].bytes = 16 * n; nout = 1;
			break;
		}
		case OP_SSD: /* par: stride_a, stride_b, w, h, n; bufs a, b, a_off, b_off */
		{
			const size_t n = (size_t)par[4];
			const uint32_t *ao = (const uint32_t *)in[2].p, *bo = (const uint32_t *)in[3].p;
			hevcasm_ssd *f = *hevcasm_get_ssd(&t_ssd, ilog2((int)par[2]));
			int *r = (int *)malloc(4 * n + 4);
			for (size_t i = 0; i < n; ++i)
				r[i] = f(in[0].p + ao[i], par[0], in[1].p + bo[i], par[1], (int)par[2], (int)par[3]);
			out[0].p = (uint8_t *)r; out[0].bytes = 4 * n; nout = 1;
			break;
		}
		case OP_SATD: /* par: stride_a, stride_b, n (2|4|8), count; bufs a, b, a_off, b_off */
		{
			const size_t n = (size_t)par[3];
			const uint32_t *ao = (const uint32_t *)in[2].p, *bo = (const uint32_t *)in[3].p;
			hevcasm_hadamard_satd *f = *hevcasm_get_hadamard_satd(&t_satd, ilog2((int)par[2]));
			int *r = (int *)malloc(4 * n + 4);
			for (size_t i = 0; i < n; ++i) r[i] = f(in[0].p + ao[i], par[0], in[1].p + bo[i], par[1]);
			out[0].p = (uint8_t *)r; out[0].bytes = 4 * n; nout = 1;
			break;
		}
		case OP_PRED_UNI: /* par: stride_dst, stride_ref, taps, w, h, n, dst_bytes; bufs ref, dst_off, ref_off, frac */
		{
			const size_t n = (size_t)par[5];
			const uint32_t *dof = (const uint32_t *)in[1].p, *rof = (const uint32_t *)in[2].p;
			const uint8_t *fr = in[3].p;
			uint8_t *dst = (uint8_t *)calloc((size_t)par[6] + 64, 1);
			for (size_t i = 0; i < n; ++i)
			{
				const int xf = fr[i] & 15, yf = fr[i] >> 4;
				hevcasm_pred_uni_8to8 *f = *hevcasm_get_pred_uni_8to8(&t_uni, (int)par[2], (int)par[3], (int)par[4], xf, yf);
				f(dst + dof[i], par[0], in[0].p + rof[i], par[1], (int)par[3], (int)par[4], xf, yf);
			}
			out[0].p = dst; out[0].bytes = (uint64_t)par[6]; nout = 1;
			break;
		}
		case OP_PRED_BI: /* par: stride_dst, stride_ref, taps, w, h, n, dst_bytes; bufs ref0, ref1, dst_off, ref0_off, ref1_off, frac0, frac1 */
		{
			const size_t n = (size_t)par[5];
			const uint32_t *dof = (const uint32_t *)in[2].p, *r0 = (const uint32_t *)in[3].p, *r1 = (const uint32_t *)in[4].p;
			const uint8_t *f0 = in[5].p, *f1 = in[6].p;
			uint8_t *dst = (uint8_t *)calloc((size_t)par[6] + 64, 1);
			for (size_t i = 0; i < n; ++i)
			{
				const int xa = f0[i] & 15, ya = f0[i] >> 4, xb = f1[i] & 15, yb = f1[i] >> 4;
				hevcasm_pred_bi_8to8 *f = *hevcasm_get_pred_bi_8to8(&t_bi, (int)par[2], (int)par[3], (int)par[4], xa, ya, xb, yb);
				f(dst + dof[i], par[0], in[0].p + r0[i], in[1].p + r1[i], par[1], (int)par[3], (int)par[4], xa, ya, xb, yb);
			}
			out[0].p = dst; out[0].bytes = (uint64_t)par[6]; nout = 1;
			break;
		}
		case OP_INV:     /* par: stride_dst, stride_pred, trType, log2, n, dst_bytes; bufs pred, dst_off, pred_off, coeffs */
		case OP_DEQ_INV: /* same + par[6] = scale, par[7] = shift; coeffs are levels */
		{
			const size_t n = (size_t)par[4];
			const int log2n = (int)par[3];
			const size_t nn = (size_t)1 << (2 * log2n);
			const uint32_t *dof = (const uint32_t *)in[1].p, *pof = (const uint32_t *)in[2].p;
			const int16_t *c = (const int16_t *)in[3].p;
			hevcasm_inverse_transform_add *f = *hevcasm_get_inverse_transform_add(&t_inv, (int)par[2], log2n);
			hevcasm_quantize_inverse *dq = *hevcasm_get_quantize_inverse(&t_deq);
			uint8_t *dst = (uint8_t *)calloc((size_t)par[5] + 64, 1);
			for (size_t i = 0; i < n; ++i)
			{
				HEVCASM_ALIGN(32, int16_t, tmp[32 * 32]);
				const int16_t *ci = c + nn * i;
				if (op == OP_DEQ_INV)
				{
					dq(tmp, ci, (int)par[6], (int)par[7], (int)nn);
					ci = tmp;
				}
				f(dst + dof[i], par[0], in[0].p + pof[i], par[1], ci);
			}
			out[0].p = dst; out[0].bytes = (uint64_t)par[5]; nout = 1;
			break;
		}
		case OP_FWD: /* par: src_stride, trType, log2, n; bufs src(int16), src_off(elements) */
		{
			const size_t n = (size_t)par[3];
			const int log2n = (int)par[2];
			const size_t nn = (size_t)1 << (2 * log2n);
			const uint32_t *sof = (const uint32_t *)in[1].p;
			hevcasm_transform *f = *hevcasm_get_transform(&t_fwd, (int)par[1], log2n);
			int16_t *r = (int16_t *)malloc(2 * nn * n + 64);
			for (size_t i = 0; i < n; ++i)
				f(r + nn * i, (const int16_t *)in[0].p + sof[i], par[0]);
			out[0].p = (uint8_t *)r; out[0].bytes = 2 * nn * n; nout = 1;
			break;
		}
		case OP_DEQUANT: /* par: scale, shift, n; buf src */
		{
			const size_t n = (size_t)par[2];
			int16_t *r = (int16_t *)malloc(2 * n + 64);
			(*hevcasm_get_quantize_inverse(&t_deq))(r, (const int16_t *)in[0].p, (int)par[0], (int)par[1], (int)n);
			out[0].p = (uint8_t *)r; out[0].bytes = 2 * n; nout = 1;
			break;
		}
		case OP_QUANT: /* par: scale, shift, offset, n_per_block, n_blocks; buf src -> dst, cbf[n_blocks] */
		{
			const size_t npb = (size_t)par[3], nb = (size_t)par[4];
			int16_t *r = (int16_t *)malloc(2 * npb * nb + 64);
			int *cbf = (int *)malloc(4 * nb + 4);
			for (size_t i = 0; i < nb; ++i)
				cbf[i] = (*hevcasm_get_quantize(&t_q))(r + npb * i, (const int16_t *)in[0].p + npb * i, (int)par[0], (int)par[1], (int)par[2], (int)npb);
			out[0].p = (uint8_t *)r; out[0].bytes = 2 * npb * nb;
			out[1].p = (uint8_t *)cbf; out[1].bytes = 4 * nb; nout = 2;
			break;
		}
		case OP_RECON: /* par: stride_rec, stride_pred, nCbS, n, rec_bytes; bufs pred, rec_off, pred_off, res */
		{
			const size_t n = (size_t)par[3];
			const int nc = (int)par[2];
			const uint32_t *rof = (const uint32_t *)in[1].p, *pof = (const uint32_t *)in[2].p;
			hevcasm_quantize_reconstruct *f = *hevcasm_get_quantize_reconstruct(&t_rec, ilog2(nc));
			uint8_t *rec = (uint8_t *)calloc((size_t)par[4] + 64, 1);
			for (size_t i = 0; i < n; ++i)
				f(rec + rof[i], par[0], in[0].p + pof[i], par[1], (const int16_t *)in[3].p + (size_t)nc * nc * i, nc);
			out[0].p = rec; out[0].bytes = (uint64_t)par[4]; nout = 1;
			break;
		}
		case OP_ME: /* par: stride_src, stride_ref, rect, range, n; bufs src, ref, src_off, ref_off -> int[3n] (dx, dy, cost) */
		{
			const size_t n = (size_t)par[4];
			const int range = (int)par[3];
			const uint32_t *so = (const uint32_t *)in[2].p, *ro = (const uint32_t *)in[3].p;
			const int w = (int)(par[2] >> 8), h = (int)(par[2] & 0xff);
			hevcasm_sad *f = *hevcasm_get_sad(&t_sad, w, h);
			int *r = (int *)malloc(12 * n + 4);
			for (size_t i = 0; i < n; ++i)
			{
				int best = 0x7fffffff, bx = 0, by = 0;
				for (int dy = -range; dy <= range; ++dy)
					for (int dx = -range; dx <= range; ++dx)
					{
						const int c = f(in[0].p + so[i], par[0], in[1].p + (ptrdiff_t)ro[i] + dy * par[1] + dx, par[1], (uint32_t)par[2]);
						if (c < best) { best = c; bx = dx; by = dy; }
					}
				r[3 * i] = bx; r[3 * i + 1] = by; r[3 * i + 2] = best;
			}
			out[0].p = (uint8_t *)r; out[0].bytes = 12 * n; nout = 1;
			break;
		}
		case OP_SUBPEL: /* par: stride_src, stride_ref, rect, n, metric; bufs src, ref, src_off, ref_off -> int[3n] (qx, qy, cost).
		                   The reference's own luma interpolation and SAD, composed per quarter-pel candidate. */
		{
			const size_t n = (size_t)par[3];
			const uint32_t *so = (const uint32_t *)in[2].p, *ro = (const uint32_t *)in[3].p;
			const int w = (int)(par[2] >> 8), h = (int)(par[2] & 0xff);
			hevcasm_sad *sad = *hevcasm_get_sad(&t_sad, w, h);
			int *r = (int *)malloc(12 * n + 4);
			uint8_t *cand = (uint8_t *)malloc(64 * 64 + 64);
			for (size_t i = 0; i < n; ++i)
			{
				int best = 0x7fffffff, bx = 0, by = 0;
				for (int qy = -3; qy <= 3; ++qy)
					for (int qx = -3; qx <= 3; ++qx)
					{
						const int xf = qx & 3, yf = qy & 3;
						hevcasm_pred_uni_8to8 *f = *hevcasm_get_pred_uni_8to8(&t_uni, 8, w, h, xf, yf);
						f(cand, 64, in[1].p + (ptrdiff_t)ro[i] + (qy >> 2) * par[1] + (qx >> 2), par[1], w, h, xf, yf);
						int c = 0;
						if (par[4] == 0) c = sad(in[0].p + so[i], par[0], cand, 64, (uint32_t)par[2]);
						else
						{
							const int tn = (w % 8 == 0 && h % 8 == 0) ? 8 : 4;
							hevcasm_hadamard_satd *satd = *hevcasm_get_hadamard_satd(&t_satd, tn == 8 ? 3 : 2);
							for (int y = 0; y < h; y += tn)
								for (int x = 0; x < w; x += tn)
									c += satd(in[0].p + so[i] + y * par[0] + x, par[0], cand + y * 64 + x, 64);
						}
						if (c < best) { best = c; bx = qx; by = qy; }
					}
				r[3 * i] = bx; r[3 * i + 1] = by; r[3 * i + 2] = best;
			}
			free(cand);
			out[0].p = (uint8_t *)r; out[0].bytes = 12 * n; nout = 1;
			break;
		}
		case OP_INTRA: /* par: neighbour_stride, intraPredMode, packed, n; bufs: neighbours -> dst[n * N * N] */
		{
			const size_t n = (size_t)par[3];
			const int size = 1 << ((par[2] >> 1) & 0x7f);
			hevcasm_pred_intra *f = *hevcasm_get_pred_intra(&t_intra, (int)par[1], (int)par[2]);
			uint8_t *dst = (uint8_t *)calloc(n * size * size + 64, 1);
			if (!f) { status = 4; free(dst); break; }
			for (size_t i = 0; i < n; ++i) f(dst + i * size * size, in[0].p + i * par[0], (int)par[1], (int)par[2]);
			out[0].p = dst; out[0].bytes = n * size * size; nout = 1;
			break;
		}
		case OP_SSD_LINEAR: /* par: size, n; bufs: src0, src1, off0, off1 -> int[n] */
		{
			const size_t n = (size_t)par[1];
			const uint32_t *o0 = (const uint32_t *)in[2].p, *o1 = (const uint32_t *)in[3].p;
			hevcasm_ssd_linear *f = hevcasm_get_ssd_linear((int)par[0], HEVCASM_C_REF);
			int *r = (int *)malloc(4 * n + 4);
			for (size_t i = 0; i < n; ++i) r[i] = f(in[0].p + o0[i], in[1].p + o1[i], (int)par[0]);
			out[0].p = (uint8_t *)r; out[0].bytes = 4 * n; nout = 1;
			break;
		}
		case OP_FRAME_SSE: /* par: stride_a, stride_b, width, height, n; bufs: a, b, a_off, b_off -> u64[n]: the reference's hevcasm_ssd_linear per row, summed in 64 bits */
		{
			const size_t n = (size_t)par[4];
			const uint32_t *o0 = (const uint32_t *)in[2].p, *o1 = (const uint32_t *)in[3].p;
			hevcasm_ssd_linear *f = hevcasm_get_ssd_linear((int)par[2], HEVCASM_C_REF);
			uint64_t *r = (uint64_t *)malloc(8 * n + 8);
			for (size_t i = 0; i < n; ++i)
			{
				uint64_t sum = 0;
				for (int64_t y = 0; y < par[3]; ++y) sum += (uint64_t)(unsigned)f(in[0].p + o0[i] + y * par[0], in[1].p + o1[i] + y * par[1], (int)par[2]);
				r[i] = sum;
			}
			out[0].p = (uint8_t *)r; out[0].bytes = 8 * n; nout = 1;
			break;
		}
		case OP_INTRA_FRAME: /* one intra frame reconstructed TU by TU with the reference's own functions wherever it has one:
		                        hevcasm_pred_intra (table; pred_intra.c:113-139 wires INTRA_DC only -> status 4 for any other mode),
		                        hevcasm_transform, hevcasm_quantize, hevcasm_quantize_inverse, hevcasm_inverse_transform_add.
		                        par: stride_recon, stride_src, frame_w, frame_h, ctu, cIdx, n_tus, encode(1|0), then 4 x { q_scale, q_shift, q_offset, dq_scale, dq_shift };
		                        bufs: src plane (encode) or levels (decode), tus (8 bytes each: u16 x, u16 y, u8 log2, u8 mode, u16 0)
		                        -> recon plane [frame_h * stride_recon], levels, cbf[n_tus] */
		{
			const ptrdiff_t sr = par[0], ss = par[1];
			const int w = (int)par[2], h = (int)par[3], ctu = (int)par[4], cIdx = (int)par[5], encode = (int)par[7];
			const size_t n_tus = (size_t)par[6];
			const int cells_w = (w + 3) / 4, cells_h = (h + 3) / 4, cols = (w + ctu - 1) / ctu, rows = (h + ctu - 1) / ctu;
			const size_t n_levels = (size_t)cols * rows * ctu * ctu;
			uint8_t *recon = (uint8_t *)calloc((size_t)h * sr + 64, 1);
			uint8_t *decoded = (uint8_t *)calloc((size_t)cells_w * cells_h + 1, 1);
			int16_t *levels = (int16_t *)calloc(n_levels + 32, 2);
			int *cbf = (int *)calloc(n_tus + 1, 4);
			if (!encode) memcpy(levels, in[0].p, in[0].bytes < 2 * n_levels ? in[0].bytes : 2 * n_levels);
			for (size_t i = 0; i < n_tus && !status; ++i)
			{
				const uint8_t *d = in[1].p + 8 * i;
				const int x = d[0] | (d[1] << 8), y = d[2] | (d[3] << 8), log2n = d[4], mode = d[5], n = 1 << log2n;
				const int packed = hevcasm_pred_intra_pack(cIdx, log2n);
				hevcasm_pred_intra *predict = (mode >= 0 && mode < 35 && log2n >= 2 && log2n <= 5) ? *hevcasm_get_pred_intra(&t_intra, mode, packed) : 0;
				if (!predict) { status = 4; break; }
				HEVCASM_ALIGN(32, uint8_t, record[192]);
				HEVCASM_ALIGN(32, uint8_t, pred[32 * 32]);
				HEVCASM_ALIGN(32, int16_t, res[32 * 32]);
				HEVCASM_ALIGN(32, int16_t, coef[32 * 32]);
				HEVCASM_ALIGN(32, int16_t, deq[32 * 32]);
				memset(record, 0, sizeof(record));
				frame_neighbours(record + 32, recon, sr, decoded, cells_w, w, h, x, y, n);
				predict(pred, record + 32, mode, packed);
				const int trType = cIdx == 0 && log2n == 2;
				/* level buffer: CTUs in raster order, 4x4 cells in z-scan order inside (include/hevcasm_hip.h) */
				unsigned z = 0;
				for (int b = 0; b < 4; ++b) z |= ((((unsigned)(x % ctu) >> 2) >> b) & 1u) << (2 * b) | ((((unsigned)(y % ctu) >> 2) >> b) & 1u) << (2 * b + 1);
				int16_t *lev = levels + ((size_t)(y / ctu) * cols + x / ctu) * ctu * ctu + 16u * z;
				const int64_t *q = par + 8 + 5 * (log2n - 2);
				if (encode)
				{
					for (int r = 0; r < n; ++r)
						for (int c = 0; c < n; ++c) res[r * n + c] = (int16_t)(in[0].p[x + c + (y + r) * ss] - pred[r * n + c]);
					(*hevcasm_get_transform(&t_fwd, trType, log2n))(coef, res, n);
					cbf[i] = (*hevcasm_get_quantize(&t_q))(lev, coef, (int)q[0], (int)q[1], (int)q[2], n * n);
				}
				(*hevcasm_get_quantize_inverse(&t_deq))(deq, lev, (int)q[3], (int)q[4], n * n);
				(*hevcasm_get_inverse_transform_add(&t_inv, trType, log2n))(recon + x + y * sr, sr, pred, n, deq);
				for (int cy = y / 4; cy < (y + n) / 4; ++cy)
					for (int cx = x / 4; cx < (x + n) / 4; ++cx) decoded[cy * cells_w + cx] = 1;
			}
			free(decoded);
			out[0].p = recon; out[0].bytes = (uint64_t)h * sr;
			out[1].p = (uint8_t *)levels; out[1].bytes = 2 * n_levels;
			out[2].p = (uint8_t *)cbf; out[2].bytes = 4 * n_tus;
			nout = 3;
			break;
		}
		case OP_SELFTEST: /* par[0] = bit mask of families; runs the reference's own self-tests on its C path. Output on stderr-free stdout is
		                     not possible here (stdout carries the protocol) so the text goes to the file named by buf 0. */
		{
			int errors = 0;
			const hevcasm_instruction_set m = (hevcasm_instruction_set)(HEVCASM_C_REF | HEVCASM_C_OPT);
			char path[512] = { 0 };
			memcpy(path, in[0].p, in[0].bytes < 511 ? in[0].bytes : 511);
			fflush(stdout);
			FILE *keep = stdout;
			FILE *f = fopen(path, "w");
			if (!f) { status = 3; break; }
			stdout = f; /* glibc: stdout is an assignable variable */
			if (par[0] & 1) hevcasm_test_sad(&errors, m);
			if (par[0] & 2) hevcasm_test_sad_multiref(&errors, m);
			if (par[0] & 4) hevcasm_test_ssd(&errors, m);
			if (par[0] & 8) hevcasm_test_pred_uni(&errors, m);
			if (par[0] & 16) hevcasm_test_pred_bi(&errors, m);
			if (par[0] & 32) hevcasm_test_inverse_transform_add(&errors, m);
			if (par[0] & 64) hevcasm_test_transform(&errors, m);
			if (par[0] & 128) hevcasm_test_quantize_inverse(&errors, m);
			if (par[0] & 256) hevcasm_test_quantize(&errors, m);
			if (par[0] & 512) hevcasm_test_quantize_reconstruct(&errors, m);
			if (par[0] & 1024) hevcasm_test_hadamard_satd(&errors, m);
			if (par[0] & 2048) hevcasm_test_pred_intra(&errors, m);
			if (par[0] & 4096) hevcasm_test_ssd_linear(&errors, m);
			fflush(f);
			stdout = keep;
			fclose(f);
			int *r = (int *)malloc(4);
			r[0] = errors;
			out[0].p = (uint8_t *)r; out[0].bytes = 4; nout = 1;
			break;
		}
		case OP_BENCH_SAD: /* CPU baseline leg of bench.py: par: stride_src, stride_ref, rect, n, ways(1|4), min_seconds;
		                      bufs as OP_SAD/OP_SAD4.  Repeats whole passes over the block list until min_seconds elapsed.
		                      -> double[2] = { seconds, blocks processed } and i64 checksum */
		{
			const size_t n = (size_t)par[3];
			const uint32_t *so = (const uint32_t *)in[2].p, *ro = (const uint32_t *)in[3].p;
			const int w = (int)(par[2] >> 8), h = (int)(par[2] & 0xff);
			hevcasm_sad *f1 = *hevcasm_get_sad(&t_sad, w, h);
			hevcasm_sad_multiref *f4 = *hevcasm_get_sad_multiref(&t_sad4, 4, w, h);
			const double min_s = (double)par[5] * 1e-3;
			double *r = (double *)malloc(24);
			int64_t sum = 0;
			double blocks = 0;
			const double t0 = now_s();
			double t1 = t0;
			do
			{
				if (par[4] == 4)
				{
					for (size_t i = 0; i < n; ++i)
					{
						const uint8_t *refs[4];
						int s[4];
						for (int k = 0; k < 4; ++k) refs[k] = in[1].p + ro[4 * i + k];
						f4(in[0].p + so[i], par[0], refs, par[1], s, (uint32_t)par[2]);
						sum += s[0] + s[1] + s[2] + s[3];
					}
				}
				else
				{
					for (size_t i = 0; i < n; ++i)
						sum += f1(in[0].p + so[i], par[0], in[1].p + ro[i], par[1], (uint32_t)par[2]);
				}
				blocks += (double)n;
				t1 = now_s();
			} while (t1 - t0 < min_s);
			r[0] = t1 - t0; r[1] = blocks; memcpy(&r[2], &sum, 8);
			out[0].p = (uint8_t *)r; out[0].bytes = 24; nout = 1;
			break;
		}
		case OP_BENCH_PRED: /* CPU baseline leg beside inter prediction (config #3): par: stride_dst, stride_ref, taps, w, h, n, dst_bytes, bi(0|1), min_ms;
		                       bufs as OP_PRED_UNI (bi = 0) or OP_PRED_BI (bi = 1).  The reference's own table entries (pred_inter.c:154-228 uni,
		                       :490-530 bi), whole passes over the PU list until min_ms elapsed.  -> double[2] = { seconds, PUs } and i64 checksum */
		{
			const size_t n = (size_t)par[5];
			const int bi = (int)par[7], taps = (int)par[2], w = (int)par[3], h = (int)par[4];
			const uint32_t *dof = (const uint32_t *)in[bi ? 2 : 1].p, *r0 = (const uint32_t *)in[bi ? 3 : 2].p, *r1 = bi ? (const uint32_t *)in[4].p : 0;
			const uint8_t *f0 = in[bi ? 5 : 3].p, *f1 = bi ? in[6].p : 0;
			uint8_t *dst = (uint8_t *)calloc((size_t)par[6] + 64, 1);
			const double min_s = (double)par[8] * 1e-3;
			double *r = (double *)malloc(24);
			double blocks = 0;
			const double t0 = now_s();
			double t1 = t0;
			do
			{
				for (size_t i = 0; i < n; ++i)
				{
					const int xa = f0[i] & 15, ya = f0[i] >> 4;
					if (bi)
					{
						const int xb = f1[i] & 15, yb = f1[i] >> 4;
						(*hevcasm_get_pred_bi_8to8(&t_bi, taps, w, h, xa, ya, xb, yb))(dst + dof[i], par[0], in[0].p + r0[i], in[1].p + r1[i], par[1], w, h, xa, ya, xb, yb);
					}
					else
						(*hevcasm_get_pred_uni_8to8(&t_uni, taps, w, h, xa, ya))(dst + dof[i], par[0], in[0].p + r0[i], par[1], w, h, xa, ya);
				}
				blocks += (double)n;
				t1 = now_s();
			} while (t1 - t0 < min_s);
			int64_t sum = 0;
			for (size_t i = 0; i < (size_t)par[6]; i += 97) sum += dst[i];
			free(dst);
			r[0] = t1 - t0; r[1] = blocks; memcpy(&r[2], &sum, 8);
			out[0].p = (uint8_t *)r; out[0].bytes = 24; nout = 1;
			break;
		}
		case OP_BENCH_INV: /* CPU baseline leg beside the inverse transforms (config #4): par: stride_dst, stride_pred, trType, log2, n, dst_bytes,
		                      fused(0|1), scale, shift, min_ms; bufs as OP_INV.  The reference's table entries (residual_decode.c:371-413, with
		                      quantize.c:53-62 in front when fused).  -> double[2] = { seconds, blocks } and i64 checksum */
		{
			const size_t n = (size_t)par[4];
			const int log2n = (int)par[3];
			const size_t nn = (size_t)1 << (2 * log2n);
			const uint32_t *dof = (const uint32_t *)in[1].p, *pof = (const uint32_t *)in[2].p;
			const int16_t *c = (const int16_t *)in[3].p;
			hevcasm_inverse_transform_add *f = *hevcasm_get_inverse_transform_add(&t_inv, (int)par[2], log2n);
			hevcasm_quantize_inverse *dq = *hevcasm_get_quantize_inverse(&t_deq);
			uint8_t *dst = (uint8_t *)calloc((size_t)par[5] + 64, 1);
			const double min_s = (double)par[9] * 1e-3;
			double *r = (double *)malloc(24);
			double blocks = 0;
			const double t0 = now_s();
			double t1 = t0;
			do
			{
				for (size_t i = 0; i < n; ++i)
				{
					HEVCASM_ALIGN(32, int16_t, tmp[32 * 32]);
					const int16_t *ci = c + nn * i;
					if (par[6])
					{
						dq(tmp, ci, (int)par[7], (int)par[8], (int)nn);
						ci = tmp;
					}
					f(dst + dof[i], par[0], in[0].p + pof[i], par[1], ci);
				}
				blocks += (double)n;
				t1 = now_s();
			} while (t1 - t0 < min_s);
			int64_t sum = 0;
			for (size_t i = 0; i < (size_t)par[5]; i += 97) sum += dst[i];
			free(dst);
			r[0] = t1 - t0; r[1] = blocks; memcpy(&r[2], &sum, 8);
			out[0].p = (uint8_t *)r; out[0].bytes = 24; nout = 1;
			break;
		}
		default:
			status = 1;
		}

		{
			const uint32_t rmagic = 0x53525648u;
			wr(&rmagic, 4); wr(&status, 4); wr(&nout, 4);
			for (uint32_t i = 0; i < nout; ++i) { wr(&out[i].bytes, 8); wr(out[i].p, out[i].bytes); free(out[i].p); }
			fflush(stdout);
		}
		for (uint32_t i = 0; i < nbuf; ++i) free(in[i].p);
	}
}
