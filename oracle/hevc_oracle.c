/*
 * hevc_oracle.c -- CPU restatement of the reference's C path (HEVCASM_C_REF).
 *
 * TEST INFRASTRUCTURE ONLY (see hevc_oracle.h).  Plain C99, one thread per
 * call; the batch drivers may use OpenMP when built with -fopenmp.
 *
 * Parity: pinned against oracle/_ref/refdrv (the reference's own C files) and
 * tests/golden/.  Citations are file:line into /root/reference/src/lib.
 */
#include "hevc_oracle.h"

#include <stdlib.h>
#include <string.h>

static int clip3(int lo, int hi, int v)
{
	return v < lo ? lo : (v > hi ? hi : v);
}

/* ------------------------------------------------------------------ SAD / SSD */

/* sad.c:47-60: rect = (width << 8) | height (hevcasm.h:156) */
int orc_sad(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref, uint32_t rect)
{
	const int w = (int)(rect >> 8), h = (int)(rect & 0xff);
	int acc = 0;
	for (int y = 0; y < h; ++y)
		for (int x = 0; x < w; ++x)
			acc += abs((int)src[x + y * stride_src] - (int)ref[x + y * stride_ref]);
	return acc;
}

/* sad.c:101-121 */
void orc_sad_multiref_4(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref[], ptrdiff_t stride_ref, int sad[], uint32_t rect)
{
	for (int way = 0; way < 4; ++way)
		sad[way] = orc_sad(src, stride_src, ref[way], stride_ref, rect);
}

/* ssd.c:43-55 */
int orc_ssd(const uint8_t *pA, ptrdiff_t strideA, const uint8_t *pB, ptrdiff_t strideB, int w, int h)
{
	int acc = 0;
	for (int y = 0; y < h; ++y)
		for (int x = 0; x < w; ++x)
		{
			const int d = (int)pA[x + y * strideA] - (int)pB[x + y * strideB];
			acc += d * d;
		}
	return acc;
}

/* ---------------------------------------------------------------- Hadamard SATD */

/* hadamard.c:45-58: one butterfly stage of span n over m values read at `stride` */
static void hadamard_stage(int m, int n, int *dst, const int *src, ptrdiff_t stride)
{
	for (int i = 0; i < m; i += 2 * n)
		for (int j = 0; j < n; ++j)
		{
			const int a = src[(i + j) * stride], b = src[(i + n + j) * stride];
			dst[i + j] = a + b;
			dst[i + n + j] = a - b;
		}
}

/* hadamard.c:60-73: stages of span n, n/2, ..., 1 */
static void hadamard_1d(int m, int n, int *dst, const int *src, ptrdiff_t stride)
{
	if (n == 1) { hadamard_stage(m, n, dst, src, stride); return; }
	int tmp[8];
	hadamard_stage(m, n, tmp, src, stride);
	hadamard_1d(m, n / 2, dst, tmp, 1);
}

/* hadamard.c:76-111: rows then columns, sum of magnitudes, (sum + n/4) / (n/2) */
int orc_hadamard_satd(int n, const uint8_t *pA, ptrdiff_t strideA, const uint8_t *pB, ptrdiff_t strideB)
{
	int inter[8][8];
	for (int y = 0; y < n; ++y)
	{
		int diff[8];
		for (int x = 0; x < n; ++x) diff[x] = pA[x + y * strideA] - pB[x + y * strideB];
		hadamard_1d(n, n / 2, inter[y], diff, 1);
	}
	int sum = n / 4;
	for (int x = 0; x < n; ++x)
	{
		int col[8];
		hadamard_1d(n, n / 2, col, &inter[0][x], 8);
		for (int y = 0; y < n; ++y) sum += abs(col[y]);
	}
	return sum / (n / 2);
}

/* ------------------------------------------------------------ inter prediction */

/* pred_inter.c:53-83: HEVC luma (8-tap, quarter-pel) and chroma (4-tap, eighth-pel) filters */
int orc_pred_coefficient(int n, int frac, int k)
{
	static const int8_t luma[4][8] = {
		{  0, 0,   0, 64,  0,   0, 0,  0 },
		{ -1, 4, -10, 58, 17,  -5, 1,  0 },
		{ -1, 4, -11, 40, 40, -11, 4, -1 },
		{  0, 1,  -5, 17, 58, -10, 4, -1 } };
	static const int8_t chroma[8][4] = {
		{  0, 64,  0,  0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
		{ -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };
	return n == 8 ? luma[frac][k] : chroma[frac][k];
}

/* pred_inter.c:90-138: one 1-D FIR pass.  src/dst element size 1 or 2 bytes; stride_tap = 1
 * (horizontal) or the row stride (vertical).  16-bit stores truncate (plain assignment, :124),
 * 8-bit stores clip to [0,255] (:128). */
static void fir_pass(void *dst, int dst_bytes, ptrdiff_t stride_dst,
		const void *src, int src_bytes, ptrdiff_t stride_src,
		int w, int h, ptrdiff_t stride_tap, int n, int frac, int shift, int add)
{
	int16_t *d16 = (int16_t *)dst;
	uint8_t *d8 = (uint8_t *)dst;
	const int16_t *s16 = (const int16_t *)src;
	const uint8_t *s8 = (const uint8_t *)src;

	for (int y = 0; y < h; ++y)
	{
		for (int x = 0; x < w; ++x)
		{
			int a = add << shift >> 1; /* :106 */
			for (int k = 0; k < n; ++k)
			{
				const ptrdiff_t o = x + (k - n / 2 + 1) * stride_tap; /* :110 */
				const int s = src_bytes == 2 ? s16[o] : s8[o];
				a += orc_pred_coefficient(n, frac, k) * s;
			}
			a >>= shift;
			if (dst_bytes == 2) d16[x] = (int16_t)a;
			else d8[x] = (uint8_t)clip3(0, 255, a);
		}
		d16 += stride_dst; d8 += stride_dst;
		s16 += stride_src; s8 += stride_src;
	}
}

/* Selector C branch pred_inter.c:234-283 picks copy / h / v / hv by (xFrac != 0, yFrac != 0);
 * bodies: copy :141-151, h :154-159 / :203-208, v :162-167 / :211-216, hv :170-179 / :219-228. */
void orc_pred_uni_8to8(int taps, uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *ref, ptrdiff_t stride_ref,
		int w, int h, int xFrac, int yFrac)
{
	if (!xFrac && !yFrac)
	{
		for (int y = 0; y < h; ++y) memcpy(dst + y * stride_dst, ref + y * stride_ref, (size_t)w);
	}
	else if (xFrac && !yFrac)
	{
		fir_pass(dst, 1, stride_dst, ref, 1, stride_ref, w, h, 1, taps, xFrac, 6, 1);
	}
	else if (!xFrac && yFrac)
	{
		fir_pass(dst, 1, stride_dst, ref, 1, stride_ref, w, h, stride_ref, taps, yFrac, 6, 1);
	}
	else
	{
		/* intermediate: (h + taps - 1) rows; row stride 64 (luma :172) or 32 (chroma :221) */
		const int is = taps == 8 ? 64 : 32;
		const int lead = taps / 2 - 1;
		int16_t tmp[(64 + 7) * 64];
		fir_pass(tmp, 2, is, ref - lead * stride_ref, 1, stride_ref, w, h + taps - 1, 1, taps, xFrac, 0, 0);
		fir_pass(dst, 1, stride_dst, tmp + lead * is, 2, is, w, h, is, taps, yFrac, 12, 1);
	}
}

/* pred_inter.c:504-530 (pipeline) + :490-501 (mean).  Runs the full H (shift 0) -> V (shift 6,
 * no rounding) pipeline for each reference whatever the fractions, then (a + b + 64) >> 7, clip. */
void orc_pred_bi_8to8(int taps, uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *ref0, const uint8_t *ref1, ptrdiff_t stride_ref,
		int w, int h, int xFrac0, int yFrac0, int xFrac1, int yFrac1)
{
	static _Thread_local int16_t t[4][(64 + 7) * 64];
	const int lead = taps / 2 - 1;

	fir_pass(t[2], 2, 64, ref0 - lead * stride_ref, 1, stride_ref, w, h + taps - 1, 1, taps, xFrac0, 0, 0);
	fir_pass(t[0], 2, 64, t[2] + lead * 64, 2, 64, w, h, 64, taps, yFrac0, 6, 0);
	fir_pass(t[3], 2, 64, ref1 - lead * stride_ref, 1, stride_ref, w, h + taps - 1, 1, taps, xFrac1, 0, 0);
	fir_pass(t[1], 2, 64, t[3] + lead * 64, 2, 64, w, h, 64, taps, yFrac1, 6, 0);

	for (int y = 0; y < h; ++y)
		for (int x = 0; x < w; ++x)
		{
			const int v = ((int)t[0][x + y * 64] + (int)t[1][x + y * 64] + (1 << 6)) >> 7;
			dst[x + y * stride_dst] = (uint8_t)clip3(0, 255, v);
		}
}

/* ------------------------------------------------------------------ transforms */

/* The HEVC core transform matrix.  The reference spells it out as literals
 * (32x32 at residual_decode.c:254-288 and :793-827; 16/8/4 at :182-200, :132-142, :99-105, which
 * are the even rows of the next size up, truncated).  Here it is generated from the 31 distinct
 * magnitudes: entry (k, i) of the 32-point matrix is +-mag[|angle|] with angle = k(2i+1) folded
 * into a quarter period.  Parity tests against refdrv prove the table. */
static const int8_t dct_mag[33] = {
	64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
	61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0 };

static int dct32(int k, int i)
{
	if (k == 0) return 64;
	int a = (k * (2 * i + 1)) & 127;  /* angle in units of pi/64, period 128 */
	if (a > 64) a = 128 - a;          /* cos is even about pi */
	return a > 32 ? -dct_mag[64 - a] : dct_mag[a];
}

int orc_dct_matrix(int n, int k, int i)
{
	return dct32(k * (32 / n), i);
}

/* Inverse N-point transform of one column by even/odd decomposition, the structure of
 * residual_decode.c:91-121 (4), :124-171 (8), :174-243 (16), :246-347 (32):
 * out[k] = E[k] + O[k], out[N-1-k] = E[k] - O[k], E = inverse (N/2)-point of the even rows.
 * 32-bit accumulation, no intermediate rounding. */
static void inv_even_odd(int n, const int *in /* n coefficients */, int *out)
{
	if (n == 2)
	{
		/* rows 0 and N/2 of any size: 64, 64 / 64, -64 (e.g. :227-229 EEEE) */
		out[0] = 64 * in[0] + 64 * in[1];
		out[1] = 64 * in[0] - 64 * in[1];
		return;
	}
	int even[16] = { 0 }, E[16] = { 0 }, O[16] = { 0 };
	const int half = n / 2;
	for (int k = 0; k < half; ++k) even[k] = in[2 * k];
	inv_even_odd(half, even, E);
	for (int k = 0; k < half; ++k)
	{
		int o = 0;
		for (int r = 1; r < n; r += 2) o += orc_dct_matrix(n, r, k) * in[r]; /* e.g. :290-297 */
		O[k] = o;
	}
	for (int k = 0; k < half; ++k)
	{
		out[k] = E[k] + O[k];
		out[n - 1 - k] = E[k] - O[k];
	}
}

/* One inverse pass over all columns: result clipped to int16 and stored transposed
 * (dst row j = transform of src column j), e.g. :113-116, :341-342. */
static void inv_pass_dct(int n, int16_t *dst, const int16_t *src, int shift)
{
	const int add = 1 << (shift - 1);
	for (int j = 0; j < n; ++j)
	{
		int in[32], out[32];
		for (int k = 0; k < n; ++k) in[k] = src[k * n + j];
		inv_even_odd(n, in, out);
		for (int k = 0; k < n; ++k) dst[j * n + k] = (int16_t)clip3(-32768, 32767, (out[k] + add) >> shift);
	}
}

/* residual_decode.c:69-88: 4x4 inverse DST-VII pass */
static void inv_pass_dst(int16_t *dst, const int16_t *src, int shift)
{
	const int add = 1 << (shift - 1);
	for (int i = 0; i < 4; ++i)
	{
		const int s0 = src[i], s1 = src[4 + i], s2 = src[8 + i], s3 = src[12 + i];
		const int c0 = s0 + s2, c1 = s2 + s3, c2 = s0 - s3, c3 = 74 * s1;
		dst[4 * i + 0] = (int16_t)clip3(-32768, 32767, (29 * c0 + 55 * c1 + c3 + add) >> shift);
		dst[4 * i + 1] = (int16_t)clip3(-32768, 32767, (55 * c2 - 29 * c1 + c3 + add) >> shift);
		dst[4 * i + 2] = (int16_t)clip3(-32768, 32767, (74 * (s0 - s2 + s3) + add) >> shift);
		dst[4 * i + 3] = (int16_t)clip3(-32768, 32767, (55 * c0 + 29 * c2 - c3 + add) >> shift);
	}
}

/* residual_decode.c:350-356: the argument is narrowed to int16_t before the comparison */
static uint8_t clip_pixel(int16_t x)
{
	if (x > 255) return 255;
	if (x < 0) return 0;
	return (uint8_t)x;
}

/* residual_decode.c:371-413 (drivers, shifts 7 then 12) + :359-368 (add residual) */
void orc_inverse_transform_add(int trType, int log2TrafoSize, uint8_t *dst, ptrdiff_t stride_dst,
		const uint8_t *pred, ptrdiff_t stride_pred, const int16_t *coeffs)
{
	const int n = 1 << log2TrafoSize;
	int16_t t0[32 * 32], t1[32 * 32];
	if (trType)
	{
		inv_pass_dst(t0, coeffs, 7);
		inv_pass_dst(t1, t0, 12);
	}
	else
	{
		inv_pass_dct(n, t0, coeffs, 7);
		inv_pass_dct(n, t1, t0, 12);
	}
	for (int y = 0; y < n; ++y)
		for (int x = 0; x < n; ++x)
			dst[x + y * stride_dst] = clip_pixel((int16_t)(pred[x + y * stride_pred] + t1[x + y * n]));
}

/* Forward N-point transform of one row by even/odd decomposition: structure of
 * residual_decode.c:613-639 (4), :642-687 (8), :690-756 (16), :759-852 (32).
 * E[k] = x[k] + x[N-1-k] feeds the even output rows, O[k] = x[k] - x[N-1-k] the odd ones. */
static void fwd_even_odd(int n, int stride_rows /* row step in the n-point matrix */, const int *x, int *out /* out[k*... ] */)
{
	/* out[r] for r = 0..n-1 in natural order of this sub-transform */
	if (n == 2)
	{
		out[0] = 64 * x[0] + 64 * x[1];
		out[1] = 64 * x[0] - 64 * x[1];
		(void)stride_rows;
		return;
	}
	int E[16] = { 0 }, O[16] = { 0 }, sub[16] = { 0 };
	const int half = n / 2;
	for (int k = 0; k < half; ++k)
	{
		E[k] = x[k] + x[n - 1 - k];
		O[k] = x[k] - x[n - 1 - k];
	}
	fwd_even_odd(half, stride_rows * 2, E, sub);
	for (int k = 0; k < half; ++k) out[2 * k] = sub[k];
	for (int r = 1; r < n; r += 2)
	{
		int a = 0;
		for (int k = 0; k < half; ++k) a += orc_dct_matrix(n, r, k) * O[k]; /* e.g. :837-846 */
		out[r] = a;
	}
}

/* One forward pass: rows of src in, columns of dst out; plain (truncating) int16 store,
 * no clipping (e.g. :631-634). */
static void fwd_pass_dct(int n, int16_t *dst, const int16_t *src, ptrdiff_t src_stride, int shift)
{
	const int add = 1 << (shift - 1);
	for (int j = 0; j < n; ++j)
	{
		int x[32], out[32];
		for (int k = 0; k < n; ++k) x[k] = src[j * src_stride + k];
		fwd_even_odd(n, 1, x, out);
		for (int k = 0; k < n; ++k) dst[k * n + j] = (int16_t)((out[k] + add) >> shift);
	}
}

/* residual_decode.c:592-610: forward 4x4 DST-VII pass */
static void fwd_pass_dst(int16_t *dst, const int16_t *src, ptrdiff_t src_stride, int shift)
{
	const int add = 1 << (shift - 1);
	for (int i = 0; i < 4; ++i)
	{
		const int s0 = src[src_stride * i + 0], s1 = src[src_stride * i + 1];
		const int s2 = src[src_stride * i + 2], s3 = src[src_stride * i + 3];
		const int c0 = s0 + s3, c1 = s1 + s3, c2 = s0 - s1, c3 = 74 * s2;
		dst[0 * 4 + i] = (int16_t)((29 * c0 + 55 * c1 + c3 + add) >> shift);
		dst[1 * 4 + i] = (int16_t)((74 * (s0 + s1 - s3) + add) >> shift);
		dst[2 * 4 + i] = (int16_t)((29 * c2 + 55 * c0 - c3 + add) >> shift);
		dst[3 * 4 + i] = (int16_t)((55 * c2 - 29 * c1 + c3 + add) >> shift);
	}
}

/* residual_decode.c:855-892: shift pairs (1,8) (2,9) (3,10) (4,11) = (log2-1, log2+6) */
void orc_transform(int trType, int log2TrafoSize, int16_t *coeffs, const int16_t *src, ptrdiff_t src_stride)
{
	const int n = 1 << log2TrafoSize;
	int16_t tmp[32 * 32];
	if (trType)
	{
		fwd_pass_dst(tmp, src, src_stride, 1);
		fwd_pass_dst(coeffs, tmp, 4, 8);
	}
	else
	{
		fwd_pass_dct(n, tmp, src, src_stride, log2TrafoSize - 1);
		fwd_pass_dct(n, coeffs, tmp, n, log2TrafoSize + 6);
	}
}

/* ---------------------------------------------------------------- quantisation */

/* quantize.c:53-62 */
void orc_quantize_inverse(int16_t *dst, const int16_t *src, int scale, int shift, int n)
{
	for (int i = 0; i < n; ++i)
		dst[i] = (int16_t)clip3(-32768, 32767, ((src[i] * scale) + (1 << (shift - 1))) >> shift);
}

/* quantize.c:160-186.  abs(-32768) = 32768 in int (:176); the return value is the OR of all
 * quantised levels (:182), of which only zero / non-zero is meaningful (:257). */
int orc_quantize(int16_t *dst, const int16_t *src, int scale, int shift, int offset, int n)
{
	offset <<= (shift - 16);
	int cbf = 0;
	for (int i = 0; i < n; ++i)
	{
		int x = src[i];
		const int sign = x < 0 ? -1 : 1;
		x = abs(x);
		x = ((x * scale) + offset) >> shift;
		x *= sign;
		x = clip3(-32768, 32767, x);
		cbf |= x;
		dst[i] = (int16_t)x;
	}
	return cbf;
}

/* quantize.c:292-302 */
void orc_quantize_reconstruct(uint8_t *rec, ptrdiff_t stride_rec, const uint8_t *pred, ptrdiff_t stride_pred, const int16_t *res, int n)
{
	for (int y = 0; y < n; ++y)
		for (int x = 0; x < n; ++x)
			rec[x + y * stride_rec] = (uint8_t)clip3(0, 255, pred[x + y * stride_pred] + res[x + y * n]);
}

/* ---------------------------------------------------------------- batch drivers */

void orc_sad_batch(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref,
		const uint32_t *src_off, const uint32_t *ref_off, int *sad, uint32_t rect, size_t n)
{
#pragma omp parallel for schedule(static)
	for (ptrdiff_t i = 0; i < (ptrdiff_t)n; ++i)
		sad[i] = orc_sad(src + src_off[i], stride_src, ref + ref_off[i], stride_ref, rect);
}

void orc_sad_multiref_batch(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref,
		const uint32_t *src_off, const uint32_t *ref_off4, int *sad4, uint32_t rect, size_t n)
{
#pragma omp parallel for schedule(static)
	for (ptrdiff_t i = 0; i < (ptrdiff_t)n; ++i)
	{
		const uint8_t *r[4];
		for (int way = 0; way < 4; ++way) r[way] = ref + ref_off4[4 * i + way];
		orc_sad_multiref_4(src + src_off[i], stride_src, r, stride_ref, sad4 + 4 * i, rect);
	}
}

void orc_ssd_batch(const uint8_t *a, ptrdiff_t stride_a, const uint8_t *b, ptrdiff_t stride_b,
		const uint32_t *a_off, const uint32_t *b_off, int *ssd, int w, int h, size_t n)
{
#pragma omp parallel for schedule(static)
	for (ptrdiff_t i = 0; i < (ptrdiff_t)n; ++i)
		ssd[i] = orc_ssd(a + a_off[i], stride_a, b + b_off[i], stride_b, w, h);
}

/* frac byte = xFrac | (yFrac << 4) */
void orc_pred_uni_batch(int taps, uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *ref, ptrdiff_t stride_ref,
		const uint32_t *dst_off, const uint32_t *ref_off, const uint8_t *frac, int w, int h, size_t n)
{
#pragma omp parallel for schedule(static)
	for (ptrdiff_t i = 0; i < (ptrdiff_t)n; ++i)
		orc_pred_uni_8to8(taps, dst + dst_off[i], stride_dst, ref + ref_off[i], stride_ref, w, h, frac[i] & 15, frac[i] >> 4);
}

void orc_pred_bi_batch(int taps, uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *ref0, const uint8_t *ref1, ptrdiff_t stride_ref,
		const uint32_t *dst_off, const uint32_t *ref0_off, const uint32_t *ref1_off,
		const uint8_t *frac0, const uint8_t *frac1, int w, int h, size_t n)
{
#pragma omp parallel for schedule(static)
	for (ptrdiff_t i = 0; i < (ptrdiff_t)n; ++i)
		orc_pred_bi_8to8(taps, dst + dst_off[i], stride_dst, ref0 + ref0_off[i], ref1 + ref1_off[i], stride_ref, w, h,
				frac0[i] & 15, frac0[i] >> 4, frac1[i] & 15, frac1[i] >> 4);
}

void orc_inverse_transform_add_batch(int trType, int log2TrafoSize, uint8_t *dst, ptrdiff_t stride_dst,
		const uint8_t *pred, ptrdiff_t stride_pred, const uint32_t *dst_off, const uint32_t *pred_off,
		const int16_t *coeffs, size_t n)
{
	const size_t nn = (size_t)1 << (2 * log2TrafoSize);
#pragma omp parallel for schedule(static)
	for (ptrdiff_t i = 0; i < (ptrdiff_t)n; ++i)
		orc_inverse_transform_add(trType, log2TrafoSize, dst + dst_off[i], stride_dst, pred + pred_off[i], stride_pred, coeffs + nn * i);
}

void orc_dequant_inverse_transform_add_batch(int trType, int log2TrafoSize, uint8_t *dst, ptrdiff_t stride_dst,
		const uint8_t *pred, ptrdiff_t stride_pred, const uint32_t *dst_off, const uint32_t *pred_off,
		const int16_t *levels, int scale, int shift, size_t n)
{
	const size_t nn = (size_t)1 << (2 * log2TrafoSize);
#pragma omp parallel for schedule(static)
	for (ptrdiff_t i = 0; i < (ptrdiff_t)n; ++i)
	{
		int16_t c[32 * 32];
		orc_quantize_inverse(c, levels + nn * i, scale, shift, (int)nn);
		orc_inverse_transform_add(trType, log2TrafoSize, dst + dst_off[i], stride_dst, pred + pred_off[i], stride_pred, c);
	}
}

/* src_off is in int16 elements */
void orc_transform_batch(int trType, int log2TrafoSize, int16_t *coeffs, const int16_t *src, ptrdiff_t src_stride,
		const uint32_t *src_off, size_t n)
{
	const size_t nn = (size_t)1 << (2 * log2TrafoSize);
#pragma omp parallel for schedule(static)
	for (ptrdiff_t i = 0; i < (ptrdiff_t)n; ++i)
		orc_transform(trType, log2TrafoSize, coeffs + nn * i, src + src_off[i], src_stride);
}

void orc_quantize_batch(int16_t *dst, const int16_t *src, int scale, int shift, int offset, int n_per_block, int *cbf, size_t n_blocks)
{
#pragma omp parallel for schedule(static)
	for (ptrdiff_t i = 0; i < (ptrdiff_t)n_blocks; ++i)
		cbf[i] = orc_quantize(dst + (size_t)n_per_block * i, src + (size_t)n_per_block * i, scale, shift, offset, n_per_block);
}

void orc_quantize_reconstruct_batch(uint8_t *rec, ptrdiff_t stride_rec, const uint8_t *pred, ptrdiff_t stride_pred,
		const uint32_t *rec_off, const uint32_t *pred_off, const int16_t *res, int nCbS, size_t n)
{
#pragma omp parallel for schedule(static)
	for (ptrdiff_t i = 0; i < (ptrdiff_t)n; ++i)
		orc_quantize_reconstruct(rec + rec_off[i], stride_rec, pred + pred_off[i], stride_pred, res + (size_t)nCbS * nCbS * i, nCbS);
}

void orc_hadamard_satd_batch(int n, const uint8_t *a, ptrdiff_t stride_a, const uint8_t *b, ptrdiff_t stride_b,
		const uint32_t *a_off, const uint32_t *b_off, int *satd, size_t count)
{
#pragma omp parallel for schedule(static)
	for (ptrdiff_t i = 0; i < (ptrdiff_t)count; ++i)
		satd[i] = orc_hadamard_satd(n, a + a_off[i], stride_a, b + b_off[i], stride_b);
}

void orc_me_full_search(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref,
		const uint32_t *src_off, const uint32_t *ref_off, int *out3, uint32_t rect, int range, size_t n)
{
#pragma omp parallel for schedule(dynamic, 1)
	for (ptrdiff_t i = 0; i < (ptrdiff_t)n; ++i)
	{
		int best = 0x7fffffff, bx = 0, by = 0;
		for (int dy = -range; dy <= range; ++dy)
			for (int dx = -range; dx <= range; ++dx)
			{
				const int c = orc_sad(src + src_off[i], stride_src, ref + (ptrdiff_t)ref_off[i] + dy * stride_ref + dx, stride_ref, rect);
				if (c < best) { best = c; bx = dx; by = dy; }
			}
		out3[3 * i + 0] = bx; out3[3 * i + 1] = by; out3[3 * i + 2] = best;
	}
}

/* cost of a candidate block against the source: metric 0 = SAD (sad.c:47-60); metric 1 = Hadamard SATD
 * (hadamard.c:76-111) summed over 8x8 tiles when both dimensions are multiples of 8, else over 4x4 tiles */
static int block_cost(int metric, const uint8_t *src, ptrdiff_t stride_src, const uint8_t *cand, ptrdiff_t stride_cand, int w, int h)
{
	if (metric == 0) return orc_sad(src, stride_src, cand, stride_cand, ((uint32_t)w << 8) | (uint32_t)h);
	const int n = (w % 8 == 0 && h % 8 == 0) ? 8 : 4;
	int total = 0;
	for (int y = 0; y < h; y += n)
		for (int x = 0; x < w; x += n)
			total += orc_hadamard_satd(n, src + y * stride_src + x, stride_src, cand + y * stride_cand + x, stride_cand);
	return total;
}

void orc_subpel_refine(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref,
		const uint32_t *src_off, const uint32_t *ref_off, int *out3, uint32_t rect, int metric, size_t n)
{
	const int w = (int)(rect >> 8), h = (int)(rect & 0xff);
#pragma omp parallel for schedule(dynamic, 1)
	for (ptrdiff_t i = 0; i < (ptrdiff_t)n; ++i)
	{
		uint8_t cand[64 * 64];
		int best = 0x7fffffff, bx = 0, by = 0;
		for (int qy = -3; qy <= 3; ++qy)
			for (int qx = -3; qx <= 3; ++qx)
			{
				const uint8_t *origin = ref + (ptrdiff_t)ref_off[i] + (qy >> 2) * stride_ref + (qx >> 2);
				orc_pred_uni_8to8(8, cand, 64, origin, stride_ref, w, h, qx & 3, qy & 3);
				const int c = block_cost(metric, src + src_off[i], stride_src, cand, 64, w, h);
				if (c < best) { best = c; bx = qx; by = qy; }
			}
		out3[3 * i + 0] = bx; out3[3 * i + 1] = by; out3[3 * i + 2] = best;
	}
}

/* pred_intra.c:52-58: sample (dx, dy) with dx == -1 (left column) or dy == -1 (top row) */
static int intra_sample(const uint8_t *neighbours, int dx, int dy)
{
	return neighbours[dy < 0 ? 64 + dx : 63 - dy];
}

/* pred_intra.c:61-101 */
void orc_pred_intra_dc(uint8_t *dst, const uint8_t *neighbours, int packed)
{
	const int k = (packed >> 1) & 0x7f;
	const int nTbS = 1 << k;
	int dcVal = nTbS;
	for (int x = 0; x < nTbS; ++x) dcVal += intra_sample(neighbours, x, -1);
	for (int y = 0; y < nTbS; ++y) dcVal += intra_sample(neighbours, -1, y);
	dcVal >>= k + 1;

	int start = 0;
	if (packed & 1)
	{
		start = 1;
		dst[0] = (uint8_t)((intra_sample(neighbours, -1, 0) + 2 * dcVal + intra_sample(neighbours, 0, -1) + 2) >> 2);
		for (int x = 1; x < nTbS; ++x) dst[x] = (uint8_t)((intra_sample(neighbours, x, -1) + 3 * dcVal + 2) >> 2);
		for (int y = 1; y < nTbS; ++y) dst[y * nTbS] = (uint8_t)((intra_sample(neighbours, -1, y) + 3 * dcVal + 2) >> 2);
	}
	for (int y = start; y < nTbS; ++y)
		for (int x = start; x < nTbS; ++x) dst[x + y * nTbS] = (uint8_t)dcVal;
}

/* ---- intra prediction beyond DC: PARITY UNPINNED -----------------------------------------------------------------------
 * The reference wires INTRA_DC only (pred_intra.c:113-139); planar and the 33 angular modes exist there as f265 assembly for
 * 4x4 and 8x8 (f265/intra.asm:216-1394), which cannot be assembled in this image, and the reference holds no vectors for them.
 * What follows restates ITU-T H.265 (04/2013) 8.4.4.2.3 (neighbour filtering), 8.4.4.2.4 (planar) and 8.4.4.2.6 (angular) as
 * the text has them, on the reference's neighbour layout (pred_intra.c:52-58 and f265/intra.asm:126-131: p[-1][y] =
 * neighbours[63 - y], p[x][-1] = neighbours[64 + x], p[-1][-1] = neighbours[128]) and with the reference's packed argument
 * (bit 0 = "edge" = cIdx == 0 && nTbS < 32, which is also the text's condition for the boundary smoothing of modes 10 and 26
 * with disableIntraBoundaryFilter == 0).  Read as TEXT, the assembly agrees with this restatement wherever its arithmetic is visible: the angle tables
 * (intra_angle 2 5 9 13 17 21 26 at intra.asm:78-84, inv_angle 256 315 390 482 630 910 1638 at :68-74); the angular blend is weights (32 - f, f)
 * on neighbouring pairs, then vpmulhrsw by 1024 = (x + 16) >> 5 (DO_PRED, :1153-1157); planar rounds with vpmulhrsw by 2048 at 8x8 = (x + 8) >> 4 and
 * by 4096 at 4x4 = (x + 4) >> 3 (:250, :1069), i.e. (sum + N) >> (log2 N + 1), with the weight pairs (7,1) (6,2) ... (0,8) = (N - 1 - x, x + 1)
 * (planar_wgt_hor, :51-52); the boundary smoothing of the pure horizontal mode is left + ((top - top_left) >> 1), arithmetic shift (:1301-1311);
 * the neighbour filter rounds with vpmulhrsw by 8192 = (x + 2) >> 2 (:863).  That is evidence by reading, not a pin: only mode 1 is pinned by the
 * reference's C. */
static int intra_p(const uint8_t *nb, int x, int y)      /* p[x][y] with x == -1 or y == -1 */
{
	if (x < 0 && y < 0) return nb[128];
	return y < 0 ? nb[64 + x] : nb[63 - y];
}

static const int8_t kIntraPredAngle[35] = { 0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
static const int16_t kInvAngle[15] = { -4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096 };   /* modes 11..25 */

/* 8.4.4.2.4 */
void orc_pred_intra_planar(uint8_t *dst, const uint8_t *nb, int packed)
{
	const int k = (packed >> 1) & 0x7f, n = 1 << k;
	for (int y = 0; y < n; ++y)
		for (int x = 0; x < n; ++x)
			dst[x + y * n] = (uint8_t)(((n - 1 - x) * intra_p(nb, -1, y) + (x + 1) * intra_p(nb, n, -1) + (n - 1 - y) * intra_p(nb, x, -1) + (y + 1) * intra_p(nb, -1, n) + n) >> (k + 1));
}

/* 8.4.4.2.6, predModeIntra 2..34 */
void orc_pred_intra_angular(uint8_t *dst, const uint8_t *nb, int mode, int packed)
{
	const int k = (packed >> 1) & 0x7f, n = 1 << k, edge = packed & 1;
	const int angle = kIntraPredAngle[mode];
	int refbuf[32 + 64 + 1], *ref = refbuf + 32;      /* ref[-nTbS .. 2 nTbS] */
	for (int i = 0; i < 32 + 64 + 1; ++i) refbuf[i] = 0;
	if (mode >= 18)
	{
		for (int x = 0; x <= n; ++x) ref[x] = intra_p(nb, -1 + x, -1);
		if (angle < 0)
		{
			const int inv = kInvAngle[mode - 11];
			if (((n * angle) >> 5) < -1)
				for (int x = (n * angle) >> 5; x <= -1; ++x) ref[x] = intra_p(nb, -1, -1 + ((x * inv + 128) >> 8));
		}
		else
			for (int x = n + 1; x <= 2 * n; ++x) ref[x] = intra_p(nb, -1 + x, -1);
		for (int y = 0; y < n; ++y)
		{
			const int iIdx = ((y + 1) * angle) >> 5, iFact = ((y + 1) * angle) & 31;
			for (int x = 0; x < n; ++x)
				dst[x + y * n] = (uint8_t)(iFact ? ((32 - iFact) * ref[x + iIdx + 1] + iFact * ref[x + iIdx + 2] + 16) >> 5 : ref[x + iIdx + 1]);
		}
		if (mode == 26 && edge)
			for (int y = 0; y < n; ++y)
			{
				const int v = intra_p(nb, 0, -1) + ((intra_p(nb, -1, y) - intra_p(nb, -1, -1)) >> 1);
				dst[y * n] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
			}
	}
	else
	{
		for (int x = 0; x <= n; ++x) ref[x] = intra_p(nb, -1, -1 + x);
		if (angle < 0)
		{
			const int inv = kInvAngle[mode - 11];
			if (((n * angle) >> 5) < -1)
				for (int x = (n * angle) >> 5; x <= -1; ++x) ref[x] = intra_p(nb, -1 + ((x * inv + 128) >> 8), -1);
		}
		else
			for (int x = n + 1; x <= 2 * n; ++x) ref[x] = intra_p(nb, -1, -1 + x);
		for (int y = 0; y < n; ++y)
			for (int x = 0; x < n; ++x)
			{
				const int iIdx = ((x + 1) * angle) >> 5, iFact = ((x + 1) * angle) & 31;
				dst[x + y * n] = (uint8_t)(iFact ? ((32 - iFact) * ref[y + iIdx + 1] + iFact * ref[y + iIdx + 2] + 16) >> 5 : ref[y + iIdx + 1]);
			}
		if (mode == 10 && edge)
			for (int x = 0; x < n; ++x)
			{
				const int v = intra_p(nb, -1, 0) + ((intra_p(nb, x, -1) - intra_p(nb, -1, -1)) >> 1);
				dst[x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
			}
	}
}

void orc_pred_intra(uint8_t *dst, const uint8_t *nb, int mode, int packed)
{
	if (mode == 0) orc_pred_intra_planar(dst, nb, packed);
	else if (mode == 1) orc_pred_intra_dc(dst, nb, packed);
	else if (mode >= 2 && mode <= 34) orc_pred_intra_angular(dst, nb, mode, packed);
}

/* one mode for the whole batch (the reference function's argument); modes != NULL: one mode per block */
void orc_pred_intra_batch(uint8_t *dst, const uint8_t *neighbours, ptrdiff_t neighbour_stride, int intraPredMode, int packed, size_t n)
{
	const int size = 1 << ((packed >> 1) & 0x7f);
	for (size_t i = 0; i < n; ++i) orc_pred_intra(dst + i * size * size, neighbours + (ptrdiff_t)i * neighbour_stride, intraPredMode, packed);
}

void orc_pred_intra_modes_batch(uint8_t *dst, const uint8_t *neighbours, ptrdiff_t neighbour_stride, const uint8_t *modes, int packed, size_t n)
{
	const int size = 1 << ((packed >> 1) & 0x7f);
	for (size_t i = 0; i < n; ++i) orc_pred_intra(dst + i * size * size, neighbours + (ptrdiff_t)i * neighbour_stride, modes[i], packed);
}

/* 8.4.4.2.3: the filtered neighbour array pF of one block (same layout as p; bytes outside the 4 nTbS + 1 samples are copied unchanged where both
 * records hold them).  strong != 0: strong_intra_smoothing_enabled_flag; the bi-linear form then applies to 32x32 blocks whose two edges are flat
 * (|p[-1][-1] + p[nTbS*2-1][-1] - 2 p[nTbS-1][-1]| < 1 << (BitDepth - 5), and the same down the left).  WHETHER a block's neighbours are filtered at all
 * (filterFlag: mode and size) is the caller's decision, as in f265 (intra.asm:748-768 takes a filter_flag). */
void orc_intra_filter_neighbours(uint8_t *out, const uint8_t *nb, int log2, int strong)
{
	const int n = 1 << log2;
	int biInt = 0;
	if (strong && n == 32)
	{
		const int a = intra_p(nb, -1, -1) + intra_p(nb, 2 * n - 1, -1) - 2 * intra_p(nb, n - 1, -1);
		const int b = intra_p(nb, -1, -1) + intra_p(nb, -1, 2 * n - 1) - 2 * intra_p(nb, -1, n - 1);
		biInt = (a < 0 ? -a : a) < (1 << (8 - 5)) && (b < 0 ? -b : b) < (1 << (8 - 5));
	}
	out[128] = biInt ? nb[128] : (uint8_t)((intra_p(nb, -1, 0) + 2 * intra_p(nb, -1, -1) + intra_p(nb, 0, -1) + 2) >> 2);
	for (int i = 0; i < 2 * n; ++i)
	{
		int left, top;
		if (i == 2 * n - 1) { left = intra_p(nb, -1, i); top = intra_p(nb, i, -1); }
		else if (biInt)
		{
			left = ((63 - i) * intra_p(nb, -1, -1) + (i + 1) * intra_p(nb, -1, 63) + 32) >> 6;
			top = ((63 - i) * intra_p(nb, -1, -1) + (i + 1) * intra_p(nb, 63, -1) + 32) >> 6;
		}
		else
		{
			left = (intra_p(nb, -1, i + 1) + 2 * intra_p(nb, -1, i) + intra_p(nb, -1, i - 1) + 2) >> 2;
			top = (intra_p(nb, i + 1, -1) + 2 * intra_p(nb, i, -1) + intra_p(nb, i - 1, -1) + 2) >> 2;
		}
		out[63 - i] = (uint8_t)left;
		out[64 + i] = (uint8_t)top;
	}
}

void orc_intra_filter_neighbours_batch(uint8_t *out, ptrdiff_t out_stride, const uint8_t *neighbours, ptrdiff_t neighbour_stride, int log2, int strong, size_t n)
{
	for (size_t i = 0; i < n; ++i) orc_intra_filter_neighbours(out + (ptrdiff_t)i * out_stride, neighbours + (ptrdiff_t)i * neighbour_stride, log2, strong);
}

/* diff.c:45-54 (unsigned accumulation: same bits as the reference's int, no UB past 2^31) */
int orc_ssd_linear(const uint8_t *p1, const uint8_t *p2, int n)
{
	unsigned sum = 0;
	for (int i = 0; i < n; ++i)
	{
		const int diff = p1[i] - p2[i];
		sum += (unsigned)(diff * diff);
	}
	return (int)sum;
}

void orc_ssd_linear_batch(const uint8_t *src0, const uint8_t *src1, const uint32_t *off0, const uint32_t *off1, int *ssd, int size, size_t n)
{
	for (size_t i = 0; i < n; ++i) ssd[i] = orc_ssd_linear(src0 + off0[i], src1 + off1[i], size);
}

/* Whole-frame SSE: the reference has no such function; the pin is the composition of hevcasm_ssd_linear (diff.c:45-54) over the rows
 * of a frame, each row exact in int (width < 33 025), summed in 64 bits. */
void orc_frame_sse_batch(const uint8_t *a, ptrdiff_t stride_a, const uint8_t *b, ptrdiff_t stride_b, const uint32_t *a_off, const uint32_t *b_off,
                         uint64_t *sse, int width, int height, size_t n)
{
	for (size_t i = 0; i < n; ++i)
	{
		uint64_t sum = 0;
		for (int y = 0; y < height; ++y)
			sum += (uint64_t)(unsigned)orc_ssd_linear(a + a_off[i] + (ptrdiff_t)y * stride_a, b + b_off[i] + (ptrdiff_t)y * stride_b, width);
		sse[i] = sum;
	}
}

/* ------------------------------------------------------------------ a whole intra frame */

/* H.265 8.4.4.2.2, sample by sample.  Search order: p[-1][2 nTbS - 1] up to p[-1][-1], then p[0][-1] to p[2 nTbS - 1][-1]. */
void orc_intra_build_neighbours(uint8_t *nb, const uint8_t *recon, ptrdiff_t stride, int frame_w, int frame_h, const int32_t *order, int cur, int x, int y, int n)
{
	const int cells_w = (frame_w + 3) / 4, total = 4 * n + 1;
	int val[129], ok[129];
	for (int s = 0; s < total; ++s)
	{
		/* s < 2n: left column bottom-up; s == 2n: corner; s > 2n: top row left to right */
		const int px = s <= 2 * n ? x - 1 : x + (s - 2 * n - 1);
		const int py = s < 2 * n ? y + 2 * n - 1 - s : y - 1;
		ok[s] = 0;
		if (px >= 0 && py >= 0 && px < frame_w && py < frame_h)
		{
			const int32_t o = order[(py >> 2) * cells_w + (px >> 2)];
			ok[s] = o >= 0 && o < cur;
		}
		val[s] = ok[s] ? recon[px + (ptrdiff_t)py * stride] : 0;
	}
	int first = -1;
	for (int s = 0; s < total && first < 0; ++s) if (ok[s]) first = s;
	if (first < 0)
		for (int s = 0; s < total; ++s) val[s] = 1 << (8 - 1);        /* nothing available: 1 << (BitDepth - 1) */
	else
	{
		for (int s = 0; s < first; ++s) val[s] = val[first];           /* p[-1][2 nTbS - 1] takes the first available sample in search order ... */
		for (int s = first + 1; s < total; ++s) if (!ok[s]) val[s] = val[s - 1];      /* ... every other unavailable sample the one before it */
	}
	for (int s = 0; s < 2 * n; ++s) nb[63 - (2 * n - 1 - s)] = (uint8_t)val[s];
	nb[128] = (uint8_t)val[2 * n];
	for (int s = 2 * n + 1; s < total; ++s) nb[64 + (s - 2 * n - 1)] = (uint8_t)val[s];
}

int orc_intra_filter_flag(int cIdx, int mode, int n)
{
	if (cIdx != 0 || mode == 1 || n == 4) return 0;
	const int dv = abs(mode - 26), dh = abs(mode - 10), dist = dv < dh ? dv : dh;
	const int thres = n == 8 ? 7 : n == 16 ? 1 : 0;                       /* intraHorVerDistThres[nTbS] */
	return dist > thres;
}

size_t orc_intra_level_offset(int x, int y, int frame_w, int ctu)
{
	const int cols = (frame_w + ctu - 1) / ctu;
	const size_t base = ((size_t)(y / ctu) * cols + (size_t)(x / ctu)) * ctu * ctu;
	const unsigned cx = (unsigned)(x % ctu) >> 2, cy = (unsigned)(y % ctu) >> 2;
	unsigned z = 0;
	for (int b = 0; b < 4; ++b) z |= ((cx >> b) & 1u) << (2 * b) | ((cy >> b) & 1u) << (2 * b + 1);
	return base + 16u * z;
}

int orc_intra_reconstruct_frame(uint8_t *recon, ptrdiff_t stride_recon, const uint8_t *src, ptrdiff_t stride_src, int16_t *levels, int *cbf,
		uint8_t *pred_out, const orc_intra_tu *tus, size_t n_tus, const orc_intra_qp *qp, int frame_w, int frame_h, int ctu, int cIdx, int strong)
{
	const int cells_w = (frame_w + 3) / 4, cells_h = (frame_h + 3) / 4;
	int32_t *order = (int32_t *)malloc(sizeof(int32_t) * (size_t)cells_w * cells_h);
	if (!order) return -1;
	for (int i = 0; i < cells_w * cells_h; ++i) order[i] = -1;
	int rc = 0;
	for (size_t i = 0; i < n_tus && !rc; ++i)
	{
		const int n = 1 << tus[i].log2, x = tus[i].x, y = tus[i].y;
		if (tus[i].log2 < 2 || tus[i].log2 > 5 || n > ctu || tus[i].mode > 34 || x % n || y % n || x + n > frame_w || y + n > frame_h) { rc = -1; break; }
		for (int cy = y / 4; cy < (y + n) / 4; ++cy)
			for (int cx = x / 4; cx < (x + n) / 4; ++cx)
			{
				if (order[cy * cells_w + cx] >= 0) rc = -1;
				order[cy * cells_w + cx] = (int32_t)i;
			}
	}
	for (int i = 0; i < cells_w * cells_h && !rc; ++i) if (order[i] < 0) rc = -1;
	for (size_t i = 0; i < n_tus && !rc; ++i)
	{
		const int log2n = tus[i].log2, n = 1 << log2n, x = tus[i].x, y = tus[i].y, mode = tus[i].mode;
		const orc_intra_qp *q = &qp[log2n - 2];
		uint8_t rec[160], flt[160], pred[32 * 32];
		int16_t res[32 * 32], coef[32 * 32], deq[32 * 32];
		int16_t *lev = levels + orc_intra_level_offset(x, y, frame_w, ctu);
		uint8_t *nb = rec + 16;
		memset(rec, 0, sizeof(rec)); memset(flt, 0, sizeof(flt));
		orc_intra_build_neighbours(nb, recon, stride_recon, frame_w, frame_h, order, (int)i, x, y, n);
		if (orc_intra_filter_flag(cIdx, mode, n))
		{
			orc_intra_filter_neighbours(flt + 16, nb, log2n, strong);
			nb = flt + 16;
		}
		/* the packed argument of the reference's function type: bit 0 = edge filters, luma below 32x32 (pred_intra.h:58-63) */
		const int packed = (8 << 8) | (log2n << 1) | ((cIdx == 0 && log2n < 5) ? 1 : 0);
		orc_pred_intra(pred, nb, mode, packed);
		if (pred_out)
			for (int r = 0; r < n; ++r) memcpy(pred_out + x + (ptrdiff_t)(y + r) * stride_recon, pred + r * n, (size_t)n);
		const int trType = cIdx == 0 && log2n == 2;                          /* DST-VII for 4x4 luma blocks of intra CUs */
		if (src)
		{
			for (int r = 0; r < n; ++r)
				for (int c = 0; c < n; ++c) res[r * n + c] = (int16_t)(src[x + c + (ptrdiff_t)(y + r) * stride_src] - pred[r * n + c]);
			orc_transform(trType, log2n, coef, res, n);
			const int c = orc_quantize(lev, coef, q->q_scale, q->q_shift, q->q_offset, n * n);
			if (cbf) cbf[i] = c;
		}
		orc_quantize_inverse(deq, lev, q->dq_scale, q->dq_shift, n * n);
		orc_inverse_transform_add(trType, log2n, recon + x + (ptrdiff_t)y * stride_recon, stride_recon, pred, n, deq);
	}
	free(order);
	return rc;
}
