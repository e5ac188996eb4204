/*
 * cpath.c -- portable C implementations behind HEVCASM_C_REF / HEVCASM_C_OPT (see cpath.h).
 *
 * Written from the HEVC specification's definitions in direct form (2-D separable filters as
 * two explicit passes; transforms as plain matrix products with the spec's intermediate
 * clipping), i.e. deliberately a different formulation from both the reference's butterflies
 * and the HIP kernels, so that the three agree only if all are right.  Semantics that are the
 * reference's own choices rather than the spec's (what is clipped, what is truncated, shifts)
 * are cited to /root/reference/src/lib.
 */
#include "cpath.h"

#include <stdlib.h>
#include <string.h>

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

/* ------------------------------------------------------------------------------ distortion */

int hevcasm_c_sad(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref, uint32_t rect)
{
	const int w = (int)(rect >> 8), h = (int)(rect & 0xff);
	int total = 0;
	for (int y = 0; y < h; ++y, src += stride_src, ref += stride_ref)
		for (int x = 0; x < w; ++x)
		{
			const int d = src[x] - ref[x];
			total += d < 0 ? -d : d;
		}
	return total;
}

void hevcasm_c_sad_multiref_4(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref[], ptrdiff_t stride_ref, int sad[], uint32_t rect)
{
	for (int i = 0; i < 4; ++i) sad[i] = hevcasm_c_sad(src, stride_src, ref[i], stride_ref, rect);
}

int hevcasm_c_ssd(const uint8_t *a, ptrdiff_t stride_a, const uint8_t *b, ptrdiff_t stride_b, int w, int h)
{
	int total = 0;
	for (int y = 0; y < h; ++y, a += stride_a, b += stride_b)
		for (int x = 0; x < w; ++x)
		{
			const int d = a[x] - b[x];
			total += d * d;
		}
	return total;
}

/* SATD by the definition: T = H D H with the (unnormalised, natural-order) Hadamard matrix
 * H[i][j] = (-1)^popcount(i & j).  The sum of magnitudes does not depend on the row order of H,
 * so it equals the reference's butterfly result (hadamard.c:76-111). */
static int satd(int n, const uint8_t *a, ptrdiff_t stride_a, const uint8_t *b, ptrdiff_t stride_b)
{
	int d[8][8], t[8][8];
	for (int y = 0; y < n; ++y)
		for (int x = 0; x < n; ++x) d[y][x] = a[y * stride_a + x] - b[y * stride_b + x];
	for (int y = 0; y < n; ++y)
		for (int u = 0; u < n; ++u)
		{
			int s = 0;
			for (int x = 0; x < n; ++x) s += __builtin_parity((unsigned)(u & x)) ? -d[y][x] : d[y][x];
			t[y][u] = s;
		}
	int total = n / 4;
	for (int v = 0; v < n; ++v)
		for (int u = 0; u < n; ++u)
		{
			int s = 0;
			for (int y = 0; y < n; ++y) s += __builtin_parity((unsigned)(v & y)) ? -t[y][u] : t[y][u];
			total += s < 0 ? -s : s;
		}
	return total / (n / 2);
}

int hevcasm_c_satd_2(const uint8_t *a, ptrdiff_t sa, const uint8_t *b, ptrdiff_t sb) { return satd(2, a, sa, b, sb); }
int hevcasm_c_satd_4(const uint8_t *a, ptrdiff_t sa, const uint8_t *b, ptrdiff_t sb) { return satd(4, a, sa, b, sb); }
int hevcasm_c_satd_8(const uint8_t *a, ptrdiff_t sa, const uint8_t *b, ptrdiff_t sb) { return satd(8, a, sa, b, sb); }

/* -------------------------------------------------------------------------- inter prediction */

int hevcasm_c_filter_tap(int n, int frac, int k)
{
	/* HEVC spec tables 8-11 (luma) and 8-12 (chroma); reference pred_inter.c:53-83 */
	static const signed char luma[4][8] = {
		{ 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 },
		{ -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
	static const signed char chroma[8][4] = {
		{ 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
		{ -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };
	return n == 8 ? luma[frac][k] : chroma[frac][k];
}

/* First (horizontal) pass of one reference at full precision into int16 rows of stride 64: row y of `rows` is window row
 * y - before of the plane (reference pred_inter.c:175, :506).  Memory read is the reference's and no more: the filter margins are
 * touched only in a direction whose fraction is non-zero -- W x H for an integer vector (the reference's memcpy,
 * pred_inter.c:141-151), (W + T - 1) x H horizontal-only, W x (H + T - 1) vertical-only (pred_inter.c:154-228).  Rows the
 * vertical tap {64} never multiplies by anything but zero are left zero without being read. */
static void horizontal_pass(int taps, int16_t *rows, const uint8_t *ref, ptrdiff_t stride_ref, int w, int h, int xFrac, int yFrac)
{
	const int before = taps / 2 - 1;
	for (int y = 0; y < h + taps - 1; ++y)
	{
		if (!yFrac && (y < before || y >= before + h))
		{
			memset(rows + y * 64, 0, sizeof(int16_t) * (size_t)w);
			continue;
		}
		const uint8_t *r = ref + (y - before) * stride_ref;
		for (int x = 0; x < w; ++x)
		{
			int a = 64 * r[x];
			if (xFrac)
			{
				a = 0;
				for (int k = 0; k < taps; ++k) a += hevcasm_c_filter_tap(taps, xFrac, k) * r[x + k - before];
			}
			rows[y * 64 + x] = (int16_t)a;
		}
	}
}

/* 14-bit intermediate prediction of one reference: horizontal pass at full precision, vertical pass >> 6 without
 * rounding (reference pred_inter.c:506-527).  out has row stride 64.  A zero fraction is the tap {64}. */
static void predict_14bit(int taps, int16_t *out, const uint8_t *ref, ptrdiff_t stride_ref, int w, int h, int xFrac, int yFrac)
{
	int16_t rows[(64 + 7) * 64];
	horizontal_pass(taps, rows, ref, stride_ref, w, h, xFrac, yFrac);
	for (int y = 0; y < h; ++y)
		for (int x = 0; x < w; ++x)
		{
			int a = 0;
			for (int k = 0; k < taps; ++k) a += hevcasm_c_filter_tap(taps, yFrac, k) * rows[(y + k) * 64 + x];
			out[y * 64 + x] = (int16_t)(a >> 6);
		}
}

/* Uni-prediction = 14-bit prediction rounded back to 8 bits: (p + 32) >> 6, clipped.  This
 * equals the reference's four special cases (copy, h: (sum+32)>>6, v: same, hv: (sum+2048)>>12;
 * pred_inter.c:141-228) because the un-filtered direction contributes exactly a factor 64. */
static void pred_uni(int taps, uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *ref, ptrdiff_t stride_ref, int w, int h, int xFrac, int yFrac)
{
	/* as predict_14bit, but all 12 fractional bits are kept until the single final rounding */
	int16_t rows[(64 + 7) * 64];
	horizontal_pass(taps, rows, ref, stride_ref, w, h, xFrac, yFrac);
	for (int y = 0; y < h; ++y)
		for (int x = 0; x < w; ++x)
		{
			int a = 0;
			for (int k = 0; k < taps; ++k) a += hevcasm_c_filter_tap(taps, yFrac, k) * rows[(y + k) * 64 + x];
			dst[y * stride_dst + x] = (uint8_t)clampi((a + 2048) >> 12, 0, 255);
		}
}

void hevcasm_c_pred_uni_8tap(uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *ref, ptrdiff_t stride_ref, int nPbW, int nPbH, int xFrac, int yFrac)
{
	pred_uni(8, dst, stride_dst, ref, stride_ref, nPbW, nPbH, xFrac, yFrac);
}

void hevcasm_c_pred_uni_4tap(uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *ref, ptrdiff_t stride_ref, int nPbW, int nPbH, int xFrac, int yFrac)
{
	pred_uni(4, dst, stride_dst, ref, stride_ref, nPbW, nPbH, xFrac, yFrac);
}

static void pred_bi(int taps, uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *ref0, const uint8_t *ref1, ptrdiff_t stride_ref,
		int w, int h, int xFrac0, int yFrac0, int xFrac1, int yFrac1)
{
	int16_t p0[64 * 64], p1[64 * 64];
	predict_14bit(taps, p0, ref0, stride_ref, w, h, xFrac0, yFrac0);
	predict_14bit(taps, p1, ref1, stride_ref, w, h, xFrac1, yFrac1);
	for (int y = 0; y < h; ++y)
		for (int x = 0; x < w; ++x)
			dst[y * stride_dst + x] = (uint8_t)clampi((p0[y * 64 + x] + p1[y * 64 + x] + 64) >> 7, 0, 255); /* pred_inter.c:490-501 */
}

void hevcasm_c_pred_bi_8tap(uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *ref0, const uint8_t *ref1, ptrdiff_t stride_ref, int nPbW, int nPbH, int xFrac0, int yFrac0, int xFrac1, int yFrac1)
{
	pred_bi(8, dst, stride_dst, ref0, ref1, stride_ref, nPbW, nPbH, xFrac0, yFrac0, xFrac1, yFrac1);
}

void hevcasm_c_pred_bi_4tap(uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *ref0, const uint8_t *ref1, ptrdiff_t stride_ref, int nPbW, int nPbH, int xFrac0, int yFrac0, int xFrac1, int yFrac1)
{
	pred_bi(4, dst, stride_dst, ref0, ref1, stride_ref, nPbW, nPbH, xFrac0, yFrac0, xFrac1, yFrac1);
}

/* --------------------------------------------------------------------------------- transforms */

/* 64 * sqrt(2) * cos(m * pi / 64) as HEVC rounds it, m = 0..32 (m = 0 is only used for row 0,
 * where the spec has 64). */
static const signed char cos_table[33] = {
	64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
	61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0 };

static signed char matrix32[32][32];
static const signed char dst4[4][4] = { { 29, 55, 74, 84 }, { 74, 74, 0, -74 }, { 84, -29, -74, 55 }, { 55, -84, 74, -29 } };

__attribute__((constructor)) static void build_matrix(void)
{
	for (int k = 0; k < 32; ++k)
		for (int i = 0; i < 32; ++i)
		{
			int m = (k * (2 * i + 1)) % 128;
			if (m > 64) m = 128 - m;
			matrix32[k][i] = (signed char)(k == 0 ? 64 : (m > 32 ? -cos_table[64 - m] : cos_table[m]));
		}
}

int hevcasm_c_transform_coefficient(int n, int k, int i)
{
	return matrix32[k * (32 / n)][i];
}

/* n x n coefficient matrix as ints, row k = basis function k */
static void load_matrix(int trType, int n, int m[32][32])
{
	for (int k = 0; k < n; ++k)
		for (int i = 0; i < n; ++i)
			m[k][i] = trType ? dst4[k][i] : matrix32[k * (32 / n)][i];
}

/* Inverse: first stage over columns (shift 7), second over rows (shift 12), each clipped to
 * int16 (reference residual_decode.c:113-116 and the drivers :371-413); then add to the
 * prediction, narrowing the sum to int16 before the 8-bit clip as the reference does (:350-368).
 * Loops are ordered so that the innermost one runs over contiguous accumulators. */
static void inverse_add(int trType, int n, uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *pred, ptrdiff_t stride_pred, const int16_t *c)
{
	int m[32][32];
	int16_t e[32 * 32];
	int acc[32];
	load_matrix(trType, n, m);

	/* e[y][x] = clip16((sum_k m[k][y] * c[k][x] + 64) >> 7) */
	for (int y = 0; y < n; ++y)
	{
		for (int x = 0; x < n; ++x) acc[x] = 64;
		for (int k = 0; k < n; ++k)
		{
			const int mk = m[k][y];
			for (int x = 0; x < n; ++x) acc[x] += mk * c[k * n + x];
		}
		for (int x = 0; x < n; ++x) e[y * n + x] = (int16_t)clampi(acc[x] >> 7, -32768, 32767);
	}
	/* r[y][x] = clip16((sum_k m[k][x] * e[y][k] + 2048) >> 12) */
	for (int y = 0; y < n; ++y)
	{
		for (int x = 0; x < n; ++x) acc[x] = 2048;
		for (int k = 0; k < n; ++k)
		{
			const int ek = e[y * n + k];
			for (int x = 0; x < n; ++x) acc[x] += m[k][x] * ek;
		}
		for (int x = 0; x < n; ++x)
		{
			const int r = clampi(acc[x] >> 12, -32768, 32767);
			const int16_t s = (int16_t)(pred[y * stride_pred + x] + r);
			dst[y * stride_dst + x] = (uint8_t)clampi(s, 0, 255);
		}
	}
}

void hevcasm_c_idst_4(uint8_t *d, ptrdiff_t sd, const uint8_t *p, ptrdiff_t sp, const int16_t *c) { inverse_add(1, 4, d, sd, p, sp, c); }
void hevcasm_c_idct_4(uint8_t *d, ptrdiff_t sd, const uint8_t *p, ptrdiff_t sp, const int16_t *c) { inverse_add(0, 4, d, sd, p, sp, c); }
void hevcasm_c_idct_8(uint8_t *d, ptrdiff_t sd, const uint8_t *p, ptrdiff_t sp, const int16_t *c) { inverse_add(0, 8, d, sd, p, sp, c); }
void hevcasm_c_idct_16(uint8_t *d, ptrdiff_t sd, const uint8_t *p, ptrdiff_t sp, const int16_t *c) { inverse_add(0, 16, d, sd, p, sp, c); }
void hevcasm_c_idct_32(uint8_t *d, ptrdiff_t sd, const uint8_t *p, ptrdiff_t sp, const int16_t *c) { inverse_add(0, 32, d, sd, p, sp, c); }

/* Forward: first stage along rows (shift log2(n) - 1), second along columns (shift log2(n) + 6);
 * results are stored to int16 by truncation, never clipped (reference residual_decode.c:631-634,
 * drivers :855-892). */
static void forward(int trType, int n, int log2n, int16_t *coeffs, const int16_t *src, ptrdiff_t src_stride)
{
	int m[32][32], mt[32][32];
	int16_t t[32 * 32];
	int acc[32];
	const int s1 = log2n - 1, s2 = log2n + 6;
	load_matrix(trType, n, m);
	for (int k = 0; k < n; ++k)
		for (int i = 0; i < n; ++i) mt[i][k] = m[k][i];

	/* t[y][k] = (sum_x m[k][x] * src[y][x] + round) >> s1 */
	for (int y = 0; y < n; ++y)
	{
		for (int k = 0; k < n; ++k) acc[k] = 1 << (s1 - 1);
		for (int x = 0; x < n; ++x)
		{
			const int sx = src[y * src_stride + x];
			for (int k = 0; k < n; ++k) acc[k] += mt[x][k] * sx;
		}
		for (int k = 0; k < n; ++k) t[y * n + k] = (int16_t)(acc[k] >> s1);
	}
	/* coeffs[k][x] = (sum_y m[k][y] * t[y][x] + round) >> s2 */
	for (int k = 0; k < n; ++k)
	{
		for (int x = 0; x < n; ++x) acc[x] = 1 << (s2 - 1);
		for (int y = 0; y < n; ++y)
		{
			const int mk = m[k][y];
			for (int x = 0; x < n; ++x) acc[x] += mk * t[y * n + x];
		}
		for (int x = 0; x < n; ++x) coeffs[k * n + x] = (int16_t)(acc[x] >> s2);
	}
}

void hevcasm_c_dst_4(int16_t *c, const int16_t *s, ptrdiff_t st) { forward(1, 4, 2, c, s, st); }
void hevcasm_c_dct_4(int16_t *c, const int16_t *s, ptrdiff_t st) { forward(0, 4, 2, c, s, st); }
void hevcasm_c_dct_8(int16_t *c, const int16_t *s, ptrdiff_t st) { forward(0, 8, 3, c, s, st); }
void hevcasm_c_dct_16(int16_t *c, const int16_t *s, ptrdiff_t st) { forward(0, 16, 4, c, s, st); }
void hevcasm_c_dct_32(int16_t *c, const int16_t *s, ptrdiff_t st) { forward(0, 32, 5, c, s, st); }

/* ------------------------------------------------------------------------------ quantisation */

void hevcasm_c_quantize_inverse(int16_t *dst, const int16_t *src, int scale, int shift, int n)
{
	const int round = 1 << (shift - 1);
	for (int i = 0; i < n; ++i) dst[i] = (int16_t)clampi((src[i] * scale + round) >> shift, -32768, 32767);
}

int hevcasm_c_quantize(int16_t *dst, const int16_t *src, int scale, int shift, int offset, int n)
{
	const int round = offset << (shift - 16);
	int any = 0;
	for (int i = 0; i < n; ++i)
	{
		const int v = src[i];
		const int mag = ((v < 0 ? -v : v) * scale + round) >> shift;
		const int q = clampi(v < 0 ? -mag : mag, -32768, 32767);
		any |= q;
		dst[i] = (int16_t)q;
	}
	return any;
}

void hevcasm_c_quantize_reconstruct(uint8_t *rec, ptrdiff_t stride_rec, const uint8_t *pred, ptrdiff_t stride_pred, const int16_t *res, int n)
{
	for (int y = 0; y < n; ++y)
		for (int x = 0; x < n; ++x)
			rec[y * stride_rec + x] = (uint8_t)clampi(pred[y * stride_pred + x] + res[y * n + x], 0, 255);
}

/* ---- intra DC (pred_intra.c:61-101), written from the HEVC text 8.4.4.2.5 ------------------ */

void hevcasm_c_pred_intra_dc(uint8_t *dst, const uint8_t *neighbours, int intraPredMode, hevcasm_pred_intra_packed packed)
{
	(void)intraPredMode;
	const int log2n = (packed >> 1) & 0x7f, n = 1 << log2n;
	const uint8_t *top = neighbours + 64;       /* top[x]  = sample (x, -1) */
	const uint8_t *left = neighbours + 63;      /* left[-y] = sample (-1, y) */
	int sum = n;
	for (int i = 0; i < n; ++i) sum += top[i] + left[-i];
	const int dc = sum >> (log2n + 1);
	for (int y = 0; y < n; ++y)
		for (int x = 0; x < n; ++x) dst[y * n + x] = (uint8_t)dc;
	if (packed & 1)
	{
		dst[0] = (uint8_t)((left[0] + 2 * dc + top[0] + 2) >> 2);
		for (int i = 1; i < n; ++i)
		{
			dst[i] = (uint8_t)((top[i] + 3 * dc + 2) >> 2);
			dst[i * n] = (uint8_t)((left[-i] + 3 * dc + 2) >> 2);
		}
	}
}

/* ---- intra planar and angular (H.265 8.4.4.2.4, 8.4.4.2.6): NOT in the reference's C (it has them as f265 assembly for 4x4 / 8x8 only, which this
 * toolchain cannot assemble), so unpinned: written from the text, on the reference's neighbour layout (left column up to neighbours[63], top row from
 * neighbours[64], corner at neighbours[128]: f265/intra.asm:126-131).  Every sample is looked up where the text's ref[] array would hold it. ---------- */

static const signed char intra_angle_of_mode[35] = { 0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
static const short intra_inverse_angle[15] = { -4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096 };

void hevcasm_c_pred_intra_planar(uint8_t *dst, const uint8_t *neighbours, int intraPredMode, hevcasm_pred_intra_packed packed)
{
	(void)intraPredMode;
	const int log2n = (packed >> 1) & 0x7f, n = 1 << log2n;
	const uint8_t *top = neighbours + 64, *left = neighbours + 63;
	const int top_right = top[n], bottom_left = left[-n];
	for (int y = 0; y < n; ++y)
		for (int x = 0; x < n; ++x)
			dst[y * n + x] = (uint8_t)(((n - 1 - x) * left[-y] + (x + 1) * top_right + (n - 1 - y) * top[x] + (y + 1) * bottom_left + n) >> (log2n + 1));
}

void hevcasm_c_pred_intra_angular(uint8_t *dst, const uint8_t *neighbours, int intraPredMode, hevcasm_pred_intra_packed packed)
{
	const int log2n = (packed >> 1) & 0x7f, n = 1 << log2n;
	const int angle = intra_angle_of_mode[intraPredMode], vertical = intraPredMode >= 18;
	const int inverse = angle < 0 ? intra_inverse_angle[intraPredMode - 11] : 0;
	/* along[i], i >= 1: sample i - 1 of the edge the mode runs along (top for vertical modes); across[k]: sample k - 1 of the other edge; [0] = corner */
	const uint8_t *top = neighbours + 64, *left = neighbours + 63;
#define ALONG(i) ((i) == 0 ? neighbours[128] : vertical ? top[(i) - 1] : left[-((i) - 1)])
#define ACROSS(k) ((k) == 0 ? neighbours[128] : vertical ? left[-((k) - 1)] : top[(k) - 1])
#define REF(i) ((i) >= 0 ? ALONG(i) : ACROSS(((i) * inverse + 128) >> 8))
	for (int v = 0; v < n; ++v)          /* distance from the edge the mode runs along */
	{
		const int shift = ((v + 1) * angle) >> 5, weight = ((v + 1) * angle) & 31;
		for (int u = 0; u < n; ++u)
		{
			const int a = REF(u + shift + 1);
			const int value = weight ? ((32 - weight) * a + weight * REF(u + shift + 2) + 16) >> 5 : a;
			dst[vertical ? v * n + u : u * n + v] = (uint8_t)value;
		}
	}
	if ((packed & 1) && angle == 0)      /* modes 10 / 26 of luma blocks below 32x32: half the gradient along the other edge onto the first row / column */
		for (int v = 0; v < n; ++v)
			dst[vertical ? v * n : v] = (uint8_t)clampi(ALONG(1) + ((ACROSS(v + 1) - neighbours[128]) >> 1), 0, 255);
#undef REF
#undef ACROSS
#undef ALONG
}

/* ---- linear SSD (diff.c:45-54) ------------------------------------------------------------ */

int hevcasm_c_ssd_linear(const uint8_t *src0, const uint8_t *src1, int size)
{
	unsigned total = 0;   /* wraps like the reference's int on two's-complement targets, without the UB */
	for (int i = 0; i < size; ++i)
	{
		const int d = src0[i] - src1[i];
		total += (unsigned)(d * d);
	}
	return (int)total;
}
