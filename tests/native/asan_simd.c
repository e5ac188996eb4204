/*
 * asan_simd.c -- the host SIMD kernels (csrc/host/{sad,pred,transform,quant}_x86.c) against the C path (cpath.c) under AddressSanitizer and
 * UndefinedBehaviorSanitizer: every operand sits flush against the END (and, for prediction windows, the START) of an exactly sized heap
 * allocation, so a read or write one byte outside the reference's footprint is a sanitizer report.  Built and run by tests/test_sanitizers.py:
 *     gcc -std=gnu11 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -I include -I hevcasm_amd/csrc/host asan_simd.c ...x86.c cpath.c
 * Exit status 0 = every comparison equal and no report.
 */
#include "cpath.h"
#include "sad_x86.h"
#include "pred_x86.h"
#include "transform_x86.h"
#include "quant_x86.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t rs = 0x48455643ull;
static uint32_t rnd(void)
{
	uint64_t z = (rs += 0x9e3779b97f4a7c15ull);
	z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
	z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
	return (uint32_t)((z ^ (z >> 31)) >> 16);
}
static int fails = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++fails; fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)

/* a w x h window of random bytes (kind 1: only 0 / 255) as an EXACT allocation: rows `stride` apart, the last row ends the allocation */
static uint8_t *plane(int w, int h, int stride, int kind, size_t *bytes)
{
	*bytes = (size_t)(h - 1) * stride + w;
	uint8_t *p = malloc(*bytes);
	for (size_t i = 0; i < *bytes; ++i) p[i] = kind ? ((rnd() & 1) ? 255 : 0) : (uint8_t)rnd();
	return p;
}

static void test_pred(void)
{
	static const int sizes[][2] = { { 64, 64 }, { 16, 16 }, { 12, 16 }, { 8, 4 }, { 4, 8 }, { 24, 32 }, { 48, 64 }, { 2, 4 }, { 6, 8 }, { 32, 8 }, { 1, 1 }, { 7, 3 }, { 17, 5 } };
	for (int taps = 8; taps >= 4; taps -= 4)
		for (unsigned s = 0; s < sizeof(sizes) / sizeof(sizes[0]); ++s)
		{
			const int w = sizes[s][0], h = sizes[s][1];
			if (taps == 4 && (w > 32 || h > 32)) continue;
			const int phases = taps == 8 ? 4 : 8, before = taps / 2 - 1;
			for (int xf = 0; xf < phases; ++xf)
				for (int yf = 0; yf < phases; ++yf)
					for (int kind = 0; kind < 2; ++kind)
					{
						/* the reference's footprint for this fraction pair: margins only where it filters */
						const int mx0 = xf ? before : 0, mx1 = xf ? taps / 2 : 0, my0 = yf ? before : 0, my1 = yf ? taps / 2 : 0;
						const int ww = w + mx0 + mx1, wh = h + my0 + my1, stride = ww + (int)(rnd() % 5);
						size_t nb;
						uint8_t *win = plane(ww, wh, stride, kind, &nb);
						const uint8_t *ref = win + my0 * stride + mx0;
						const int sd = w;                       /* dst exactly w x h */
						uint8_t *a = malloc((size_t)w * h), *b = malloc((size_t)w * h);
						(taps == 8 ? hevcasm_c_pred_uni_8tap : hevcasm_c_pred_uni_4tap)(a, sd, ref, stride, w, h, xf, yf);
						if (xf | yf) (taps == 8 ? hevcasm_sse4_pred_uni_8tap : hevcasm_sse4_pred_uni_4tap)(b, sd, ref, stride, w, h, xf, yf);
						else hevcasm_sse2_pred_uni_copy(b, sd, ref, stride, w, h, 0, 0);
						CHECK(!memcmp(a, b, (size_t)w * h), "pred_uni taps %d %dx%d frac %d,%d kind %d", taps, w, h, xf, yf, kind);
						/* bi: the second reference with its own fraction pair and its own exact window */
						const int xg = (int)(rnd() % phases), yg = (int)(rnd() % phases);
						const int nx0 = xg ? before : 0, nx1 = xg ? taps / 2 : 0, ny0 = yg ? before : 0, ny1 = yg ? taps / 2 : 0;
						/* both references share one stride: build the second window with the same stride, exactly sized */
						size_t nb2 = (size_t)(h + ny0 + ny1 - 1) * stride + (w + nx0 + nx1);
						uint8_t *win2 = malloc(nb2);
						for (size_t i = 0; i < nb2; ++i) win2[i] = kind ? ((rnd() & 1) ? 255 : 0) : (uint8_t)rnd();
						const uint8_t *ref2 = win2 + ny0 * stride + nx0;
						if (stride >= w + nx0 + nx1)
						{
							(taps == 8 ? hevcasm_c_pred_bi_8tap : hevcasm_c_pred_bi_4tap)(a, sd, ref, ref2, stride, w, h, xf, yf, xg, yg);
							if (xf | yf | xg | yg) (taps == 8 ? hevcasm_sse4_pred_bi_8tap : hevcasm_sse4_pred_bi_4tap)(b, sd, ref, ref2, stride, w, h, xf, yf, xg, yg);
							else hevcasm_sse2_pred_bi_copy(b, sd, ref, ref2, stride, w, h, 0, 0, 0, 0);
							CHECK(!memcmp(a, b, (size_t)w * h), "pred_bi taps %d %dx%d frac %d,%d %d,%d kind %d", taps, w, h, xf, yf, xg, yg, kind);
						}
						free(win); free(win2); free(a); free(b);
					}
		}
}

static void test_transforms(void)
{
	for (int log2n = 2; log2n <= 5; ++log2n)
		for (int dst4 = 0; dst4 <= (log2n == 2); ++dst4)
			for (int kind = 0; kind < 4; ++kind)
			{
				const int n = 1 << log2n, stride = n + (int)(rnd() % 7);
				int16_t *c = malloc(sizeof(int16_t) * n * n);
				for (int i = 0; i < n * n; ++i)
					c[i] = kind == 0 ? (int16_t)rnd() : kind == 1 ? (int16_t)((rnd() % 9) - 4) : kind == 2 ? ((i & 1) ? -32768 : 32767) : (int16_t)(i == 0 ? 32767 : 0);
				size_t nb;
				uint8_t *pred = plane(n, n, stride, 0, &nb);
				uint8_t *a = malloc(nb), *b = malloc(nb), *b2 = malloc(nb);
				memset(a, 0, nb); memset(b, 0, nb); memset(b2, 0, nb);
				hevcasm_inverse_transform_add *cf = dst4 ? hevcasm_c_idst_4 : log2n == 2 ? hevcasm_c_idct_4 : log2n == 3 ? hevcasm_c_idct_8 : log2n == 4 ? hevcasm_c_idct_16 : hevcasm_c_idct_32;
				hevcasm_inverse_transform_add *vf = dst4 ? hevcasm_avx2_idst_4 : log2n == 2 ? hevcasm_avx2_idct_4 : log2n == 3 ? hevcasm_avx2_idct_8 : log2n == 4 ? hevcasm_avx2_idct_16 : hevcasm_avx2_idct_32;
				cf(a, stride, pred, stride, c);
				vf(b, stride, pred, stride, c);
				CHECK(!memcmp(a, b, nb), "inverse avx2 log2 %d dst %d kind %d", log2n, dst4, kind);
				if (!dst4 && (log2n == 3 || log2n == 4))
				{
					(log2n == 3 ? hevcasm_ssse3_idct_8 : hevcasm_ssse3_idct_16)(b2, stride, pred, stride, c);
					CHECK(!memcmp(a, b2, nb), "inverse ssse3 log2 %d kind %d", log2n, kind);
				}
				if (log2n == 4)
				{
					/* forward 16x16: the source as an exact strided allocation of int16 */
					const int ss = 16 + (int)(rnd() % 5);
					int16_t *src = malloc(sizeof(int16_t) * ((size_t)15 * ss + 16)), *o1 = malloc(512), *o2 = malloc(512);
					for (int i = 0; i < 15 * ss + 16; ++i) src[i] = kind == 0 ? (int16_t)rnd() : (int16_t)((rnd() & 0x1ff) - 0x100);
					hevcasm_c_dct_16(o1, src, ss);
					hevcasm_ssse3_dct_16(o2, src, ss);
					CHECK(!memcmp(o1, o2, 512), "forward ssse3 16 kind %d", kind);
					free(src); free(o1); free(o2);
				}
				free(c); free(pred); free(a); free(b); free(b2);
			}
}

static void test_quant_and_costs(void)
{
	for (int n = 16; n <= 1024; n *= 4)
		for (int kind = 0; kind < 3; ++kind)
		{
			int16_t *s = malloc(2 * (size_t)n), *a = malloc(2 * (size_t)n), *b = malloc(2 * (size_t)n);
			for (int i = 0; i < n; ++i) s[i] = kind == 0 ? (int16_t)rnd() : kind == 1 ? (int16_t)-32768 : (int16_t)((rnd() % 7) - 3);
			const int scale = 1 + (int)(rnd() % 0x7fff), shift = 1 + (int)(rnd() % 14);
			hevcasm_c_quantize_inverse(a, s, scale, shift, n);
			hevcasm_sse4_quantize_inverse(b, s, scale, shift, n);
			CHECK(!memcmp(a, b, 2 * (size_t)n), "dequant n %d kind %d", n, kind);
			const int qshift = 16 + (int)(rnd() % 12), off = (int)(rnd() % 0x8000);
			const int ca = hevcasm_c_quantize(a, s, scale, qshift, off, n), cb = hevcasm_sse4_quantize(b, s, scale, qshift, off, n);
			CHECK(!memcmp(a, b, 2 * (size_t)n) && ca == cb, "quant n %d kind %d cbf %d %d", n, kind, ca, cb);
			/* zero coefficients under a rounding offset of 1.0 and more (outside the reference's asserted domain): the C path keeps
			 * +((offset << (shift - 16)) >> shift) for x == 0, and so must the SIMD entry (a sign restored with _mm_sign_epi32 gave 0) */
			for (int i = 0; i < n; i += 3) s[i] = 0;
			const int ca2 = hevcasm_c_quantize(a, s, 16384, 20, 70000, n), cb2 = hevcasm_sse4_quantize(b, s, 16384, 20, 70000, n);
			CHECK(!memcmp(a, b, 2 * (size_t)n) && ca2 == cb2 && a[0] == 1, "quant zero inputs, offset 70000: n %d kind %d first %d %d", n, kind, a[0], b[0]);
			free(s); free(a); free(b);
		}
	for (int size = 4; size <= 64; size *= 2)
	{
		const int stride = size + (int)(rnd() % 9);
		size_t nb;
		uint8_t *x = plane(size, size, stride, 0, &nb), *y = plane(size, size, stride, 1, &nb);
		if (size >= 16) CHECK(hevcasm_c_ssd(x, stride, y, stride, size, size) == hevcasm_avx_ssd(x, stride, y, stride, size, size), "ssd %d", size);
		if (size <= 32)
		{
			int16_t *res = malloc(2 * (size_t)size * size);
			for (int i = 0; i < size * size; ++i) res[i] = (int16_t)rnd();
			uint8_t *a = malloc(nb), *b = malloc(nb);
			memset(a, 0, nb); memset(b, 0, nb);
			hevcasm_c_quantize_reconstruct(a, stride, x, stride, res, size);
			hevcasm_sse4_quantize_reconstruct(b, stride, x, stride, res, size);
			CHECK(!memcmp(a, b, nb), "reconstruct %d", size);
			free(res); free(a); free(b);
		}
		/* SAD: the SSE2 / AVX2 kernels on exact planes, a misaligned reference */
		if (size >= 8)
		{
			const uint32_t rect = HEVCASM_RECT(size, size);
			CHECK(hevcasm_c_sad(x, stride, y, stride, rect) == hevcasm_sse2_sad(x, stride, y, stride, rect), "sad sse2 %d", size);
			const uint8_t *refs[4] = { y, y, y, y };
			int s1[4], s2[4];
			hevcasm_c_sad_multiref_4(x, stride, refs, stride, s1, rect);
			hevcasm_avx2_sad_multiref_4(x, stride, refs, stride, s2, rect);
			CHECK(!memcmp(s1, s2, sizeof(s1)), "sad4 avx2 %d", size);
		}
		free(x); free(y);
	}
}

/* the host C intra entries (planar, DC, every angular mode) on records of EXACTLY the bytes the layout promises: the 4N + 1 samples at the end of a
 * heap block that stops at byte 128, preceded by nothing below byte 64 - 2N (the record pointer is moved so that byte 64 - 2N is the block's first):
 * a read outside [64 - 2N, 64 + 2N) or [128] is an ASan error; for sizes below 32 the bytes between 64 + 2N and 128 are poison values (they lie inside
 * the block) and must not change the result. */
static void test_intra(void)
{
	for (int log2n = 2; log2n <= 5; ++log2n)
		for (int edge = 0; edge <= (log2n < 5); ++edge)
		{
			const int n = 1 << log2n, packed = (8 << 8) | (log2n << 1) | edge;
			const size_t first = (size_t)(64 - 2 * n), bytes = 129 - first;
			uint8_t *block = malloc(bytes), *nb = block - first;          /* nb[64 - 2N] is block[0]; nothing below it exists */
			uint8_t *a = malloc((size_t)n * n), *b = malloc((size_t)n * n);
			for (int mode = 0; mode < 35; ++mode)
			{
				for (size_t i = 0; i < bytes; ++i) block[i] = (uint8_t)rnd();
				hevcasm_pred_intra *f = mode == 0 ? hevcasm_c_pred_intra_planar : mode == 1 ? hevcasm_c_pred_intra_dc : hevcasm_c_pred_intra_angular;
				f(a, nb, mode, packed);
				for (int i = 64 + 2 * n; i < 128; ++i) nb[i] ^= 0x5a;     /* bytes of the record that no sample lives in */
				f(b, nb, mode, packed);
				CHECK(!memcmp(a, b, (size_t)n * n), "intra %dx%d mode %d reads bytes outside its samples", n, n, mode);
			}
			free(block); free(a); free(b);
		}
}

int main(void)
{
	test_intra();
	test_pred();
	test_transforms();
	test_quant_and_costs();
	printf("asan_simd: %d failures\n", fails);
	return fails != 0;
}
