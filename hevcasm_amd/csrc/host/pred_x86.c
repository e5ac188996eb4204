/*
 * pred_x86.c -- host SIMD inter prediction behind the reference's own x86 instruction-set bits.
 *
 * The reference fills its prediction tables with x86 assembly when the caller's mask has the bits (reference
 * pred_inter.c:296-346 uni, :570-593 bi): SSE2 block copies / pavgb for integer vectors, and SSE4.1 kernels for the 8-tap
 * and 4-tap filters (pred_inter_a.asm:174-576: pmaddubsw on the bytes for the first pass, pmaddwd on int16 rows for the
 * second).  That assembly cannot be built in this toolchain (no yasm / nasm), so these are the same kernels written with
 * compiler intrinsics, selected where the reference selects its assembly (csrc/host/pred_inter.c), verified against the C
 * path by hevcasm_test()'s SSE2 / SSE4.1 columns and tests/test_cpath.py, and timed by bench.py as the host-SIMD baseline
 * beside the GPU's prediction rows.  One body serves every width (8 output pixels per step, a partial last step), where the
 * reference has a kernel per width class.
 *
 * Arithmetic is the reference C path's, bit for bit (pred_inter.c:90-228, :490-530):
 *   uni  copy | (H + 32) >> 6 | (V + 32) >> 6 | H at full precision into int16, then (V + 2048) >> 12; clipped to 8 bits
 *   bi   per reference H at full precision -> V >> 6 WITHOUT rounding, stored to int16 by truncation (the sum can reach
 *        32 783: pred_inter.c:125 assigns an int to an int16_t), then (a + b + 64) >> 7 clipped
 * Every first-pass sum fits int16 (positive taps sum to at most 88, negative to -22: 88 * 255 = 22 440), so pmaddubsw /
 * paddw never saturate.  Memory read is the reference's footprint and no more: margins only in a direction whose fraction
 * is non-zero, and a 16-byte load that would pass the end of a window row is replaced by an exact-length copy.
 */
#include "pred_x86.h"

#if defined(__x86_64__) || defined(__i386__)

#include "cpath.h"

#include <immintrin.h>
#include <string.h>

#define TARGET_SSE2 __attribute__((target("sse2")))
#define TARGET_SSE41 __attribute__((target("sse4.1")))
#define INLINE static inline __attribute__((always_inline))

/* at most 16 bytes from p, of which n (> 0) are readable */
TARGET_SSE41 INLINE __m128i load_bytes(const uint8_t *p, int n)
{
	if (n >= 16) return _mm_loadu_si128((const __m128i *)p);
	uint8_t tmp[16] = { 0 };
	memcpy(tmp, p, (size_t)n);
	return _mm_loadu_si128((const __m128i *)tmp);
}

/* at most 8 bytes from p (n > 0 readable) in the low half */
TARGET_SSE41 INLINE __m128i load_bytes8(const uint8_t *p, int n)
{
	if (n >= 8) return _mm_loadl_epi64((const __m128i *)p);
	uint8_t tmp[16] = { 0 };
	memcpy(tmp, p, (size_t)n);
	return _mm_loadl_epi64((const __m128i *)tmp);
}

TARGET_SSE41 INLINE void store_bytes8(uint8_t *p, __m128i v, int n)     /* the low min(n, 8) bytes of v */
{
	if (n >= 8) _mm_storel_epi64((__m128i *)p, v);
	else
	{
		uint8_t tmp[16];
		_mm_storeu_si128((__m128i *)tmp, v);
		memcpy(p, tmp, (size_t)n);
	}
}

/* tap pair (k, k + 1) of the filter as interleaved signed bytes (for pmaddubsw) and as interleaved int16 (for pmaddwd) */
TARGET_SSE41 INLINE __m128i coef_bytes(int taps, int frac, int k)
{
	const int c0 = hevcasm_c_filter_tap(taps, frac, k), c1 = hevcasm_c_filter_tap(taps, frac, k + 1);
	return _mm_set1_epi16((short)(((c1 & 0xff) << 8) | (c0 & 0xff)));
}
TARGET_SSE41 INLINE __m128i coef_words(int taps, int frac, int k)
{
	const int c0 = hevcasm_c_filter_tap(taps, frac, k), c1 = hevcasm_c_filter_tap(taps, frac, k + 1);
	return _mm_set1_epi32((int)(((unsigned)c1 << 16) | ((unsigned)c0 & 0xffffu)));
}

/* 8 horizontal sums at full precision from the 16 bytes that start taps / 2 - 1 pixels left of the first output */
TARGET_SSE41 INLINE __m128i hsum8(const int taps, __m128i v, const __m128i *cb)
{
	const __m128i m0 = _mm_setr_epi8(0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8);
	__m128i s = _mm_maddubs_epi16(_mm_shuffle_epi8(v, m0), cb[0]);
	s = _mm_add_epi16(s, _mm_maddubs_epi16(_mm_shuffle_epi8(v, _mm_add_epi8(m0, _mm_set1_epi8(2))), cb[1]));
	if (taps == 8)
	{
		s = _mm_add_epi16(s, _mm_maddubs_epi16(_mm_shuffle_epi8(v, _mm_add_epi8(m0, _mm_set1_epi8(4))), cb[2]));
		s = _mm_add_epi16(s, _mm_maddubs_epi16(_mm_shuffle_epi8(v, _mm_add_epi8(m0, _mm_set1_epi8(6))), cb[3]));
	}
	return s;
}

/* First pass of one reference into int16 rows of stride 64: out[y][x] = sum_k c[k] ref[y][x + k - before], or ref[y][x] << 6 for
 * an integer xFrac (the tap {64}).  `ref` points at the first row wanted. */
TARGET_SSE41 INLINE void first_pass(const int taps, int16_t *out, const uint8_t *ref, ptrdiff_t sr, int w, int rows, int xFrac)
{
	const int before = taps / 2 - 1;
	if (!xFrac)
	{
		for (int y = 0; y < rows; ++y, ref += sr, out += 64)
			for (int x = 0; x < w; x += 8)
				_mm_storeu_si128((__m128i *)(out + x), _mm_slli_epi16(_mm_cvtepu8_epi16(load_bytes8(ref + x, w - x)), 6));
		return;
	}
	__m128i cb[4];
	for (int k = 0; k < taps / 2; ++k) cb[k] = coef_bytes(taps, xFrac, 2 * k);
	for (int y = 0; y < rows; ++y, ref += sr, out += 64)
		for (int x = 0; x < w; x += 8)
			_mm_storeu_si128((__m128i *)(out + x), hsum8(taps, load_bytes(ref + x - before, w + taps - 1 - x), cb));
}

/* Second pass on int16 rows (stride 64): 8 sums of taps rows starting at t, as two vectors of four 32-bit sums */
TARGET_SSE41 INLINE void vsum8(const int taps, const int16_t *t, const __m128i *cw, __m128i *lo, __m128i *hi)
{
	__m128i a = _mm_setzero_si128(), b = _mm_setzero_si128();
	for (int k = 0; k < taps; k += 2)
	{
		const __m128i r0 = _mm_loadu_si128((const __m128i *)(t + 64 * k)), r1 = _mm_loadu_si128((const __m128i *)(t + 64 * (k + 1)));
		a = _mm_add_epi32(a, _mm_madd_epi16(_mm_unpacklo_epi16(r0, r1), cw[k / 2]));
		b = _mm_add_epi32(b, _mm_madd_epi16(_mm_unpackhi_epi16(r0, r1), cw[k / 2]));
	}
	*lo = a; *hi = b;
}

TARGET_SSE41 INLINE void pred_uni(const int taps, uint8_t *dst, ptrdiff_t sd, const uint8_t *ref, ptrdiff_t sr, int w, int h, int xFrac, int yFrac)
{
	const int before = taps / 2 - 1;
	if (xFrac && !yFrac)          /* (H + 32) >> 6 */
	{
		__m128i cb[4];
		for (int k = 0; k < taps / 2; ++k) cb[k] = coef_bytes(taps, xFrac, 2 * k);
		const __m128i rnd = _mm_set1_epi16(32);
		for (int y = 0; y < h; ++y, ref += sr, dst += sd)
			for (int x = 0; x < w; x += 8)
			{
				const __m128i s = _mm_srai_epi16(_mm_add_epi16(hsum8(taps, load_bytes(ref + x - before, w + taps - 1 - x), cb), rnd), 6);
				store_bytes8(dst + x, _mm_packus_epi16(s, s), w - x);
			}
		return;
	}
	if (!xFrac)                   /* (V + 32) >> 6 straight from the bytes: rows interleaved pairwise for pmaddubsw */
	{
		__m128i cb[4];
		for (int k = 0; k < taps / 2; ++k) cb[k] = coef_bytes(taps, yFrac, 2 * k);
		const __m128i rnd = _mm_set1_epi16(32);
		for (int y = 0; y < h; ++y, dst += sd)
			for (int x = 0; x < w; x += 8)
			{
				const int n = w - x < 8 ? w - x : 8;
				const uint8_t *p = ref + (ptrdiff_t)(y - before) * sr + x;
				__m128i s = rnd;
				for (int k = 0; k < taps; k += 2)
					s = _mm_add_epi16(s, _mm_maddubs_epi16(_mm_unpacklo_epi8(load_bytes8(p + k * sr, n), load_bytes8(p + (k + 1) * sr, n)), cb[k / 2]));
				s = _mm_srai_epi16(s, 6);
				store_bytes8(dst + x, _mm_packus_epi16(s, s), n);
			}
		return;
	}
	/* H at full precision, then (V + 2048) >> 12 (pred_inter.c:171-179) */
	int16_t rows[(64 + 7) * 64];
	first_pass(taps, rows, ref - before * sr, sr, w, h + taps - 1, xFrac);
	__m128i cw[4];
	for (int k = 0; k < taps / 2; ++k) cw[k] = coef_words(taps, yFrac, 2 * k);
	const __m128i rnd = _mm_set1_epi32(2048);
	for (int y = 0; y < h; ++y, dst += sd)
		for (int x = 0; x < w; x += 8)
		{
			__m128i lo, hi;
			vsum8(taps, rows + 64 * y + x, cw, &lo, &hi);
			const __m128i s = _mm_packs_epi32(_mm_srai_epi32(_mm_add_epi32(lo, rnd), 12), _mm_srai_epi32(_mm_add_epi32(hi, rnd), 12));
			store_bytes8(dst + x, _mm_packus_epi16(s, s), w - x);
		}
}

/* 14-bit prediction of one reference into int16 rows of stride 64 (pred_inter.c:506-527): first pass at full precision, second
 * pass >> 6 without rounding, truncated to int16 */
TARGET_SSE41 INLINE void predict_14bit(const int taps, int16_t *out, const uint8_t *ref, ptrdiff_t sr, int w, int h, int xFrac, int yFrac)
{
	if (!yFrac)
	{
		first_pass(taps, out, ref, sr, w, h, xFrac);       /* (64 * t) >> 6 == t */
		return;
	}
	const int before = taps / 2 - 1;
	int16_t rows[(64 + 7) * 64];
	first_pass(taps, rows, ref - before * sr, sr, w, h + taps - 1, xFrac);
	__m128i cw[4];
	for (int k = 0; k < taps / 2; ++k) cw[k] = coef_words(taps, yFrac, 2 * k);
	for (int y = 0; y < h; ++y)
		for (int x = 0; x < w; x += 8)
		{
			__m128i lo, hi;
			vsum8(taps, rows + 64 * y + x, cw, &lo, &hi);
			/* truncation to int16, not saturation: shift the low halves into sign position and back before the pack */
			lo = _mm_srai_epi32(_mm_slli_epi32(_mm_srai_epi32(lo, 6), 16), 16);
			hi = _mm_srai_epi32(_mm_slli_epi32(_mm_srai_epi32(hi, 6), 16), 16);
			_mm_storeu_si128((__m128i *)(out + 64 * y + x), _mm_packs_epi32(lo, hi));
		}
}

TARGET_SSE41 INLINE void pred_bi(const int taps, uint8_t *dst, ptrdiff_t sd, const uint8_t *ref0, const uint8_t *ref1, ptrdiff_t sr,
		int w, int h, int xFrac0, int yFrac0, int xFrac1, int yFrac1)
{
	int16_t p0[64 * 64 + 8], p1[64 * 64 + 8];
	predict_14bit(taps, p0, ref0, sr, w, h, xFrac0, yFrac0);
	predict_14bit(taps, p1, ref1, sr, w, h, xFrac1, yFrac1);
	const __m128i one = _mm_set1_epi16(1), rnd = _mm_set1_epi32(64);
	for (int y = 0; y < h; ++y, dst += sd)
		for (int x = 0; x < w; x += 8)
		{
			const __m128i a = _mm_loadu_si128((const __m128i *)(p0 + 64 * y + x)), b = _mm_loadu_si128((const __m128i *)(p1 + 64 * y + x));
			const __m128i lo = _mm_srai_epi32(_mm_add_epi32(_mm_madd_epi16(_mm_unpacklo_epi16(a, b), one), rnd), 7);      /* a + b in 32 bits */
			const __m128i hi = _mm_srai_epi32(_mm_add_epi32(_mm_madd_epi16(_mm_unpackhi_epi16(a, b), one), rnd), 7);
			const __m128i s = _mm_packs_epi32(lo, hi);
			store_bytes8(dst + x, _mm_packus_epi16(s, s), w - x);
		}
}

TARGET_SSE41 void hevcasm_sse4_pred_uni_8tap(uint8_t *dst, ptrdiff_t sd, const uint8_t *ref, ptrdiff_t sr, int w, int h, int xFrac, int yFrac)
{
	if (!xFrac && !yFrac) hevcasm_sse2_pred_uni_copy(dst, sd, ref, sr, w, h, 0, 0); else pred_uni(8, dst, sd, ref, sr, w, h, xFrac, yFrac);
}

TARGET_SSE41 void hevcasm_sse4_pred_uni_4tap(uint8_t *dst, ptrdiff_t sd, const uint8_t *ref, ptrdiff_t sr, int w, int h, int xFrac, int yFrac)
{
	if (!xFrac && !yFrac) hevcasm_sse2_pred_uni_copy(dst, sd, ref, sr, w, h, 0, 0); else pred_uni(4, dst, sd, ref, sr, w, h, xFrac, yFrac);
}

TARGET_SSE41 void hevcasm_sse4_pred_bi_8tap(uint8_t *dst, ptrdiff_t sd, const uint8_t *ref0, const uint8_t *ref1, ptrdiff_t sr, int w, int h, int xFrac0, int yFrac0, int xFrac1, int yFrac1)
{
	if (!(xFrac0 | yFrac0 | xFrac1 | yFrac1)) hevcasm_sse2_pred_bi_copy(dst, sd, ref0, ref1, sr, w, h, 0, 0, 0, 0);
	else pred_bi(8, dst, sd, ref0, ref1, sr, w, h, xFrac0, yFrac0, xFrac1, yFrac1);
}

TARGET_SSE41 void hevcasm_sse4_pred_bi_4tap(uint8_t *dst, ptrdiff_t sd, const uint8_t *ref0, const uint8_t *ref1, ptrdiff_t sr, int w, int h, int xFrac0, int yFrac0, int xFrac1, int yFrac1)
{
	if (!(xFrac0 | yFrac0 | xFrac1 | yFrac1)) hevcasm_sse2_pred_bi_copy(dst, sd, ref0, ref1, sr, w, h, 0, 0, 0, 0);
	else pred_bi(4, dst, sd, ref0, ref1, sr, w, h, xFrac0, yFrac0, xFrac1, yFrac1);
}

/* integer vectors: a block copy (pred_inter.c:141-151; pred_inter_a.asm's movdqu loops), exactly w bytes per row */
TARGET_SSE2 void hevcasm_sse2_pred_uni_copy(uint8_t *dst, ptrdiff_t sd, const uint8_t *ref, ptrdiff_t sr, int w, int h, int xFrac, int yFrac)
{
	(void)xFrac; (void)yFrac;
	for (int y = 0; y < h; ++y, dst += sd, ref += sr)
	{
		int x = 0;
		for (; x + 16 <= w; x += 16) _mm_storeu_si128((__m128i *)(dst + x), _mm_loadu_si128((const __m128i *)(ref + x)));
		if (x < w) memcpy(dst + x, ref + x, (size_t)(w - x));
	}
}

/* integer vectors, two references: (64 a + 64 b + 64) >> 7 == (a + b + 1) >> 1 == pavgb (pred_inter_a.asm:580-600) */
TARGET_SSE2 void hevcasm_sse2_pred_bi_copy(uint8_t *dst, ptrdiff_t sd, const uint8_t *ref0, const uint8_t *ref1, ptrdiff_t sr, int w, int h, int xFrac0, int yFrac0, int xFrac1, int yFrac1)
{
	(void)xFrac0; (void)yFrac0; (void)xFrac1; (void)yFrac1;
	for (int y = 0; y < h; ++y, dst += sd, ref0 += sr, ref1 += sr)
	{
		int x = 0;
		for (; x + 16 <= w; x += 16)
			_mm_storeu_si128((__m128i *)(dst + x), _mm_avg_epu8(_mm_loadu_si128((const __m128i *)(ref0 + x)), _mm_loadu_si128((const __m128i *)(ref1 + x))));
		for (; x < w; ++x) dst[x] = (uint8_t)((ref0[x] + ref1[x] + 1) >> 1);
	}
}

#endif
