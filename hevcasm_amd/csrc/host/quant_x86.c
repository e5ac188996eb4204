/*
 * quant_x86.c -- host SIMD (de)quantisation, reconstruction and SSD behind the reference's own x86 instruction-set bits.
 *
 * The reference selects assembly for these under HEVCASM_SSE41 (quantize.c:65-79 dequantise, :189-204 quantise, :305-330
 * reconstruct; quantize_a.asm:53-329) and under HEVCASM_AVX for the 16x16 / 32x32 / 64x64 SSD (ssd.c:63-86; ssd_a.asm:47-185).
 * None of it can be assembled in this toolchain (no yasm / nasm); these are intrinsics kernels for the same table slots, with the
 * C path's arithmetic (reference quantize.c:53-62, :160-186, :292-302; ssd.c:43-55) in 32-bit lanes: bit-identical on every
 * input, abs(-32768) = 32768 included (the reference's own SSE4 quantiser uses pabsw and differs there: SURVEY.md appendix A).
 * No alignment or n % 16 precondition: tails are finished in scalar code.
 */
#include "quant_x86.h"

#if defined(__x86_64__) || defined(__i386__)

#include <immintrin.h>
#include <string.h>

#define TARGET_SSE41 __attribute__((target("sse4.1")))
#define TARGET_AVX __attribute__((target("avx")))

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

TARGET_SSE41 void hevcasm_sse4_quantize_inverse(int16_t *dst, const int16_t *src, int scale, int shift, int n)
{
	const __m128i vs = _mm_set1_epi32(scale), vr = _mm_set1_epi32(1 << (shift - 1));
	int i = 0;
	for (; i + 8 <= n; i += 8)
	{
		const __m128i v = _mm_loadu_si128((const __m128i *)(src + i));
		const __m128i lo = _mm_srai_epi32(_mm_add_epi32(_mm_mullo_epi32(_mm_cvtepi16_epi32(v), vs), vr), shift);
		const __m128i hi = _mm_srai_epi32(_mm_add_epi32(_mm_mullo_epi32(_mm_cvtepi16_epi32(_mm_srli_si128(v, 8)), vs), vr), shift);
		_mm_storeu_si128((__m128i *)(dst + i), _mm_packs_epi32(lo, hi));      /* the pack is the clip to int16 */
	}
	for (; i < n; ++i) dst[i] = (int16_t)clampi((src[i] * scale + (1 << (shift - 1))) >> shift, -32768, 32767);
}

TARGET_SSE41 int hevcasm_sse4_quantize(int16_t *dst, const int16_t *src, int scale, int shift, int offset, int n)
{
	const int round = offset << (shift - 16);
	const __m128i vs = _mm_set1_epi32(scale), vr = _mm_set1_epi32(round);
	__m128i any = _mm_setzero_si128();
	int i = 0, cbf = 0;
	for (; i + 8 <= n; i += 8)
	{
		const __m128i v = _mm_loadu_si128((const __m128i *)(src + i));
		const __m128i a = _mm_cvtepi16_epi32(v), b = _mm_cvtepi16_epi32(_mm_srli_si128(v, 8));
		/* |x| in 32 bits (32768 for -32768), (|x| scale + round) >> shift, then the sign by mask: (q ^ m) - m with m = x >> 31.  Not _mm_sign_epi32,
		 * which returns 0 for x == 0 where the C path keeps +round >> shift (non-zero once offset >= 0x10000: outside the reference's asserted
		 * domain, quantize.c:163, but this entry has no assert and must still equal the C path there) */
		const __m128i ma = _mm_srai_epi32(a, 31), mb = _mm_srai_epi32(b, 31);
		const __m128i qa = _mm_sub_epi32(_mm_xor_si128(_mm_srai_epi32(_mm_add_epi32(_mm_mullo_epi32(_mm_abs_epi32(a), vs), vr), shift), ma), ma);
		const __m128i qb = _mm_sub_epi32(_mm_xor_si128(_mm_srai_epi32(_mm_add_epi32(_mm_mullo_epi32(_mm_abs_epi32(b), vs), vr), shift), mb), mb);
		const __m128i q = _mm_packs_epi32(qa, qb);
		any = _mm_or_si128(any, q);
		_mm_storeu_si128((__m128i *)(dst + i), q);
	}
	for (; i < n; ++i)
	{
		const int v = src[i], mag = ((v < 0 ? -v : v) * scale + round) >> shift;
		const int q = clampi(v < 0 ? -mag : mag, -32768, 32767);
		cbf |= q;
		dst[i] = (int16_t)q;
	}
	/* OR of the (sign-extended) levels, as the C path returns it */
	int16_t lanes[8];
	_mm_storeu_si128((__m128i *)lanes, any);
	for (int k = 0; k < 8; ++k) cbf |= lanes[k];
	return cbf;
}

TARGET_SSE41 void hevcasm_sse4_quantize_reconstruct(uint8_t *rec, ptrdiff_t stride_rec, const uint8_t *pred, ptrdiff_t stride_pred, const int16_t *res, int n)
{
	for (int y = 0; y < n; ++y, rec += stride_rec, pred += stride_pred, res += n)
	{
		int x = 0;
		for (; x + 8 <= n; x += 8)
		{
			/* saturating 16-bit add, then the unsigned-saturating pack: equal to clip8 of the exact sum for every int16 residual */
			const __m128i s = _mm_adds_epi16(_mm_cvtepu8_epi16(_mm_loadl_epi64((const __m128i *)(pred + x))), _mm_loadu_si128((const __m128i *)(res + x)));
			_mm_storel_epi64((__m128i *)(rec + x), _mm_packus_epi16(s, s));
		}
		for (; x < n; ++x) rec[x] = (uint8_t)clampi(pred[x] + res[x], 0, 255);
	}
}

TARGET_AVX int hevcasm_avx_ssd(const uint8_t *a, ptrdiff_t stride_a, const uint8_t *b, ptrdiff_t stride_b, int w, int h)
{
	__m128i acc = _mm_setzero_si128();
	int tail = 0;
	const __m128i zero = _mm_setzero_si128();
	for (int y = 0; y < h; ++y, a += stride_a, b += stride_b)
	{
		int x = 0;
		for (; x + 16 <= w; x += 16)
		{
			const __m128i va = _mm_loadu_si128((const __m128i *)(a + x)), vb = _mm_loadu_si128((const __m128i *)(b + x));
			const __m128i d0 = _mm_sub_epi16(_mm_unpacklo_epi8(va, zero), _mm_unpacklo_epi8(vb, zero));
			const __m128i d1 = _mm_sub_epi16(_mm_unpackhi_epi8(va, zero), _mm_unpackhi_epi8(vb, zero));
			acc = _mm_add_epi32(acc, _mm_add_epi32(_mm_madd_epi16(d0, d0), _mm_madd_epi16(d1, d1)));     /* at most 64 * 64 * 255^2 < 2^31 in all */
		}
		for (; x < w; ++x) tail += (a[x] - b[x]) * (a[x] - b[x]);
	}
	acc = _mm_add_epi32(acc, _mm_srli_si128(acc, 8));
	acc = _mm_add_epi32(acc, _mm_srli_si128(acc, 4));
	return _mm_cvtsi128_si32(acc) + tail;
}

#endif
