/*
 * sad_x86.c -- host SIMD implementations of SAD and 4-way SAD behind the reference's own x86 instruction-set bits.
 *
 * The reference fills its SAD tables with x86 assembly when the caller's mask has the bits (reference sad.c:63-86:
 * libvpx's SSE2 kernels for the eleven named block sizes; sad.c:128-167: SSE2 "x4d" kernels for the same sizes and its
 * own AVX2 kernels, sad_a.asm:56-315, for every width 4 ... 64 at any height).  That assembly cannot be built in this
 * toolchain (no yasm / nasm), so a codec that masks HEVCASM_HIP off -- the right thing for per-block calls -- would be
 * left with plain C.  These are the same kernels written with compiler intrinsics: psadbw on 8 / 16 / 32 source bytes at
 * a time, one source load shared by the four candidates.  Selected exactly where the reference selects its assembly
 * (csrc/host/sad.c), verified against the C path by hevcasm_test()'s SSE2 / AVX2 columns, and timed by bench.py as the
 * host-SIMD baseline.  Results are exact integers: bit-identical to the C path by construction.
 *
 * Unaligned loads throughout: the reference's C path has no alignment requirement and neither have these.
 */
#include "sad_x86.h"

#if defined(__x86_64__) || defined(__i386__)

#include <immintrin.h>
#include <string.h>

#define TARGET_SSE2 __attribute__((target("sse2")))
#define TARGET_AVX2 __attribute__((target("avx2")))

static inline uint32_t load32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }

TARGET_SSE2 static inline int hsum_sse2(__m128i v)     /* two 64-bit psadbw sums */
{
	return _mm_cvtsi128_si32(v) + _mm_cvtsi128_si32(_mm_srli_si128(v, 8));
}

/* ---- SSE2: the reference's eleven named sizes (widths 8, 16, 32, 64) ------------------------------------------------- */

/* every kernel body is inlined once per width (a compile-time constant there), so its column loop unrolls fully */
#define INLINE static inline __attribute__((always_inline))

TARGET_SSE2 INLINE int sse2_sad_w(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref, const int w, const int h)
{
	__m128i acc = _mm_setzero_si128();
	if (w == 8)
	{
		for (int y = 0; y < h; ++y, src += stride_src, ref += stride_ref)
			acc = _mm_add_epi32(acc, _mm_sad_epu8(_mm_loadl_epi64((const __m128i *)src), _mm_loadl_epi64((const __m128i *)ref)));
	}
	else
	{
		for (int y = 0; y < h; ++y, src += stride_src, ref += stride_ref)
			for (int x = 0; x < w; x += 16)
				acc = _mm_add_epi32(acc, _mm_sad_epu8(_mm_loadu_si128((const __m128i *)(src + x)), _mm_loadu_si128((const __m128i *)(ref + x))));
	}
	return hsum_sse2(acc);
}

TARGET_SSE2 int hevcasm_sse2_sad(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref, uint32_t rect)
{
	const int h = (int)(rect & 0xff);
	switch (rect >> 8)
	{
	case 8: return sse2_sad_w(src, stride_src, ref, stride_ref, 8, h);
	case 16: return sse2_sad_w(src, stride_src, ref, stride_ref, 16, h);
	case 32: return sse2_sad_w(src, stride_src, ref, stride_ref, 32, h);
	case 48: return sse2_sad_w(src, stride_src, ref, stride_ref, 48, h);
	default: return sse2_sad_w(src, stride_src, ref, stride_ref, 64, h);
	}
}

TARGET_SSE2 INLINE void sse2_sad4_w(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref[], ptrdiff_t stride_ref, int sad[], const int w, const int h)
{
	const uint8_t *r0 = ref[0], *r1 = ref[1], *r2 = ref[2], *r3 = ref[3];
	__m128i a0 = _mm_setzero_si128(), a1 = a0, a2 = a0, a3 = a0;
	for (int y = 0; y < h; ++y)
	{
		if (w == 8)
		{
			const __m128i s = _mm_loadl_epi64((const __m128i *)src);
			a0 = _mm_add_epi32(a0, _mm_sad_epu8(s, _mm_loadl_epi64((const __m128i *)r0)));
			a1 = _mm_add_epi32(a1, _mm_sad_epu8(s, _mm_loadl_epi64((const __m128i *)r1)));
			a2 = _mm_add_epi32(a2, _mm_sad_epu8(s, _mm_loadl_epi64((const __m128i *)r2)));
			a3 = _mm_add_epi32(a3, _mm_sad_epu8(s, _mm_loadl_epi64((const __m128i *)r3)));
		}
		else
		{
			for (int x = 0; x < w; x += 16)
			{
				const __m128i s = _mm_loadu_si128((const __m128i *)(src + x));
				a0 = _mm_add_epi32(a0, _mm_sad_epu8(s, _mm_loadu_si128((const __m128i *)(r0 + x))));
				a1 = _mm_add_epi32(a1, _mm_sad_epu8(s, _mm_loadu_si128((const __m128i *)(r1 + x))));
				a2 = _mm_add_epi32(a2, _mm_sad_epu8(s, _mm_loadu_si128((const __m128i *)(r2 + x))));
				a3 = _mm_add_epi32(a3, _mm_sad_epu8(s, _mm_loadu_si128((const __m128i *)(r3 + x))));
			}
		}
		src += stride_src; r0 += stride_ref; r1 += stride_ref; r2 += stride_ref; r3 += stride_ref;
	}
	sad[0] = hsum_sse2(a0); sad[1] = hsum_sse2(a1); sad[2] = hsum_sse2(a2); sad[3] = hsum_sse2(a3);
}

TARGET_SSE2 void hevcasm_sse2_sad_multiref_4(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref[], ptrdiff_t stride_ref, int sad[], uint32_t rect)
{
	const int h = (int)(rect & 0xff);
	switch (rect >> 8)
	{
	case 8: sse2_sad4_w(src, stride_src, ref, stride_ref, sad, 8, h); break;
	case 16: sse2_sad4_w(src, stride_src, ref, stride_ref, sad, 16, h); break;
	case 32: sse2_sad4_w(src, stride_src, ref, stride_ref, sad, 32, h); break;
	case 48: sse2_sad4_w(src, stride_src, ref, stride_ref, sad, 48, h); break;
	default: sse2_sad4_w(src, stride_src, ref, stride_ref, sad, 64, h); break;
	}
}

/* ---- AVX2: 4-way SAD of any width that is a multiple of 4, keyed on the width like the reference's kernels ------------ */

/* one source load of N bytes against the four candidates */
#define ROW4_256(x) do { \
		const __m256i s_ = _mm256_loadu_si256((const __m256i *)(src + (x))); \
		b0 = _mm256_add_epi32(b0, _mm256_sad_epu8(s_, _mm256_loadu_si256((const __m256i *)(r0 + (x))))); \
		b1 = _mm256_add_epi32(b1, _mm256_sad_epu8(s_, _mm256_loadu_si256((const __m256i *)(r1 + (x))))); \
		b2 = _mm256_add_epi32(b2, _mm256_sad_epu8(s_, _mm256_loadu_si256((const __m256i *)(r2 + (x))))); \
		b3 = _mm256_add_epi32(b3, _mm256_sad_epu8(s_, _mm256_loadu_si256((const __m256i *)(r3 + (x))))); } while (0)
#define ROW4_128(x) do { \
		const __m128i s_ = _mm_loadu_si128((const __m128i *)(src + (x))); \
		a0 = _mm_add_epi32(a0, _mm_sad_epu8(s_, _mm_loadu_si128((const __m128i *)(r0 + (x))))); \
		a1 = _mm_add_epi32(a1, _mm_sad_epu8(s_, _mm_loadu_si128((const __m128i *)(r1 + (x))))); \
		a2 = _mm_add_epi32(a2, _mm_sad_epu8(s_, _mm_loadu_si128((const __m128i *)(r2 + (x))))); \
		a3 = _mm_add_epi32(a3, _mm_sad_epu8(s_, _mm_loadu_si128((const __m128i *)(r3 + (x))))); } while (0)
#define ROW4_64(x) do { \
		const __m128i s_ = _mm_loadl_epi64((const __m128i *)(src + (x))); \
		a0 = _mm_add_epi32(a0, _mm_sad_epu8(s_, _mm_loadl_epi64((const __m128i *)(r0 + (x))))); \
		a1 = _mm_add_epi32(a1, _mm_sad_epu8(s_, _mm_loadl_epi64((const __m128i *)(r1 + (x))))); \
		a2 = _mm_add_epi32(a2, _mm_sad_epu8(s_, _mm_loadl_epi64((const __m128i *)(r2 + (x))))); \
		a3 = _mm_add_epi32(a3, _mm_sad_epu8(s_, _mm_loadl_epi64((const __m128i *)(r3 + (x))))); } while (0)
#define ROW4_32(x) do { \
		const __m128i s_ = _mm_cvtsi32_si128((int)load32(src + (x))); \
		a0 = _mm_add_epi32(a0, _mm_sad_epu8(s_, _mm_cvtsi32_si128((int)load32(r0 + (x))))); \
		a1 = _mm_add_epi32(a1, _mm_sad_epu8(s_, _mm_cvtsi32_si128((int)load32(r1 + (x))))); \
		a2 = _mm_add_epi32(a2, _mm_sad_epu8(s_, _mm_cvtsi32_si128((int)load32(r2 + (x))))); \
		a3 = _mm_add_epi32(a3, _mm_sad_epu8(s_, _mm_cvtsi32_si128((int)load32(r3 + (x))))); } while (0)

TARGET_AVX2 INLINE void avx2_sad4_w(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref[], ptrdiff_t stride_ref, int sad[], const int w, const int h)
{
	const uint8_t *r0 = ref[0], *r1 = ref[1], *r2 = ref[2], *r3 = ref[3];
	__m256i b0 = _mm256_setzero_si256(), b1 = b0, b2 = b0, b3 = b0;
	__m128i a0 = _mm_setzero_si128(), a1 = a0, a2 = a0, a3 = a0;
	for (int y = 0; y < h; ++y)
	{
		int x = 0;
		for (; x + 32 <= w; x += 32) ROW4_256(x);
		if (x + 16 <= w) { ROW4_128(x); x += 16; }
		if (x + 8 <= w) { ROW4_64(x); x += 8; }
		if (x + 4 <= w) { ROW4_32(x); }
		src += stride_src; r0 += stride_ref; r1 += stride_ref; r2 += stride_ref; r3 += stride_ref;
	}
#define TOTAL(k) hsum_sse2(_mm_add_epi32(_mm_add_epi32(_mm256_castsi256_si128(b##k), _mm256_extracti128_si256(b##k, 1)), a##k))
	sad[0] = TOTAL(0); sad[1] = TOTAL(1); sad[2] = TOTAL(2); sad[3] = TOTAL(3);
#undef TOTAL
}

TARGET_AVX2 void hevcasm_avx2_sad_multiref_4(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref[], ptrdiff_t stride_ref, int sad[], uint32_t rect)
{
	const int h = (int)(rect & 0xff);
	switch (rect >> 8)
	{
#define W(n) case n: avx2_sad4_w(src, stride_src, ref, stride_ref, sad, n, h); break;
	W(4) W(8) W(12) W(16) W(20) W(24) W(28) W(32) W(36) W(40) W(44) W(48) W(52) W(56) W(60)
#undef W
	default: avx2_sad4_w(src, stride_src, ref, stride_ref, sad, 64, h); break;
	}
}

/* single reference, widths 32 and 64 (libvpx has AVX2 kernels for 64x64, 64x32, 32x64, 32x32, 32x16; the reference does not wire
 * them, vp9_rtcd.h:468-561 -- offered here behind the AVX2 bit for those five sizes only) */
TARGET_AVX2 INLINE int avx2_sad_w(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref, const int w, const int h)
{
	__m256i acc = _mm256_setzero_si256();
	for (int y = 0; y < h; ++y, src += stride_src, ref += stride_ref)
		for (int x = 0; x < w; x += 32)
			acc = _mm256_add_epi32(acc, _mm256_sad_epu8(_mm256_loadu_si256((const __m256i *)(src + x)), _mm256_loadu_si256((const __m256i *)(ref + x))));
	return hsum_sse2(_mm_add_epi32(_mm256_castsi256_si128(acc), _mm256_extracti128_si256(acc, 1)));
}

TARGET_AVX2 int hevcasm_avx2_sad(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref, uint32_t rect)
{
	const int h = (int)(rect & 0xff);
	return (rect >> 8) == 32 ? avx2_sad_w(src, stride_src, ref, stride_ref, 32, h) : avx2_sad_w(src, stride_src, ref, stride_ref, 64, h);
}

int hevcasm_x86_simd_built(void) { return 1; }

#else

int hevcasm_x86_simd_built(void) { return 0; }

#endif
