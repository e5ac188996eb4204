/*
 * transform_x86.c -- host SIMD inverse transform + add and forward transform behind the reference's own x86 instruction-set bits.
 *
 * The reference selects assembly for these (reference residual_decode.c:461-500, :906-928): SSSE3 partial butterflies for the
 * 8x8 and 16x16 inverse transforms and the 16x16 forward transform (residual_decode_a.asm:224-1104), and f265's AVX2 kernels for
 * the inverse 4x4 (DCT and DST), 8x8, 16x16 and -- encoder only, the reference documents it as non-conforming on extreme
 * coefficients (residual_decode.c:489-495) -- 32x32.  None of that can be assembled in this toolchain (no yasm / nasm), so the
 * same table slots get intrinsics kernels: pmaddwd on pairs of rows, even/odd factorisation, exact 32-bit sums, the reference's
 * shifts, int16 clipping (inverse) and truncation (forward) -- bit-identical to the C path on EVERY input, the 32x32 included
 * (so it is installed whatever the `encoder` flag says).  One body (transform_x86_body.h) is compiled for 128-bit and for 256-bit
 * vectors.  Verified against the C path by hevcasm_test()'s SSSE3 / AVX2 columns and tests/test_cpath.py; timed by bench.py as the
 * host-SIMD baseline beside the GPU's transform rows.
 */
#include "transform_x86.h"

#if defined(__x86_64__) || defined(__i386__)

#include "cpath.h"

#include <immintrin.h>
#include <string.h>

static inline int pair_coef(int lo, int hi) { return (int)(((unsigned)hi << 16) | ((unsigned)lo & 0xffffu)); }

/* even/odd stages: eo_coef[log2 n][i][q] = rows (4 q + 1, 4 q + 3) of the n-point matrix at column i */
static int32_t eo_coef[6][16][8];
/* plain products: [which][i * (N / 2) + q] = (M(2 q, i), M(2 q + 1, i)); which = 0 inverse DST 4, 1 forward DCT 16, 2 forward DCT 4/8/32 unused */
static int32_t direct_idst4[4 * 2], direct_fdct16[16 * 8];

static const signed char dst4_matrix[4][4] = { { 29, 55, 74, 84 }, { 74, 74, 0, -74 }, { 84, -29, -74, 55 }, { 55, -84, 74, -29 } };

__attribute__((constructor)) static void build_tables(void)
{
	for (int lg = 2; lg <= 5; ++lg)
		for (int i = 0; i < (1 << lg) / 2; ++i)
			for (int q = 0; q < (1 << lg) / 4; ++q)
				eo_coef[lg][i][q] = pair_coef(hevcasm_c_transform_coefficient(1 << lg, 4 * q + 1, i), hevcasm_c_transform_coefficient(1 << lg, 4 * q + 3, i));
	for (int i = 0; i < 4; ++i)
		for (int q = 0; q < 2; ++q) direct_idst4[i * 2 + q] = pair_coef(dst4_matrix[2 * q][i], dst4_matrix[2 * q + 1][i]);       /* inverse: M(k, i) = m[k][i] */
	for (int i = 0; i < 16; ++i)
		for (int q = 0; q < 8; ++q)
			direct_fdct16[i * 8 + q] = pair_coef(hevcasm_c_transform_coefficient(16, i, 2 * q), hevcasm_c_transform_coefficient(16, i, 2 * q + 1));   /* forward: M(k, i) = m[i][k] */
}

/* the plain-product table of a configuration: only the inverse 4x4 DST and the forward 16x16 DCT take that path (a third would need its own table:
 * trap rather than multiply by the wrong matrix) */
static inline __attribute__((always_inline)) const int32_t *direct_coef(int N, int DST, int INVERSE)
{
	if (INVERSE && DST && N == 4) return direct_idst4;
	if (!INVERSE && !DST && N == 16) return direct_fdct16;
	__builtin_trap();
}

/* ---- 128-bit vectors: a group of 8 columns is two vectors of four 32-bit sums ---------------------------------------- */

typedef struct { __m128i lo, hi; } vec128x2;
#define TGT __attribute__((target("ssse3")))
#define SFX(name) name##_v128
#define VEC vec128x2
TGT static inline __attribute__((always_inline)) vec128x2 v128_set1(int v) { const vec128x2 r = { _mm_set1_epi32(v), _mm_set1_epi32(v) }; return r; }
TGT static inline __attribute__((always_inline)) vec128x2 v128_pair(__m128i a, __m128i b) { const vec128x2 r = { _mm_unpacklo_epi16(a, b), _mm_unpackhi_epi16(a, b) }; return r; }
TGT static inline __attribute__((always_inline)) vec128x2 v128_madd(vec128x2 p, vec128x2 c) { const vec128x2 r = { _mm_madd_epi16(p.lo, c.lo), _mm_madd_epi16(p.hi, c.hi) }; return r; }
TGT static inline __attribute__((always_inline)) vec128x2 v128_add(vec128x2 a, vec128x2 b) { const vec128x2 r = { _mm_add_epi32(a.lo, b.lo), _mm_add_epi32(a.hi, b.hi) }; return r; }
TGT static inline __attribute__((always_inline)) vec128x2 v128_sub(vec128x2 a, vec128x2 b) { const vec128x2 r = { _mm_sub_epi32(a.lo, b.lo), _mm_sub_epi32(a.hi, b.hi) }; return r; }
TGT static inline __attribute__((always_inline)) __m128i v128_pack_sat(vec128x2 v, int s) { return _mm_packs_epi32(_mm_srai_epi32(v.lo, s), _mm_srai_epi32(v.hi, s)); }
TGT static inline __attribute__((always_inline)) __m128i v128_pack_trunc(vec128x2 v, int s)
{
	/* (int16_t)(v >> s): move the low 16 bits of the shifted value into sign position and back, then the pack cannot saturate */
	return _mm_packs_epi32(_mm_srai_epi32(_mm_slli_epi32(_mm_srai_epi32(v.lo, s), 16), 16), _mm_srai_epi32(_mm_slli_epi32(_mm_srai_epi32(v.hi, s), 16), 16));
}
#define V_SET1 v128_set1
#define V_PAIR v128_pair
#define V_MADD v128_madd
#define V_ADD v128_add
#define V_SUB v128_sub
#define V_PACK_SAT v128_pack_sat
#define V_PACK_TRUNC v128_pack_trunc
#include "transform_x86_body.h"
#undef TGT
#undef SFX
#undef VEC
#undef V_SET1
#undef V_PAIR
#undef V_MADD
#undef V_ADD
#undef V_SUB
#undef V_PACK_SAT
#undef V_PACK_TRUNC

/* ---- 256-bit vectors: a group of 8 columns is one vector --------------------------------------------------------------- */

#define TGT __attribute__((target("avx2")))
#define SFX(name) name##_v256
#define VEC __m256i
TGT static inline __attribute__((always_inline)) __m256i v256_pair(__m128i a, __m128i b)
{
	return _mm256_inserti128_si256(_mm256_castsi128_si256(_mm_unpacklo_epi16(a, b)), _mm_unpackhi_epi16(a, b), 1);
}
TGT static inline __attribute__((always_inline)) __m128i v256_pack_sat(__m256i v, int s)
{
	const __m256i t = _mm256_srai_epi32(v, s);
	return _mm_packs_epi32(_mm256_castsi256_si128(t), _mm256_extracti128_si256(t, 1));
}
TGT static inline __attribute__((always_inline)) __m128i v256_pack_trunc(__m256i v, int s)
{
	const __m256i t = _mm256_srai_epi32(_mm256_slli_epi32(_mm256_srai_epi32(v, s), 16), 16);
	return _mm_packs_epi32(_mm256_castsi256_si128(t), _mm256_extracti128_si256(t, 1));
}
#define V_SET1 _mm256_set1_epi32
#define V_PAIR v256_pair
#define V_MADD _mm256_madd_epi16
#define V_ADD _mm256_add_epi32
#define V_SUB _mm256_sub_epi32
#define V_PACK_SAT v256_pack_sat
#define V_PACK_TRUNC v256_pack_trunc
#include "transform_x86_body.h"

/* ---- table entries --------------------------------------------------------------------------------------------------- */

#define T128 __attribute__((target("ssse3")))
#define T256 __attribute__((target("avx2")))

T128 void hevcasm_ssse3_idct_8(uint8_t *d, ptrdiff_t sd, const uint8_t *p, ptrdiff_t sp, const int16_t *c) { inverse_add_v128(8, 0, d, sd, p, sp, c); }
T128 void hevcasm_ssse3_idct_16(uint8_t *d, ptrdiff_t sd, const uint8_t *p, ptrdiff_t sp, const int16_t *c) { inverse_add_v128(16, 0, d, sd, p, sp, c); }
T128 void hevcasm_ssse3_dct_16(int16_t *c, const int16_t *s, ptrdiff_t ss) { forward_v128(16, 4, c, s, ss); }

T256 void hevcasm_avx2_idst_4(uint8_t *d, ptrdiff_t sd, const uint8_t *p, ptrdiff_t sp, const int16_t *c) { inverse_add_v256(4, 1, d, sd, p, sp, c); }
T256 void hevcasm_avx2_idct_4(uint8_t *d, ptrdiff_t sd, const uint8_t *p, ptrdiff_t sp, const int16_t *c) { inverse_add_v256(4, 0, d, sd, p, sp, c); }
T256 void hevcasm_avx2_idct_8(uint8_t *d, ptrdiff_t sd, const uint8_t *p, ptrdiff_t sp, const int16_t *c) { inverse_add_v256(8, 0, d, sd, p, sp, c); }
T256 void hevcasm_avx2_idct_16(uint8_t *d, ptrdiff_t sd, const uint8_t *p, ptrdiff_t sp, const int16_t *c) { inverse_add_v256(16, 0, d, sd, p, sp, c); }
T256 void hevcasm_avx2_idct_32(uint8_t *d, ptrdiff_t sd, const uint8_t *p, ptrdiff_t sp, const int16_t *c) { inverse_add_v256(32, 0, d, sd, p, sp, c); }

#endif
