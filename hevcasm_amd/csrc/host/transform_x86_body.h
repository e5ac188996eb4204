/*
 * transform_x86_body.h -- the body of the host SIMD transforms, included twice by transform_x86.c: once with 128-bit vectors
 * (two per group of 8 columns; selected by the reference's SSSE3 bit) and once with 256-bit ones (AVX2 bit).
 *
 * The including file defines
 *   SFX(name)        name mangler               TGT            target attribute
 *   VEC              8 x int32, one per column  V_SET1(int)    broadcast
 *   V_PAIR(a, b)     the 8 (a[x], b[x]) int16 pairs of two rows (__m128i of 8 int16 each), one pair per int32 lane, column order
 *   V_MADD(p, c)     pmaddwd                    V_ADD / V_SUB
 *   V_PACK_SAT(v, s) __m128i of 8 x clip16(v >> s)       V_PACK_TRUNC(v, s) __m128i of 8 x (int16_t)(v >> s)
 *
 * One routine serves both passes of every transform: a 1-D transform down the rows of an N-row int16 matrix, for 8 columns at a
 * time, its results handed out row by row.  Inverse DCT: the even/odd factorisation (the odd rows of each stage two by two into
 * pmaddwd, exact 32-bit sums: equal to the plain matrix product, reference residual_decode.c:91-347, bit for bit); DST-VII and the
 * forward transforms: the plain matrix product on row pairs.
 */

/* rows y0 .. y0 + 7 of `r` (8 int16 each; the caller zeroes rows past N) -> their transpose */
TGT static inline __attribute__((always_inline)) void SFX(transpose8)(const __m128i *r, __m128i *t)
{
	const __m128i a0 = _mm_unpacklo_epi16(r[0], r[1]), a1 = _mm_unpackhi_epi16(r[0], r[1]);
	const __m128i a2 = _mm_unpacklo_epi16(r[2], r[3]), a3 = _mm_unpackhi_epi16(r[2], r[3]);
	const __m128i a4 = _mm_unpacklo_epi16(r[4], r[5]), a5 = _mm_unpackhi_epi16(r[4], r[5]);
	const __m128i a6 = _mm_unpacklo_epi16(r[6], r[7]), a7 = _mm_unpackhi_epi16(r[6], r[7]);
	const __m128i b0 = _mm_unpacklo_epi32(a0, a2), b1 = _mm_unpackhi_epi32(a0, a2);
	const __m128i b2 = _mm_unpacklo_epi32(a1, a3), b3 = _mm_unpackhi_epi32(a1, a3);
	const __m128i b4 = _mm_unpacklo_epi32(a4, a6), b5 = _mm_unpackhi_epi32(a4, a6);
	const __m128i b6 = _mm_unpacklo_epi32(a5, a7), b7 = _mm_unpackhi_epi32(a5, a7);
	t[0] = _mm_unpacklo_epi64(b0, b4); t[1] = _mm_unpackhi_epi64(b0, b4);
	t[2] = _mm_unpacklo_epi64(b1, b5); t[3] = _mm_unpackhi_epi64(b1, b5);
	t[4] = _mm_unpacklo_epi64(b2, b6); t[5] = _mm_unpackhi_epi64(b2, b6);
	t[6] = _mm_unpacklo_epi64(b3, b7); t[7] = _mm_unpackhi_epi64(b3, b7);
}

/* up to 8 int16 of a row (N = 4: the upper half is zero) */
TGT static inline __attribute__((always_inline)) __m128i SFX(load_row)(const int16_t *p, const int N)
{
	return N >= 8 ? _mm_loadu_si128((const __m128i *)p) : _mm_loadl_epi64((const __m128i *)p);
}

/* out[i] (i < N) = 8 columns of   pack((bias + sum_k M(k, i) * in[k]) >> shift),  M = the N-point HEVC matrix (DST: the 4-point
 * DST-VII one) read as M(k, i) = m[k][i] for the inverse (INVERSE = 1) or m[i][k] for the forward transform.  `in` = N rows of
 * `stride` int16, of which columns [0, 8) (N = 4: [0, 4)) are transformed. */
TGT static inline __attribute__((always_inline)) void SFX(cols8)(const int N, const int DST, const int INVERSE, const int TRUNC,
		const int16_t *in, ptrdiff_t stride, int bias, int shift, __m128i *out)
{
	VEC e[32];
	if (INVERSE && !DST)
	{
		VEC o[16], t[16];
		{
			const VEC p = V_PAIR(SFX(load_row)(in, N), SFX(load_row)(in + (N / 2) * stride, N));
			e[0] = V_ADD(V_MADD(p, V_SET1(pair_coef(64, 64))), V_SET1(bias));
			e[1] = V_ADD(V_MADD(p, V_SET1(pair_coef(64, -64))), V_SET1(bias));
		}
		for (int n = 4, lg = 2; n <= N; n *= 2, ++lg)
		{
			const int step = N / n;        /* this stage's inputs are rows step * (2 j + 1) */
			VEC pr[8];
			for (int q = 0; q < n / 4; ++q)
				pr[q] = V_PAIR(SFX(load_row)(in + (step * (4 * q + 1)) * stride, N), SFX(load_row)(in + (step * (4 * q + 3)) * stride, N));
			for (int i = 0; i < n / 2; ++i)
			{
				VEC acc = V_MADD(pr[0], V_SET1(eo_coef[lg][i][0]));       /* rows (4 q + 1, 4 q + 3) of the n-point matrix, column i */
				for (int q = 1; q < n / 4; ++q) acc = V_ADD(acc, V_MADD(pr[q], V_SET1(eo_coef[lg][i][q])));
				o[i] = acc;
			}
			for (int i = 0; i < n / 2; ++i) t[i] = e[i];
			for (int i = 0; i < n / 2; ++i)
			{
				e[i] = V_ADD(t[i], o[i]);
				e[n - 1 - i] = V_SUB(t[i], o[i]);
			}
		}
	}
	else
	{
		VEC pr[16];
		for (int q = 0; q < N / 2; ++q) pr[q] = V_PAIR(SFX(load_row)(in + (2 * q) * stride, N), SFX(load_row)(in + (2 * q + 1) * stride, N));
		for (int i = 0; i < N; ++i)
		{
			VEC acc = V_SET1(bias);
			for (int q = 0; q < N / 2; ++q) acc = V_ADD(acc, V_MADD(pr[q], V_SET1(direct_coef(N, DST, INVERSE)[i * (N / 2) + q])));
			e[i] = acc;
		}
	}
	for (int i = 0; i < N; ++i) out[i] = TRUNC ? V_PACK_TRUNC(e[i], shift) : V_PACK_SAT(e[i], shift);
}

/* one pass over an N x N int16 matrix `in` (row stride `stride`): transform down the rows, write the result TRANSPOSED
 * (out[x][i] = result row i, column x; out has row stride N) -- or as it comes (out[i][x]) when TRANSPOSED = 0 */
TGT static inline __attribute__((always_inline)) void SFX(pass)(const int N, const int DST, const int INVERSE, const int TRUNC, const int TRANSPOSED,
		const int16_t *in, ptrdiff_t stride, int bias, int shift, int16_t *out)
{
	const int G = N < 8 ? N : 8;          /* columns per group */
	for (int x0 = 0; x0 < N; x0 += 8)
	{
		__m128i r[32 + 8];
		SFX(cols8)(N, DST, INVERSE, TRUNC, in + x0, stride, bias, shift, r);
		if (!TRANSPOSED)
		{
			for (int i = 0; i < N; ++i)
			{
				if (G == 8) _mm_storeu_si128((__m128i *)(out + i * N + x0), r[i]); else _mm_storel_epi64((__m128i *)(out + i * N + x0), r[i]);
			}
			continue;
		}
		for (int i = N; i < ((N + 7) & ~7); ++i) r[i] = _mm_setzero_si128();
		for (int i0 = 0; i0 < N; i0 += 8)
		{
			__m128i t[8];
			SFX(transpose8)(r + i0, t);
			for (int j = 0; j < G; ++j)
			{
				if (N >= 8) _mm_storeu_si128((__m128i *)(out + (x0 + j) * N + i0), t[j]); else _mm_storel_epi64((__m128i *)(out + (x0 + j) * N + i0), t[j]);
			}
		}
	}
}

/* inverse transform + add (residual_decode.c:371-413): columns with shift 7, rows with shift 12, both clipped to int16; then
 * clip8(pred + residual) (the narrowing of the sum to int16, residual_decode.c:365, can never act: DESIGN.md) */
TGT static inline __attribute__((always_inline)) void SFX(inverse_add)(const int N, const int DST, uint8_t *dst, ptrdiff_t sd, const uint8_t *pred, ptrdiff_t sp, const int16_t *coeffs)
{
	int16_t tmp[32 * 32 + 8], res[32 * 32 + 8];
	SFX(pass)(N, DST, 1, 0, 1, coeffs, N, 64, 7, tmp);
	SFX(pass)(N, DST, 1, 0, 1, tmp, N, 2048, 12, res);
	for (int y = 0; y < N; ++y, dst += sd, pred += sp)
	{
		if (N == 4)
		{
			uint32_t p4;
			memcpy(&p4, pred, 4);
			const __m128i s = _mm_adds_epi16(_mm_unpacklo_epi8(_mm_cvtsi32_si128((int)p4), _mm_setzero_si128()), _mm_loadl_epi64((const __m128i *)(res + 4 * y)));
			p4 = (uint32_t)_mm_cvtsi128_si32(_mm_packus_epi16(s, s));
			memcpy(dst, &p4, 4);
		}
		else
			for (int x = 0; x < N; x += 8)
			{
				const __m128i s = _mm_adds_epi16(_mm_unpacklo_epi8(_mm_loadl_epi64((const __m128i *)(pred + x)), _mm_setzero_si128()), _mm_loadu_si128((const __m128i *)(res + N * y + x)));
				_mm_storel_epi64((__m128i *)(dst + x), _mm_packus_epi16(s, s));
			}
	}
}

/* forward transform (residual_decode.c:855-892): rows of src with shift log2 N - 1, then columns with shift log2 N + 6; int16 by truncation */
TGT static inline __attribute__((always_inline)) void SFX(forward)(const int N, const int LOG2, int16_t *coeffs, const int16_t *src, ptrdiff_t ss)
{
	int16_t srcT[32 * 32 + 8], t[32 * 32 + 8];
	/* srcT[x][y] = src[y][x]: the first pass runs along the rows of src */
	for (int x0 = 0; x0 < N; x0 += 8)
		for (int y0 = 0; y0 < N; y0 += 8)
		{
			__m128i r[8], q[8];
			for (int j = 0; j < 8; ++j) r[j] = _mm_loadu_si128((const __m128i *)(src + (y0 + j) * ss + x0));
			SFX(transpose8)(r, q);
			for (int j = 0; j < 8; ++j) _mm_storeu_si128((__m128i *)(srcT + (x0 + j) * N + y0), q[j]);
		}
	SFX(pass)(N, 0, 0, 1, 1, srcT, N, 1 << (LOG2 - 2), LOG2 - 1, t);        /* t[y][k] */
	SFX(pass)(N, 0, 0, 1, 0, t, N, 1 << (LOG2 + 5), LOG2 + 6, coeffs);     /* coeffs[k][x] */
}
