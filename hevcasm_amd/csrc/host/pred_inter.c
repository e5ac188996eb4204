/*
 * pred_inter.c -- selectors and self-tests for uni- and bi-directional inter prediction.
 *
 * The reference's tables are keyed by (taps, width class, fractional or not)
 * (pred_inter.c:351-367, :599-612) because its SIMD kernels are width-specialised.  One
 * implementation per tap count serves every width and fraction here, so all slots of a tap
 * count receive the same pointer.  Self-tests follow pred_inter.c:371-486 (uni) and :616-739
 * (bi): 24 / 22 PU sizes (halved for chroma), strides 192, reference origin at row 8 of an
 * 80-row random plane, fractions {0,1} x {0,1}.
 */
#include "pred_inter.h"
#include "cpath.h"
#include "perblock.h"
#include "pred_x86.h"
#include "hevcasm_test.h"

#include <stdlib.h>
#include <string.h>

void hevcasm_populate_pred_uni_8to8(hevcasm_table_pred_uni_8to8 *table, hevcasm_instruction_set mask)
{
	mask = (hevcasm_instruction_set)hevcasm_hip1_effective_mask(mask);   /* HEVCASM_HIP only while the device is usable and unlatched */
	for (int taps = 4; taps <= 8; taps += 4)
	{
		hevcasm_pred_uni_8to8 *f = 0;
		if (mask & (HEVCASM_C_REF | HEVCASM_C_OPT)) f = taps == 8 ? hevcasm_c_pred_uni_8tap : hevcasm_c_pred_uni_4tap;
		hevcasm_pred_uni_8to8 *copy = f, *filter = f;
#if defined(__x86_64__) || defined(__i386__)
		/* where the reference selects its assembly (pred_inter.c:296-346): SSE2 block copies for integer vectors, SSE4.1 filters otherwise */
		if (mask & HEVCASM_SSE2) copy = hevcasm_sse2_pred_uni_copy;
		if (mask & HEVCASM_SSE41) filter = taps == 8 ? hevcasm_sse4_pred_uni_8tap : hevcasm_sse4_pred_uni_4tap;
#endif
		if (mask & HEVCASM_HIP) copy = filter = taps == 8 ? hevcasm_hip1_pred_uni_8tap : hevcasm_hip1_pred_uni_4tap;
		for (int w = 0; w <= 8 * taps; w += taps)
			for (int xFrac = 0; xFrac < 2; ++xFrac)
				for (int yFrac = 0; yFrac < 2; ++yFrac)
					*hevcasm_get_pred_uni_8to8(table, taps, w, 0, xFrac, yFrac) = (xFrac | yFrac) ? filter : copy;
	}
}

void hevcasm_populate_pred_bi_8to8(hevcasm_table_pred_bi_8to8 *table, hevcasm_instruction_set mask)
{
	mask = (hevcasm_instruction_set)hevcasm_hip1_effective_mask(mask);   /* HEVCASM_HIP only while the device is usable and unlatched */
	for (int taps = 4; taps <= 8; taps += 4)
	{
		hevcasm_pred_bi_8to8 *f = 0;
		if (mask & (HEVCASM_C_REF | HEVCASM_C_OPT)) f = taps == 8 ? hevcasm_c_pred_bi_8tap : hevcasm_c_pred_bi_4tap;
		hevcasm_pred_bi_8to8 *copy = f, *filter = f;
#if defined(__x86_64__) || defined(__i386__)
		/* reference pred_inter.c:570-593: pavgb when all four fractions are zero (SSE2), the SSE4.1 filters otherwise */
		if (mask & HEVCASM_SSE2) copy = hevcasm_sse2_pred_bi_copy;
		if (mask & HEVCASM_SSE41) filter = taps == 8 ? hevcasm_sse4_pred_bi_8tap : hevcasm_sse4_pred_bi_4tap;
#endif
		if (mask & HEVCASM_HIP) copy = filter = taps == 8 ? hevcasm_hip1_pred_bi_8tap : hevcasm_hip1_pred_bi_4tap;
		for (int w = 0; w <= 8 * taps; w += 2 * taps)
			for (int frac = 0; frac < 2; ++frac)
				*hevcasm_get_pred_bi_8to8(table, taps, w, 0, frac, frac, frac, frac) = frac ? filter : copy;
	}
}

/* luma PU sizes of reference pred_inter.c:438-444; the bi test skips 8x4 and 4x8 (:687-693) */
static const unsigned char pu_sizes[24][2] = {
	{ 8, 4 }, { 4, 8 }, { 8, 8 },
	{ 16, 4 }, { 16, 8 }, { 16, 12 }, { 16, 16 }, { 12, 16 }, { 8, 16 }, { 4, 16 },
	{ 32, 8 }, { 32, 16 }, { 32, 24 }, { 32, 32 }, { 24, 32 }, { 16, 32 }, { 8, 32 },
	{ 64, 16 }, { 64, 32 }, { 64, 48 }, { 64, 64 }, { 48, 64 }, { 32, 64 }, { 16, 64 } };

#define TEST_STRIDE 192

typedef struct
{
	HEVCASM_ALIGN(32, uint8_t, dst[64 * TEST_STRIDE]);
	hevcasm_pred_uni_8to8 *f;
	const uint8_t *ref;
	int w, h, xFrac, yFrac, taps;
} bound_pred_uni;

static int get_pred_uni(void *p, hevcasm_instruction_set mask)
{
	bound_pred_uni *s = (bound_pred_uni *)p;
	hevcasm_table_pred_uni_8to8 table;
	hevcasm_populate_pred_uni_8to8(&table, mask);
	s->f = *hevcasm_get_pred_uni_8to8(&table, s->taps, s->w, s->h, s->xFrac, s->yFrac);
	if (s->f && mask == HEVCASM_C_REF)
		printf("\t%dx%d %d-tap %s%s : ", s->w, s->h, s->taps, s->xFrac ? "H" : "", s->yFrac ? "V" : "");
	memset(s->dst, 0, sizeof(s->dst));
	return s->f != 0;
}

static void invoke_pred_uni(void *p, int n)
{
	bound_pred_uni *s = (bound_pred_uni *)p;
	while (n--) s->f(s->dst, TEST_STRIDE, s->ref, TEST_STRIDE, s->w, s->h, s->xFrac, s->yFrac);
}

static int mismatch_pred_uni(void *a, void *b)
{
	const bound_pred_uni *r = (const bound_pred_uni *)a, *t = (const bound_pred_uni *)b;
	for (int y = 0; y < r->h; ++y)
		if (memcmp(&r->dst[y * TEST_STRIDE], &t->dst[y * TEST_STRIDE], (size_t)r->w)) return 1;
	return 0;
}

void hevcasm_test_pred_uni(int *error_count, hevcasm_instruction_set mask)
{
	printf("\nhevcasm_pred_uni - Unireference Inter Prediction (single-reference motion compensation)\n");

	static bound_pred_uni b[2];
	static HEVCASM_ALIGN(32, uint8_t, ref[80 * TEST_STRIDE]);
	for (int i = 0; i < 80 * TEST_STRIDE; ++i) ref[i] = (uint8_t)rand();
	b[0].ref = ref + 8 * TEST_STRIDE;

	for (b[0].taps = 8; b[0].taps >= 4; b[0].taps -= 4)
		for (b[0].yFrac = 0; b[0].yFrac < 2; ++b[0].yFrac)
			for (b[0].xFrac = 0; b[0].xFrac < 2; ++b[0].xFrac)
				for (int k = 0; k < 24; ++k)
				{
					b[0].w = pu_sizes[k][0] * b[0].taps / 8;
					b[0].h = pu_sizes[k][1] * b[0].taps / 8;
					b[1] = b[0];
					*error_count += hevcasm_test(&b[0], &b[1], get_pred_uni, invoke_pred_uni, mismatch_pred_uni, mask, 1000);
				}
}

typedef struct
{
	hevcasm_pred_bi_8to8 *f;
	HEVCASM_ALIGN(32, uint8_t, dst[64 * TEST_STRIDE]);
	const uint8_t *refA, *refB;
	int w, h, xFracA, yFracA, xFracB, yFracB, taps;
} bound_pred_bi;

static int get_pred_bi(void *p, hevcasm_instruction_set mask)
{
	bound_pred_bi *s = (bound_pred_bi *)p;
	hevcasm_table_pred_bi_8to8 table;
	hevcasm_populate_pred_bi_8to8(&table, mask);
	s->f = *hevcasm_get_pred_bi_8to8(&table, s->taps, s->w, s->h, s->xFracA, s->yFracA, s->xFracB, s->yFracB);
	if (s->f && mask == HEVCASM_C_REF)
		printf("\t%dx%d %d-tap %s%s %s%s : ", s->w, s->h, s->taps, s->xFracA ? "H" : "", s->yFracA ? "V" : "", s->xFracB ? "H" : "", s->yFracB ? "V" : "");
	memset(s->dst, 0, sizeof(s->dst));
	return s->f != 0;
}

static void invoke_pred_bi(void *p, int n)
{
	bound_pred_bi *s = (bound_pred_bi *)p;
	while (n--) s->f(s->dst, TEST_STRIDE, s->refA, s->refB, TEST_STRIDE, s->w, s->h, s->xFracA, s->yFracA, s->xFracB, s->yFracB);
}

static int mismatch_pred_bi(void *a, void *b)
{
	const bound_pred_bi *r = (const bound_pred_bi *)a, *t = (const bound_pred_bi *)b;
	for (int y = 0; y < r->h; ++y)
		if (memcmp(&r->dst[y * TEST_STRIDE], &t->dst[y * TEST_STRIDE], (size_t)r->w)) return 1;
	return 0;
}

void hevcasm_test_pred_bi(int *error_count, hevcasm_instruction_set mask)
{
	printf("\nhevcasm_pred_bi - Bireference Inter Prediction (two-reference motion compensation)\n");

	static bound_pred_bi b[2];
	static HEVCASM_ALIGN(32, uint8_t, ref[2][80 * TEST_STRIDE]);
	for (int i = 0; i < 80 * TEST_STRIDE; ++i)
	{
		ref[0][i] = (uint8_t)rand();
		ref[1][i] = (uint8_t)rand();
	}
	b[0].refA = ref[0] + 8 * TEST_STRIDE;
	b[0].refB = ref[1] + 8 * TEST_STRIDE;

	for (b[0].taps = 8; b[0].taps >= 4; b[0].taps -= 4)
		for (int frac = 0; frac < 2; ++frac)
		{
			b[0].xFracA = b[0].yFracA = b[0].xFracB = b[0].yFracB = frac;
			for (int k = 0; k < 24; ++k)
			{
				if (pu_sizes[k][0] + pu_sizes[k][1] == 12) continue; /* 8x4, 4x8: not bi-predicted in HEVC */
				b[0].w = pu_sizes[k][0] * b[0].taps / 8;
				b[0].h = pu_sizes[k][1] * b[0].taps / 8;
				b[1] = b[0];
				*error_count += hevcasm_test(&b[0], &b[1], get_pred_bi, invoke_pred_bi, mismatch_pred_bi, mask, 1000);
			}
		}
}
