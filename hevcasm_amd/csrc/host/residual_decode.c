/*
 * residual_decode.c -- selectors and self-tests for inverse transform + add and forward transform.
 *
 * Selection as reference residual_decode.c:461-510 and :906-938, with HEVCASM_HIP as the only
 * accelerated bit.  Self-tests as :514-589 (full-range 16-bit random coefficients) and
 * :941-1015 (9-bit residuals, source stride 32).
 */
#include "residual_decode.h"
#include "cpath.h"
#include "perblock.h"
#include "transform_x86.h"
#include "hevcasm_test.h"

#include <stdlib.h>
#include <string.h>

void hevcasm_populate_inverse_transform_add(hevcasm_table_inverse_transform_add *table, hevcasm_instruction_set mask, int encoder)
{
	mask = (hevcasm_instruction_set)hevcasm_hip1_effective_mask(mask);   /* HEVCASM_HIP only while the device is usable and unlatched */
	static hevcasm_inverse_transform_add *const c_path[5] = { hevcasm_c_idst_4, hevcasm_c_idct_4, hevcasm_c_idct_8, hevcasm_c_idct_16, hevcasm_c_idct_32 };
	static hevcasm_inverse_transform_add *const hip_path[5] = { hevcasm_hip1_idst_4, hevcasm_hip1_idct_4, hevcasm_hip1_idct_8, hevcasm_hip1_idct_16, hevcasm_hip1_idct_32 };
	hevcasm_inverse_transform_add *f[5] = { 0, 0, 0, 0, 0 };
	(void)encoder; /* every kernel here is exact on all inputs: residual_decode.h */
	if (mask & (HEVCASM_C_REF | HEVCASM_C_OPT)) memcpy(f, c_path, sizeof(f));
#if defined(__x86_64__)
	/* where the reference selects its assembly (residual_decode.c:475-497); our AVX2 32x32 is exact, so it needs no `encoder` condition */
	if (mask & HEVCASM_SSSE3) { f[2] = hevcasm_ssse3_idct_8; f[3] = hevcasm_ssse3_idct_16; }
	if (mask & HEVCASM_AVX2) { f[0] = hevcasm_avx2_idst_4; f[1] = hevcasm_avx2_idct_4; f[2] = hevcasm_avx2_idct_8; f[3] = hevcasm_avx2_idct_16; f[4] = hevcasm_avx2_idct_32; }
#endif
	if (mask & HEVCASM_HIP) memcpy(f, hip_path, sizeof(f));
	*hevcasm_get_inverse_transform_add(table, 1, 2) = f[0];
	for (int log2 = 2; log2 <= 5; ++log2)
		*hevcasm_get_inverse_transform_add(table, 0, log2) = f[log2 - 1];
}

void hevcasm_populate_transform(hevcasm_table_transform *table, hevcasm_instruction_set mask)
{
	mask = (hevcasm_instruction_set)hevcasm_hip1_effective_mask(mask);   /* HEVCASM_HIP only while the device is usable and unlatched */
	static hevcasm_transform *const c_path[5] = { hevcasm_c_dst_4, hevcasm_c_dct_4, hevcasm_c_dct_8, hevcasm_c_dct_16, hevcasm_c_dct_32 };
	static hevcasm_transform *const hip_path[5] = { hevcasm_hip1_dst_4, hevcasm_hip1_dct_4, hevcasm_hip1_dct_8, hevcasm_hip1_dct_16, hevcasm_hip1_dct_32 };
	hevcasm_transform *f[5] = { 0, 0, 0, 0, 0 };
	if (mask & (HEVCASM_C_REF | HEVCASM_C_OPT)) memcpy(f, c_path, sizeof(f));
#if defined(__x86_64__)
	if (mask & HEVCASM_SSSE3) f[3] = hevcasm_ssse3_dct_16;      /* reference residual_decode.c:918-923 */
#endif
	if (mask & HEVCASM_HIP) memcpy(f, hip_path, sizeof(f));
	*hevcasm_get_transform(table, 1, 2) = f[0];
	for (int log2 = 2; log2 <= 5; ++log2)
		*hevcasm_get_transform(table, 0, log2) = f[log2 - 1];
}

typedef struct
{
	const int16_t *coefficients;
	const uint8_t *predicted;
	hevcasm_inverse_transform_add *f;
	int log2TrafoSize, trType;
	uint8_t dst[32 * 32];
} bound_inverse;

static int get_inverse(void *p, hevcasm_instruction_set mask)
{
	bound_inverse *s = (bound_inverse *)p;
	hevcasm_table_inverse_transform_add table;
	hevcasm_populate_inverse_transform_add(&table, mask, 1);
	s->f = *hevcasm_get_inverse_transform_add(&table, s->trType, s->log2TrafoSize);
	if (s->f && mask == HEVCASM_C_REF)
		printf("\t%s %dx%d : ", s->trType ? "sine" : "cosine", 1 << s->log2TrafoSize, 1 << s->log2TrafoSize);
	return s->f != 0;
}

static void invoke_inverse(void *p, int n)
{
	bound_inverse *s = (bound_inverse *)p;
	const ptrdiff_t stride = (ptrdiff_t)1 << s->log2TrafoSize;
	while (n--) s->f(s->dst, stride, s->predicted, stride, s->coefficients);
}

static int mismatch_inverse(void *a, void *b)
{
	const bound_inverse *r = (const bound_inverse *)a, *t = (const bound_inverse *)b;
	return memcmp(r->dst, t->dst, (size_t)1 << (2 * r->log2TrafoSize)) != 0;
}

void hevcasm_test_inverse_transform_add(int *error_count, hevcasm_instruction_set mask)
{
	printf("\ninverse_transform_add - Inverse Transform, then add to predicted\n");

	static HEVCASM_ALIGN(32, int16_t, coefficients[32 * 32]);
	static HEVCASM_ALIGN(32, uint8_t, predicted[32 * 32]);
	for (int i = 0; i < 32 * 32; ++i) coefficients[i] = (int16_t)((((unsigned)rand() << 1) ^ (unsigned)rand()) & 0xffff);
	for (int i = 0; i < 32 * 32; ++i) predicted[i] = (uint8_t)rand();

	bound_inverse b[2];
	memset(b, 0, sizeof(b));
	b[0].coefficients = coefficients;
	b[0].predicted = predicted;
	for (int j = 1; j < 6; ++j)
	{
		b[0].trType = j == 1;
		b[0].log2TrafoSize = j == 1 ? 2 : j;
		b[1] = b[0];
		*error_count += hevcasm_test(&b[0], &b[1], get_inverse, invoke_inverse, mismatch_inverse, mask, 100000);
	}
}

typedef struct
{
	hevcasm_transform *f;
	HEVCASM_ALIGN(32, int16_t, dst[32 * 32]);
	const int16_t *src;
	ptrdiff_t src_stride;
	int trType, log2TrafoSize;
} bound_transform;

static int get_transform(void *p, hevcasm_instruction_set mask)
{
	bound_transform *s = (bound_transform *)p;
	hevcasm_table_transform table;
	hevcasm_populate_transform(&table, mask);
	s->f = *hevcasm_get_transform(&table, s->trType, s->log2TrafoSize);
	if (s->f && mask == HEVCASM_C_REF)
		printf("\t%s %dx%d : ", s->trType ? "sine" : "cosine", 1 << s->log2TrafoSize, 1 << s->log2TrafoSize);
	return s->f != 0;
}

static void invoke_transform(void *p, int n)
{
	bound_transform *s = (bound_transform *)p;
	while (n--) s->f(s->dst, s->src, s->src_stride);
}

static int mismatch_transform(void *a, void *b)
{
	const bound_transform *r = (const bound_transform *)a, *t = (const bound_transform *)b;
	return memcmp(r->dst, t->dst, sizeof(int16_t) << (2 * r->log2TrafoSize)) != 0;
}

void hevcasm_test_transform(int *error_count, hevcasm_instruction_set mask)
{
	printf("\nhevcasm_transform - Forward Transform\n");

	static HEVCASM_ALIGN(32, int16_t, src[32 * 32]);
	for (int i = 0; i < 32 * 32; ++i) src[i] = (int16_t)((rand() & 0x1ff) - 0x100);

	static bound_transform b[2];
	b[0].src = src;
	b[0].src_stride = 32;
	for (int j = 1; j < 6; ++j)
	{
		b[0].trType = j == 1;
		b[0].log2TrafoSize = j == 1 ? 2 : j;
		b[1] = b[0];
		*error_count += hevcasm_test(&b[0], &b[1], get_transform, invoke_transform, mismatch_transform, mask, 100000);
	}
}
