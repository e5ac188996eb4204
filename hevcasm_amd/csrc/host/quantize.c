/*
 * quantize.c -- selectors and self-tests for inverse quantisation, quantisation, reconstruction.
 *
 * Selection as reference quantize.c:65-84, :189-204, :305-330; self-tests as :87-155
 * (dequantise: src in [-256,-1], scale 51, shift 14), :208-283 (quantise: rand()-rand(), scale
 * 51, shift 20, offset 14; compares !!cbf and the levels) and :333-424 (reconstruct: 9-bit
 * residuals, strides 32).
 */
#include "quantize.h"
#include "cpath.h"
#include "perblock.h"
#include "quant_x86.h"
#include "hevcasm_test.h"

#include <stdlib.h>
#include <string.h>

void hevcasm_populate_quantize_inverse(hevcasm_table_quantize_inverse *table, hevcasm_instruction_set mask)
{
	mask = (hevcasm_instruction_set)hevcasm_hip1_effective_mask(mask);   /* HEVCASM_HIP only while the device is usable and unlatched */
	hevcasm_quantize_inverse *f = 0;
	if (mask & (HEVCASM_C_REF | HEVCASM_C_OPT)) f = hevcasm_c_quantize_inverse;
#if defined(__x86_64__) || defined(__i386__)
	if (mask & HEVCASM_SSE41) f = hevcasm_sse4_quantize_inverse;      /* where the reference selects its assembly (quantize.c:71) */
#endif
	if (mask & HEVCASM_HIP) f = hevcasm_hip1_quantize_inverse;
	*hevcasm_get_quantize_inverse(table) = f;
}

void hevcasm_populate_quantize(hevcasm_table_quantize *table, hevcasm_instruction_set mask)
{
	mask = (hevcasm_instruction_set)hevcasm_hip1_effective_mask(mask);   /* HEVCASM_HIP only while the device is usable and unlatched */
	hevcasm_quantize *f = 0;
	if (mask & (HEVCASM_C_REF | HEVCASM_C_OPT)) f = hevcasm_c_quantize;
#if defined(__x86_64__) || defined(__i386__)
	if (mask & HEVCASM_SSE41) f = hevcasm_sse4_quantize;      /* where the reference selects its assembly (quantize.c:195) */
#endif
	if (mask & HEVCASM_HIP) f = hevcasm_hip1_quantize;
	*hevcasm_get_quantize(table) = f;
}

void hevcasm_populate_quantize_reconstruct(hevcasm_table_quantize_reconstruct *table, hevcasm_instruction_set mask)
{
	mask = (hevcasm_instruction_set)hevcasm_hip1_effective_mask(mask);   /* HEVCASM_HIP only while the device is usable and unlatched */
	hevcasm_quantize_reconstruct *f = 0;
	if (mask & (HEVCASM_C_REF | HEVCASM_C_OPT)) f = hevcasm_c_quantize_reconstruct;
#if defined(__x86_64__) || defined(__i386__)
	if (mask & HEVCASM_SSE41) f = hevcasm_sse4_quantize_reconstruct;      /* where the reference selects its assembly (quantize.c:311-318) */
#endif
	if (mask & HEVCASM_HIP) f = hevcasm_hip1_quantize_reconstruct;
	for (int log2 = 2; log2 <= 5; ++log2) *hevcasm_get_quantize_reconstruct(table, log2) = f;
}

static void print_size(int log2)
{
	printf("\t%dx%d : ", 1 << log2, 1 << log2);
}

typedef struct
{
	const int16_t *src;
	HEVCASM_ALIGN(32, int16_t, dst[32 * 32]);
	hevcasm_quantize_inverse *f;
	int scale, shift, log2TrafoSize;
} bound_dequant;

static int get_dequant(void *p, hevcasm_instruction_set mask)
{
	bound_dequant *s = (bound_dequant *)p;
	hevcasm_table_quantize_inverse table;
	hevcasm_populate_quantize_inverse(&table, mask);
	s->f = *hevcasm_get_quantize_inverse(&table);
	if (s->f && mask == HEVCASM_C_REF) print_size(s->log2TrafoSize);
	return s->f != 0;
}

static void invoke_dequant(void *p, int n)
{
	bound_dequant *s = (bound_dequant *)p;
	while (n--) s->f(s->dst, s->src, s->scale, s->shift, 1 << (2 * s->log2TrafoSize));
}

static int mismatch_dequant(void *a, void *b)
{
	const bound_dequant *r = (const bound_dequant *)a, *t = (const bound_dequant *)b;
	return memcmp(r->dst, t->dst, sizeof(int16_t) << (2 * r->log2TrafoSize)) != 0;
}

void hevcasm_test_quantize_inverse(int *error_count, hevcasm_instruction_set mask)
{
	printf("\nhevcasm_quantize_inverse - Inverse Quantization (\"scaling\")\n");

	static HEVCASM_ALIGN(32, int16_t, src[32 * 32]);
	for (int i = 0; i < 32 * 32; ++i) src[i] = (int16_t)((rand() & 0xff) - 0x100);

	static bound_dequant b[2];
	b[0].src = src;
	b[0].scale = 51;
	b[0].shift = 14;
	for (b[0].log2TrafoSize = 2; b[0].log2TrafoSize <= 5; ++b[0].log2TrafoSize)
	{
		b[1] = b[0];
		*error_count += hevcasm_test(&b[0], &b[1], get_dequant, invoke_dequant, mismatch_dequant, mask, 100000);
	}
}

typedef struct
{
	const int16_t *src;
	HEVCASM_ALIGN(32, int16_t, dst[32 * 32]);
	hevcasm_quantize *f;
	int scale, shift, offset, log2TrafoSize, cbf;
} bound_quant;

static int get_quant(void *p, hevcasm_instruction_set mask)
{
	bound_quant *s = (bound_quant *)p;
	hevcasm_table_quantize table;
	hevcasm_populate_quantize(&table, mask);
	s->f = *hevcasm_get_quantize(&table);
	if (mask == HEVCASM_C_REF) print_size(s->log2TrafoSize);
	return s->f != 0;
}

static void invoke_quant(void *p, int n)
{
	bound_quant *s = (bound_quant *)p;
	while (n--) s->cbf = s->f(s->dst, s->src, s->scale, s->shift, s->offset, 1 << (2 * s->log2TrafoSize));
}

static int mismatch_quant(void *a, void *b)
{
	const bound_quant *r = (const bound_quant *)a, *t = (const bound_quant *)b;
	if (!r->cbf != !t->cbf) return 1;
	return memcmp(r->dst, t->dst, sizeof(int16_t) << (2 * r->log2TrafoSize)) != 0;
}

void hevcasm_test_quantize(int *error_count, hevcasm_instruction_set mask)
{
	printf("\nhevcasm_quantize - Quantization\n");

	static HEVCASM_ALIGN(32, int16_t, src[32 * 32]);
	for (int i = 0; i < 32 * 32; ++i) src[i] = (int16_t)(rand() - rand());

	static bound_quant b[2];
	b[0].src = src;
	b[0].scale = 51;
	b[0].shift = 20;
	b[0].offset = 14;
	for (b[0].log2TrafoSize = 2; b[0].log2TrafoSize <= 5; ++b[0].log2TrafoSize)
	{
		b[1] = b[0];
		*error_count += hevcasm_test(&b[0], &b[1], get_quant, invoke_quant, mismatch_quant, mask, 100000);
	}
}

typedef struct
{
	HEVCASM_ALIGN(32, uint8_t, rec[32 * 32]);
	const uint8_t *pred;
	const int16_t *res;
	int log2TrafoSize;
	hevcasm_quantize_reconstruct *f;
} bound_reconstruct;

static int get_reconstruct(void *p, hevcasm_instruction_set mask)
{
	bound_reconstruct *s = (bound_reconstruct *)p;
	hevcasm_table_quantize_reconstruct table;
	hevcasm_populate_quantize_reconstruct(&table, mask);
	s->f = *hevcasm_get_quantize_reconstruct(&table, s->log2TrafoSize);
	if (mask == HEVCASM_C_REF) print_size(s->log2TrafoSize);
	return s->f != 0;
}

static void invoke_reconstruct(void *p, int n)
{
	bound_reconstruct *s = (bound_reconstruct *)p;
	while (n--) s->f(s->rec, 32, s->pred, 32, s->res, 1 << s->log2TrafoSize);
}

static int mismatch_reconstruct(void *a, void *b)
{
	const bound_reconstruct *r = (const bound_reconstruct *)a, *t = (const bound_reconstruct *)b;
	const int size = 1 << r->log2TrafoSize;
	for (int y = 0; y < size; ++y)
		if (memcmp(&r->rec[y * 32], &t->rec[y * 32], (size_t)size)) return 1;
	return 0;
}

void hevcasm_test_quantize_reconstruct(int *error_count, hevcasm_instruction_set mask)
{
	printf("\nhevcasm_quantize_reconstruct - Reconstruction\n");

	static HEVCASM_ALIGN(32, uint8_t, pred[32 * 32]);
	static HEVCASM_ALIGN(32, int16_t, res[32 * 32]);
	for (int i = 0; i < 32 * 32; ++i)
	{
		pred[i] = (uint8_t)rand();
		res[i] = (int16_t)((rand() & 0x1ff) - 0x100);
	}

	static bound_reconstruct b[2];
	b[0].pred = pred;
	b[0].res = res;
	for (b[0].log2TrafoSize = 2; b[0].log2TrafoSize <= 5; ++b[0].log2TrafoSize)
	{
		b[1] = b[0];
		*error_count += hevcasm_test(&b[0], &b[1], get_reconstruct, invoke_reconstruct, mismatch_reconstruct, mask, 100000);
	}
}
