/*
 * ssd.c -- selector and self-test for block SSD (reference ssd.c:63-86 and :89-162).
 */
#include "ssd.h"
#include "cpath.h"
#include "perblock.h"
#include "quant_x86.h"
#include "hevcasm_test.h"

#include <stdlib.h>

void hevcasm_populate_ssd(hevcasm_table_ssd *table, hevcasm_instruction_set mask)
{
	mask = (hevcasm_instruction_set)hevcasm_hip1_effective_mask(mask);   /* HEVCASM_HIP only while the device is usable and unlatched */
	hevcasm_ssd *f = 0;
	if (mask & (HEVCASM_C_REF | HEVCASM_C_OPT)) f = hevcasm_c_ssd;
	hevcasm_ssd *wide = f;
#if defined(__x86_64__) || defined(__i386__)
	if (mask & HEVCASM_AVX) wide = hevcasm_avx_ssd;      /* reference ssd.c:80-85: 16x16, 32x32, 64x64 only */
#endif
	if (mask & HEVCASM_HIP) f = wide = hevcasm_hip1_ssd;
	for (int log2 = 2; log2 <= 6; ++log2) *hevcasm_get_ssd(table, log2) = log2 >= 4 ? wide : f;
}

typedef struct
{
	uint8_t *srcA, *srcB;
	int log2TrafoSize, ssd;
	hevcasm_ssd *f;
} bound_ssd;

static int get_ssd(void *p, hevcasm_instruction_set mask)
{
	bound_ssd *s = (bound_ssd *)p;
	hevcasm_table_ssd table;
	hevcasm_populate_ssd(&table, mask);
	s->f = *hevcasm_get_ssd(&table, s->log2TrafoSize);
	if (mask == HEVCASM_C_REF) printf("\t%dx%d : ", 1 << s->log2TrafoSize, 1 << s->log2TrafoSize);
	return s->f != 0;
}

static void invoke_ssd(void *p, int n)
{
	bound_ssd *s = (bound_ssd *)p;
	const int size = 1 << s->log2TrafoSize;
	while (n--) s->ssd = s->f(s->srcA, 2 * size, s->srcB, 2 * size, size, size);
}

static int mismatch_ssd(void *a, void *b)
{
	return ((bound_ssd *)a)->ssd != ((bound_ssd *)b)->ssd;
}

void hevcasm_test_ssd(int *error_count, hevcasm_instruction_set mask)
{
	printf("\nhevcasm_ssd - Sum of Square Differences\n");

	/* 64 rows at stride 128 need 64 * 128 bytes; the reference sizes these 64 * 96 (ssd.c:143-144)
	 * and reads 32 rows past them at 64x64 */
	static uint8_t srcA[64 * 128], srcB[64 * 128];
	for (int i = 0; i < 64 * 128; ++i)
	{
		srcA[i] = (uint8_t)rand();
		srcB[i] = (uint8_t)rand();
	}

	bound_ssd b[2];
	b[0].srcA = srcA;
	b[0].srcB = srcB;
	for (b[0].log2TrafoSize = 2; b[0].log2TrafoSize <= 6; ++b[0].log2TrafoSize)
	{
		b[1] = b[0];
		*error_count += hevcasm_test(&b[0], &b[1], get_ssd, invoke_ssd, mismatch_ssd, mask, 100000);
	}
}
