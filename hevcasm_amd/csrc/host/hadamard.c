/*
 * hadamard.c -- selector and self-test for Hadamard SATD (reference hadamard.c:135-158, :161-237).
 * Self-test as the reference's: sizes 8, 4, 2 on 16x8 random planes with stride 2 * size.
 */
#include "hadamard.h"
#include "cpath.h"
#include "perblock.h"
#include "hevcasm_test.h"

#include <stdlib.h>

void hevcasm_populate_hadamard_satd(hevcasm_table_hadamard_satd *table, hevcasm_instruction_set mask)
{
	mask = (hevcasm_instruction_set)hevcasm_hip1_effective_mask(mask);   /* HEVCASM_HIP only while the device is usable and unlatched */
	static hevcasm_hadamard_satd *const c_path[3] = { hevcasm_c_satd_2, hevcasm_c_satd_4, hevcasm_c_satd_8 };
	static hevcasm_hadamard_satd *const hip_path[3] = { hevcasm_hip1_satd_2, hevcasm_hip1_satd_4, hevcasm_hip1_satd_8 };
	hevcasm_hadamard_satd *const *src = 0;
	if (mask & (HEVCASM_C_REF | HEVCASM_C_OPT)) src = c_path;
	if (mask & HEVCASM_HIP) src = hip_path;
	for (int log2 = 1; log2 <= 3; ++log2) *hevcasm_get_hadamard_satd(table, log2) = src ? src[log2 - 1] : 0;
}

typedef struct
{
	uint8_t *srcA, *srcB;
	int log2TrafoSize, satd;
	hevcasm_hadamard_satd *f;
} bound_satd;

static int get_satd(void *p, hevcasm_instruction_set mask)
{
	bound_satd *s = (bound_satd *)p;
	hevcasm_table_hadamard_satd table;
	hevcasm_populate_hadamard_satd(&table, mask);
	s->f = *hevcasm_get_hadamard_satd(&table, s->log2TrafoSize);
	if (mask == HEVCASM_C_REF) printf("\t%dx%d : ", 1 << s->log2TrafoSize, 1 << s->log2TrafoSize);
	return s->f != 0;
}

static void invoke_satd(void *p, int n)
{
	bound_satd *s = (bound_satd *)p;
	const int size = 1 << s->log2TrafoSize;
	while (n--) s->satd = s->f(s->srcA, 2 * size, s->srcB, 2 * size);
}

static int mismatch_satd(void *a, void *b)
{
	return ((bound_satd *)a)->satd != ((bound_satd *)b)->satd;
}

void hevcasm_test_hadamard_satd(int *error_count, hevcasm_instruction_set mask)
{
	printf("\nhevcasm_hadamard_satd - Hadamard Sum of Absolute Transformed Differences\n");

	static uint8_t srcA[16 * 8], srcB[16 * 8];
	for (int i = 0; i < 16 * 8; ++i)
	{
		srcA[i] = (uint8_t)rand();
		srcB[i] = (uint8_t)rand();
	}

	bound_satd b[2];
	b[0].srcA = srcA;
	b[0].srcB = srcB;
	for (b[0].log2TrafoSize = 3; b[0].log2TrafoSize >= 1; --b[0].log2TrafoSize)
	{
		b[1] = b[0];
		*error_count += hevcasm_test(&b[0], &b[1], get_satd, invoke_satd, mismatch_satd, mask, 100000);
	}
}
