/*
 * diff.c -- selector and self-test for linear SSD (reference diff.c:57-67 and :72-129).
 */
#include "diff.h"
#include "cpath.h"
#include "perblock.h"
#include "hevcasm_test.h"

#include <stdlib.h>

hevcasm_ssd_linear *hevcasm_get_ssd_linear(int size, hevcasm_instruction_set mask)
{
	mask = (hevcasm_instruction_set)hevcasm_hip1_effective_mask(mask);   /* HEVCASM_HIP only while the device is usable and unlatched */
	(void)size;
	hevcasm_ssd_linear *f = 0;
	if (mask & (HEVCASM_C_REF | HEVCASM_C_OPT)) f = hevcasm_c_ssd_linear;
	if (mask & HEVCASM_HIP) f = hevcasm_hip1_ssd_linear;
	return f;
}

enum { RUN_BYTES = 0x200 };   /* the reference's test length (diff.c:70) */

typedef struct
{
	uint8_t data[2][RUN_BYTES];
	hevcasm_ssd_linear *f;
	int ssd;
} bound_ssd_linear;

static int get_ssd_linear(void *p, hevcasm_instruction_set mask)
{
	bound_ssd_linear *s = (bound_ssd_linear *)p;
	s->f = hevcasm_get_ssd_linear(RUN_BYTES, mask);
	if (mask == HEVCASM_C_REF) printf("\t%d:", RUN_BYTES);
	return s->f != 0;
}

static void invoke_ssd_linear(void *p, int n)
{
	bound_ssd_linear *s = (bound_ssd_linear *)p;
	while (n--) s->ssd = s->f(s->data[0], s->data[1], RUN_BYTES);
}

static int mismatch_ssd_linear(void *a, void *b)
{
	return ((bound_ssd_linear *)a)->ssd != ((bound_ssd_linear *)b)->ssd;
}

void hevcasm_test_ssd_linear(int *error_count, hevcasm_instruction_set mask)
{
	printf("\nhevcasm_ssd_linear - Linear Sum of Square Differences\n");

	static bound_ssd_linear b[2];
	for (int k = 0; k < 2; ++k)
		for (int i = 0; i < RUN_BYTES; ++i) b[0].data[k][i] = (uint8_t)rand();
	b[1] = b[0];
	*error_count += hevcasm_test(&b[0], &b[1], get_ssd_linear, invoke_ssd_linear, mismatch_ssd_linear, mask, 100000);
}
