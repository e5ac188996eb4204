/*
 * sad.c -- selectors and self-tests for SAD and 4-way multi-reference SAD.
 *
 * Selection rule (reference sad.c:63-98, :128-180): a table slot gets the best implementation
 * whose instruction-set bit is in `mask`, or NULL when none is: the HIP per-block entry when
 * HEVCASM_HIP is set; else, where the reference has x86 assembly, the same kernel written with
 * intrinsics (sad_x86.c) -- HEVCASM_SSE2: the eleven named sizes, one reference (sad.c:65-78) and four
 * (sad.c:134-147); HEVCASM_AVX2: four references at every width, keyed on the width, except that at
 * width 8 an SSE2 entry is kept when there is one (sad.c:149-159), and one reference at widths 32 and 64;
 * else the C function when C_REF or C_OPT is set.  Self-tests follow reference sad.c:184-340: 23
 * partitions, 128x128 random planes, strides 64, references deliberately misaligned.
 */
#include "sad.h"
#include "cpath.h"
#include "perblock.h"
#include "sad_x86.h"
#include "hevcasm_test.h"

#include <stdlib.h>

static int named_size(int w, int h)
{
	static const unsigned char named[11][2] = { { 64, 64 }, { 64, 32 }, { 32, 64 }, { 32, 32 }, { 32, 16 }, { 16, 32 }, { 16, 16 }, { 16, 8 }, { 8, 16 }, { 8, 8 }, { 8, 4 } };
	for (int i = 0; i < 11; ++i)
		if (named[i][0] == w && named[i][1] == h) return 1;
	return 0;
}

static hevcasm_sad *select_sad(int w, int h, hevcasm_instruction_set mask)
{
	if (mask & HEVCASM_HIP) return hevcasm_hip1_sad;
#if defined(__x86_64__) || defined(__i386__)
	if ((mask & HEVCASM_AVX2) && (w == 64 || w == 32) && named_size(w, h)) return hevcasm_avx2_sad;
	if ((mask & HEVCASM_SSE2) && named_size(w, h)) return hevcasm_sse2_sad;
#endif
	if (mask & (HEVCASM_C_REF | HEVCASM_C_OPT)) return hevcasm_c_sad;
	return 0;
}

void hevcasm_populate_sad(hevcasm_table_sad *table, hevcasm_instruction_set mask)
{
	mask = (hevcasm_instruction_set)hevcasm_hip1_effective_mask(mask);   /* HEVCASM_HIP only while the device is usable and unlatched */
	for (int h = 4; h <= 64; h += 4)
		for (int w = 4; w <= 64; w += 4)
			*hevcasm_get_sad(table, w, h) = select_sad(w, h, mask);
}

static hevcasm_sad_multiref *select_sad_multiref(int w, int h, hevcasm_instruction_set mask)
{
	hevcasm_sad_multiref *f = 0;
	if (mask & HEVCASM_HIP) return hevcasm_hip1_sad_multiref_4;
#if defined(__x86_64__) || defined(__i386__)
	if ((mask & HEVCASM_SSE2) && named_size(w, h)) f = hevcasm_sse2_sad_multiref_4;
	if ((mask & HEVCASM_AVX2) && !(w == 8 && f)) f = hevcasm_avx2_sad_multiref_4;      /* reference sad.c:157: case 8: if (!f) */
#endif
	if (!f && (mask & (HEVCASM_C_REF | HEVCASM_C_OPT))) f = hevcasm_c_sad_multiref_4;
	return f;
}

void hevcasm_populate_sad_multiref(hevcasm_table_sad_multiref *table, hevcasm_instruction_set mask)
{
	mask = (hevcasm_instruction_set)hevcasm_hip1_effective_mask(mask);   /* HEVCASM_HIP only while the device is usable and unlatched */
	for (int h = 4; h <= 64; h += 4)
		for (int w = 4; w <= 64; w += 4)
			*hevcasm_get_sad_multiref(table, 4, w, h) = select_sad_multiref(w, h, mask);
	table->sadGeneric_4 = select_sad_multiref(4, 4, mask & ~HEVCASM_SSE2);
}

/* the partitions of reference sad.c:231-240 */
static const unsigned char partitions[][2] = {
	{ 64, 64 }, { 64, 48 }, { 64, 32 }, { 64, 16 }, { 48, 64 },
	{ 32, 64 }, { 32, 32 }, { 32, 24 }, { 32, 16 }, { 32, 8 }, { 24, 32 },
	{ 16, 64 }, { 16, 32 }, { 16, 16 }, { 16, 12 }, { 16, 8 }, { 16, 4 }, { 12, 16 },
	{ 8, 32 }, { 8, 16 }, { 8, 8 }, { 8, 4 }, { 4, 8 } };
#define N_PARTITIONS ((int)(sizeof(partitions) / sizeof(partitions[0])))

typedef struct
{
	HEVCASM_ALIGN(32, uint8_t, src[128 * 128]);
	HEVCASM_ALIGN(32, uint8_t, ref[128 * 128]);
	hevcasm_sad *f;
	int width, height, sad;
} bound_sad;

static int get_sad(void *p, hevcasm_instruction_set mask)
{
	bound_sad *s = (bound_sad *)p;
	hevcasm_table_sad table;
	hevcasm_populate_sad(&table, mask);
	s->f = *hevcasm_get_sad(&table, s->width, s->height);
	if (mask == HEVCASM_C_REF) printf("\t%dx%d:", s->width, s->height);
	return s->f != 0;
}

static void invoke_sad(void *p, int n)
{
	bound_sad *s = (bound_sad *)p;
	const uint8_t *unaligned_ref = &s->ref[1 + 1 * 128];
	while (n--) s->sad = s->f(s->src, 64, unaligned_ref, 64, HEVCASM_RECT(s->width, s->height));
}

static int mismatch_sad(void *a, void *b)
{
	return ((bound_sad *)a)->sad != ((bound_sad *)b)->sad;
}

void hevcasm_test_sad(int *error_count, hevcasm_instruction_set mask)
{
	printf("\nhevcasm_sad - Sum of Absolute Differences\n");

	static bound_sad b[2];
	for (int i = 0; i < 128 * 128; ++i) b[0].src[i] = (uint8_t)rand();
	for (int i = 0; i < 128 * 128; ++i) b[0].ref[i] = (uint8_t)rand();

	for (int i = 0; i < N_PARTITIONS; ++i)
	{
		b[0].width = partitions[i][0];
		b[0].height = partitions[i][1];
		b[1] = b[0];
		*error_count += hevcasm_test(&b[0], &b[1], get_sad, invoke_sad, mismatch_sad, mask, 100000);
	}
}

typedef struct
{
	hevcasm_sad_multiref *f;
	int ways, width, height;
	HEVCASM_ALIGN(32, uint8_t, src[128 * 128]);
	HEVCASM_ALIGN(32, uint8_t, ref[128 * 128]);
	const uint8_t *ref_array[4];
	int sad[4];
} bound_sad_multiref;

static int get_sad_multiref(void *p, hevcasm_instruction_set mask)
{
	bound_sad_multiref *s = (bound_sad_multiref *)p;
	hevcasm_table_sad_multiref table;
	hevcasm_populate_sad_multiref(&table, mask);
	s->f = *hevcasm_get_sad_multiref(&table, s->ways, s->width, s->height);
	if (s->f && mask == HEVCASM_C_REF) printf("\t%d-way %dx%d : ", s->ways, s->width, s->height);
	return s->f != 0;
}

static void invoke_sad_multiref(void *p, int n)
{
	bound_sad_multiref *s = (bound_sad_multiref *)p;
	while (n--) s->f(s->src, 64, s->ref_array, 64, s->sad, HEVCASM_RECT(s->width, s->height));
}

static int mismatch_sad_multiref(void *a, void *b)
{
	const bound_sad_multiref *r = (const bound_sad_multiref *)a, *t = (const bound_sad_multiref *)b;
	for (int i = 0; i < r->ways; ++i)
		if (r->sad[i] != t->sad[i]) return 1;
	return 0;
}

void hevcasm_test_sad_multiref(int *error_count, hevcasm_instruction_set mask)
{
	static bound_sad_multiref b[2];
	b[0].ways = 4;
	printf("\nhevcasm_sad_multiref - Sum Of Absolute Differences with multiple references (%d candidate references)\n", b[0].ways);

	for (int i = 0; i < 128 * 128; ++i) b[0].src[i] = (uint8_t)rand();
	for (int i = 0; i < 128 * 128; ++i) b[0].ref[i] = (uint8_t)rand();

	/* four candidates around (2, 2), as reference sad.c:328-331; both bounds read b[0].ref */
	b[0].ref_array[0] = &b[0].ref[1 + 2 * 128];
	b[0].ref_array[1] = &b[0].ref[2 + 1 * 128];
	b[0].ref_array[2] = &b[0].ref[3 + 2 * 128];
	b[0].ref_array[3] = &b[0].ref[2 + 3 * 128];

	for (int i = 0; i < N_PARTITIONS; ++i)
	{
		b[0].width = partitions[i][0];
		b[0].height = partitions[i][1];
		b[1] = b[0];
		*error_count += hevcasm_test(&b[0], &b[1], get_sad_multiref, invoke_sad_multiref, mismatch_sad_multiref, mask, 1);
	}
}
