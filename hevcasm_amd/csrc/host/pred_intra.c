/*
 * pred_intra.c -- selector and self-test for intra prediction (reference pred_intra.c:113-139
 * and :143-227).  hevcasm_populate_pred_intra() fills the table as the reference does: INTRA_DC only, every other slot NULL (the reference has
 * the other 34 modes only as f265 assembly for 4x4 / 8x8, never wired) -- so a non-NULL slot still means "pinned to the reference's C".
 * The other modes -- planar and angular from H.265 8.4.4.2.4 / .6 (cpath.c; the GPU kernel behind the HIP bit), the one part of this library
 * with no reference C to pin it -- are an explicit opt-in: hevcasm_populate_pred_intra_all_modes() (round 6; until
 * round 5 the plain populate filled them, which made the table differ from the reference's).  The self-test prints the reference's rows only (DC).
 */
#include "pred_intra.h"
#include "cpath.h"
#include "perblock.h"
#include "hevcasm_test.h"

#include <stdlib.h>
#include <string.h>

static void populate(hevcasm_table_pred_intra *table, hevcasm_instruction_set mask, int all_modes)
{
	mask = (hevcasm_instruction_set)hevcasm_hip1_effective_mask(mask);   /* HEVCASM_HIP only while the device is usable and unlatched */
	memset(table, 0, sizeof(*table));
	hevcasm_pred_intra *dc = 0;
	if (mask & (HEVCASM_C_REF | HEVCASM_C_OPT)) dc = hevcasm_c_pred_intra_dc;
	if (mask & HEVCASM_HIP) dc = hevcasm_hip1_pred_intra_dc;
	hevcasm_pred_intra *planar = 0, *angular = 0;
	if (all_modes && (mask & (HEVCASM_C_REF | HEVCASM_C_OPT))) { planar = hevcasm_c_pred_intra_planar; angular = hevcasm_c_pred_intra_angular; }
	if (all_modes && (mask & HEVCASM_HIP)) planar = angular = hevcasm_hip1_pred_intra_any;
	for (int log2 = 2; log2 <= 5; ++log2)
		for (int cIdx = 0; cIdx < 2; ++cIdx)
		{
			const hevcasm_pred_intra_packed packed = hevcasm_pred_intra_pack(cIdx, log2);
			*hevcasm_get_pred_intra(table, 0, packed) = planar;
			*hevcasm_get_pred_intra(table, 1, packed) = dc;
			for (int mode = 2; mode < 35; ++mode) *hevcasm_get_pred_intra(table, mode, packed) = angular;
		}
}

/* the reference's table: INTRA_DC, nothing else (pred_intra.c:113-139) */
void hevcasm_populate_pred_intra(hevcasm_table_pred_intra *table, hevcasm_instruction_set mask) { populate(table, mask, 0); }

/* opt-in: all 35 modes; the 34 beyond DC follow the text of H.265, not any reference C (unpinned) */
void hevcasm_populate_pred_intra_all_modes(hevcasm_table_pred_intra *table, hevcasm_instruction_set mask) { populate(table, mask, 1); }

typedef struct
{
	hevcasm_pred_intra *f;
	uint8_t dst[32 * 32];
	const uint8_t *neighbours;
	int intraPredMode, packed;
} bound_pred_intra;

static int get_pred_intra(void *p, hevcasm_instruction_set mask)
{
	bound_pred_intra *s = (bound_pred_intra *)p;
	if (s->intraPredMode != 1) return 0;   /* the only named mode: "DC" (reference pred_intra.c:157) */
	hevcasm_table_pred_intra table;
	hevcasm_populate_pred_intra(&table, mask);
	s->f = *hevcasm_get_pred_intra(&table, s->intraPredMode, s->packed);
	if (s->f && mask == HEVCASM_C_REF)
	{
		const int n = 1 << ((s->packed >> 1) & 0x7f);
		printf("\t%dx%d DC%s", n, n, (s->packed & 1) ? " edge" : "");
	}
	memset(s->dst, 0, sizeof(s->dst));
	return s->f != 0;
}

static void invoke_pred_intra(void *p, int n)
{
	bound_pred_intra *s = (bound_pred_intra *)p;
	while (n--) s->f(s->dst, s->neighbours, s->intraPredMode, s->packed);
}

static int mismatch_pred_intra(void *a, void *b)
{
	const bound_pred_intra *x = (const bound_pred_intra *)a, *y = (const bound_pred_intra *)b;
	const int n = 1 << ((x->packed >> 1) & 0x7f);
	return memcmp(x->dst, y->dst, (size_t)n * n);
}

void hevcasm_test_pred_intra(int *error_count, hevcasm_instruction_set mask)
{
	printf("\nhevcasm_pred_intra - Intra Prediction\n");

	static uint8_t samples[256];
	for (int i = 0; i < 256; ++i) samples[i] = (uint8_t)rand();

	bound_pred_intra b[2];
	b[0].neighbours = samples + 128;
	b[0].intraPredMode = 1;
	/* the reference tests the unfiltered (chroma) packing only (pred_intra.c:218); the luma packing
	 * with the edge filter is tested as well */
	for (int cIdx = 1; cIdx >= 0; --cIdx)
		for (int log2 = 2; log2 <= 5; ++log2)
		{
			if (cIdx == 0 && log2 == 5) continue;   /* same function as the chroma row */
			b[0].packed = hevcasm_pred_intra_pack(cIdx, log2);
			b[1] = b[0];
			*error_count += hevcasm_test(&b[0], &b[1], get_pred_intra, invoke_pred_intra, mismatch_pred_intra, mask, 1000);
		}
}
