/*
 * pred_intra.h -- HEVC intra prediction of one N x N block from its neighbouring samples.
 *
 * API-compatible with the reference's src/lib/pred_intra.h: function type (:55, f265's
 * prototype), the packed parameter and its packer (:52, :58-63), the table p[4][35] +
 * p_dc_filter[3] (:65-70), the getter (:72-84), populate / self-test (:86-88).
 * As in the reference, only INTRA_DC (mode 1) has implementations -- plain and, for luma blocks
 * below 32x32, with the first row / column blended into the neighbours (pred_intra.c:61-101);
 * every other table slot is NULL.
 *
 * `neighbours` points into the middle of the sample array: the left column runs upwards and
 * ends at neighbours[63] (sample (-1, y) = neighbours[63 - y]), the top row starts at
 * neighbours[64] (sample (x, -1) = neighbours[64 + x]); pred_intra.c:52-58.  dst is written
 * with stride N.  Batched GPU entry point: hevcasm_hip.h.
 */
#ifndef INCLUDED_hevcasm_prediction_intra_h
#define INCLUDED_hevcasm_prediction_intra_h

#include "hevcasm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* bits 15..8 bit depth, bits 7..1 log2 of the block size, bit 0 edge-filter flag */
typedef int hevcasm_pred_intra_packed;

typedef void HEVCASM_API hevcasm_pred_intra(uint8_t *dst, const uint8_t *neighbours, int intraPredMode, hevcasm_pred_intra_packed packed);

/* cIdx 0 = luma.  The edge filter is on for luma blocks smaller than 32x32. */
static inline hevcasm_pred_intra_packed hevcasm_pred_intra_pack(int cIdx, int log2CbSize)
{
	const int filtered = cIdx == 0 && log2CbSize < 5;
	return (8 << 8) | (log2CbSize << 1) | (filtered ? 1 : 0);
}

typedef struct
{
	hevcasm_pred_intra *p[4][35];          /* [log2CbSize - 2][intraPredMode] */
	hevcasm_pred_intra *p_dc_filter[3];    /* [log2CbSize - 2], DC with the edge filter */
}
hevcasm_table_pred_intra;

static inline hevcasm_pred_intra **hevcasm_get_pred_intra(hevcasm_table_pred_intra *table, int intraPredMode, hevcasm_pred_intra_packed packed)
{
	const int log2CbSize = (packed >> 1) & 0x7f;
	if ((packed & 1) && intraPredMode == 1) return &table->p_dc_filter[log2CbSize - 2];
	return &table->p[log2CbSize - 2][intraPredMode];
}

void HEVCASM_API hevcasm_populate_pred_intra(hevcasm_table_pred_intra *table, hevcasm_instruction_set mask);

/* Not in the reference: the same table with ALL 35 modes filled -- planar (0) and angular (2..34) after H.265 8.4.4.2.4 / 8.4.4.2.6 on the same
 * neighbour layout (the corner sample p[-1][-1] at neighbours[128], f265/intra.asm:126-131), bit 0 of `packed` also switching the boundary
 * smoothing of modes 10 / 26.  The reference has these modes only as un-wired f265 assembly, so nothing pins them to it: an explicit opt-in, so
 * that a non-NULL slot of hevcasm_populate_pred_intra()'s table keeps meaning "the reference's function". */
void HEVCASM_API hevcasm_populate_pred_intra_all_modes(hevcasm_table_pred_intra *table, hevcasm_instruction_set mask);

void HEVCASM_API hevcasm_test_pred_intra(int *error_count, hevcasm_instruction_set mask);

#ifdef __cplusplus
}
#endif

#endif
