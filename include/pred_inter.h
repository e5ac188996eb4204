/*
 * pred_inter.h -- HEVC inter prediction (non-weighted motion compensation), 8-bit.
 *
 * API-compatible with the reference's src/lib/pred_inter.h: uni-prediction typedef (:56),
 * table p[2][9][2][2] (:58-62) and getter (:64-67); bi-prediction typedef (:76), table
 * p[2][5][2] (:78-82) and getter (:84-88).  `ref` points at the integer-pel top-left sample of
 * the block; the filters read 3 (luma) or 1 (chroma) samples before it and 4 / 2 after it in
 * each filtered direction.  The reference allows its kernels to write to the right of the
 * destination block (:42); the kernels of this library never do.
 * Batched GPU entry points: hevcasm_hip.h.
 */
#ifndef INCLUDED_hevcasm_prediction_inter_h
#define INCLUDED_hevcasm_prediction_inter_h

#include "hevcasm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Uni-directional prediction.  xFrac, yFrac: quarter-sample (luma) or eighth-sample (chroma)
 * phase of the motion vector. */
typedef void HEVCASM_API hevcasm_pred_uni_8to8(uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *ref, ptrdiff_t stride_ref, int nPbW, int nPbH, int xFrac, int yFrac);

/* Indexed [taps/4 - 1][width class][xFrac != 0][yFrac != 0]. */
typedef struct
{
	hevcasm_pred_uni_8to8 *p[2][9][2][2];
}
hevcasm_table_pred_uni_8to8;

static inline hevcasm_pred_uni_8to8 **hevcasm_get_pred_uni_8to8(hevcasm_table_pred_uni_8to8 *table, int taps, int w, int h, int xFrac, int yFrac)
{
	(void)h;
	return &table->p[taps / 4 - 1][(w + taps - 1) / taps][xFrac != 0][yFrac != 0];
}

void HEVCASM_API hevcasm_populate_pred_uni_8to8(hevcasm_table_pred_uni_8to8 *table, hevcasm_instruction_set mask);

hevcasm_test_function hevcasm_test_pred_uni;

/* Bi-directional prediction: the rounded mean of two 14-bit intermediate predictions. */
typedef void hevcasm_pred_bi_8to8(uint8_t *dst0, ptrdiff_t stride_dst, const uint8_t *ref0, const uint8_t *ref1, ptrdiff_t stride_ref, int nPbW, int nPbH, int xFrac0, int yFrac0, int xFrac1, int yFrac1);

/* Indexed [taps/4 - 1][width class][any fraction non-zero]. */
typedef struct
{
	hevcasm_pred_bi_8to8 *p[2][5][2];
}
hevcasm_table_pred_bi_8to8;

static inline hevcasm_pred_bi_8to8 **hevcasm_get_pred_bi_8to8(hevcasm_table_pred_bi_8to8 *table, int taps, int w, int h, int xFracA, int yFracA, int xFracB, int yFracB)
{
	(void)h;
	return &table->p[taps / 4 - 1][(w + 2 * taps - 1) / (2 * taps)][(xFracA | yFracA | xFracB | yFracB) != 0];
}

void HEVCASM_API hevcasm_populate_pred_bi_8to8(hevcasm_table_pred_bi_8to8 *table, hevcasm_instruction_set mask);

hevcasm_test_function hevcasm_test_pred_bi;

#ifdef __cplusplus
}
#endif

#endif
