/*
 * quantize.h -- HEVC inverse quantisation ("scaling"), simple forward quantisation, and
 * reconstruction (prediction + residual).
 *
 * API-compatible with the reference's src/lib/quantize.h: inverse typedef (:57) / table (:59-63)
 * / getter (:65-68); forward typedef (:78) / table (:80-84) / getter (:86-89); reconstruct
 * typedef (:99) / table p[4] (:101-105) / getter (:107-110).  Flat scale only (no scaling
 * lists), as in the reference.  Batched GPU entry points: hevcasm_hip.h.
 */
#ifndef INCLUDED_quantize_h
#define INCLUDED_quantize_h

#include "hevcasm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- function types ---------------------------------------------------------------------------- */

/* inverse quantisation: dst[i] = clip16((src[i] * scale + (1 << (shift - 1))) >> shift), i < n */
typedef void hevcasm_quantize_inverse(int16_t *dst, const int16_t *src, int scale, int shift, int n);

/* forward quantisation: dst[i] = clip16(sign(src[i]) * ((|src[i]| * scale + (offset << (shift - 16))) >> shift));
 * returns the OR of all dst[i]: zero exactly when the block quantises to all zeros.
 * Requires scale < 0x8000, offset < 0x8000, 16 <= shift <= 27. */
typedef int hevcasm_quantize(int16_t *dst, const int16_t *src, int scale, int shift, int offset, int n);

/* reconstruction: rec = clip8(pred + res) over an n x n block; res is contiguous with row stride n. */
typedef void hevcasm_quantize_reconstruct(uint8_t *rec, ptrdiff_t stride_rec, const uint8_t *pred, ptrdiff_t stride_pred, const int16_t *res, int n);

/* ---- caller-owned tables (layout is ABI) and their slot getters ----------------------------------- */

typedef struct { hevcasm_quantize_inverse *p; } hevcasm_table_quantize_inverse;
typedef struct { hevcasm_quantize *p; } hevcasm_table_quantize;
typedef struct { hevcasm_quantize_reconstruct *p[4]; /* by log2 size - 2: 4x4 ... 32x32 */ } hevcasm_table_quantize_reconstruct;

static inline hevcasm_quantize_inverse **hevcasm_get_quantize_inverse(hevcasm_table_quantize_inverse *table) { return &table->p; }
static inline hevcasm_quantize **hevcasm_get_quantize(hevcasm_table_quantize *table) { return &table->p; }
static inline hevcasm_quantize_reconstruct **hevcasm_get_quantize_reconstruct(hevcasm_table_quantize_reconstruct *table, int log2TrafoSize)
{
	return &table->p[log2TrafoSize - 2];
}

/* ---- selectors and self-tests --------------------------------------------------------------------- */

void HEVCASM_API hevcasm_populate_quantize_inverse(hevcasm_table_quantize_inverse *table, hevcasm_instruction_set mask);
void HEVCASM_API hevcasm_populate_quantize(hevcasm_table_quantize *table, hevcasm_instruction_set mask);
void HEVCASM_API hevcasm_populate_quantize_reconstruct(hevcasm_table_quantize_reconstruct *table, hevcasm_instruction_set mask);

hevcasm_test_function hevcasm_test_quantize_inverse;
hevcasm_test_function hevcasm_test_quantize;
hevcasm_test_function hevcasm_test_quantize_reconstruct;

#ifdef __cplusplus
}
#endif

#endif
