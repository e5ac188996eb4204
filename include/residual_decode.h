/*
 * residual_decode.h -- HEVC inverse transform + add to prediction, and forward transform.
 *
 * API-compatible with the reference's src/lib/residual_decode.h: inverse typedef (:54), table
 * {dst, dct[4]} (:56-61), getter (:63-74), populate with the extra `encoder` argument (:76);
 * forward typedef (:82), table (:84-89), getter (:91-102), populate (:104).
 * `coeffs` is an N x N block of int16, contiguous, row-major.  trType 1 selects the 4x4 DST-VII.
 * Batched GPU entry points (including the fused dequantise -> inverse transform -> add):
 * hevcasm_hip.h.
 */
#ifndef INCLUDED_hevcasm_residual_decode_h
#define INCLUDED_hevcasm_residual_decode_h

#include "hevcasm.h"

#include <assert.h>

#ifdef __cplusplus
extern "C" {
#endif

/* dst = clip8(pred + inverse_transform(coeffs)) */
typedef void hevcasm_inverse_transform_add(uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *pred, ptrdiff_t stride_pred, const int16_t *coeffs);

typedef struct
{
	hevcasm_inverse_transform_add *dst;
	hevcasm_inverse_transform_add *dct[4];
}
hevcasm_table_inverse_transform_add;

static inline hevcasm_inverse_transform_add **hevcasm_get_inverse_transform_add(hevcasm_table_inverse_transform_add *table, int trType, int log2TrafoSize)
{
	if (!trType) return &table->dct[log2TrafoSize - 2];
	assert(log2TrafoSize == 2);
	return &table->dst;
}

/* `encoder` is accepted for source compatibility.  The reference uses it to gate an AVX2
 * 32x32 kernel that is inexact on extreme coefficients (residual_decode.c:489-495); every
 * kernel of this library is exact, so the flag changes nothing. */
void HEVCASM_API hevcasm_populate_inverse_transform_add(hevcasm_table_inverse_transform_add *table, hevcasm_instruction_set mask, int encoder);

void HEVCASM_API hevcasm_test_inverse_transform_add(int *error_count, hevcasm_instruction_set mask);

/* coeffs = forward_transform(src); src is an N x N int16 residual with row stride src_stride. */
typedef void hevcasm_transform(int16_t *coeffs, const int16_t *src, ptrdiff_t src_stride);

typedef struct
{
	hevcasm_transform *dst;
	hevcasm_transform *dct[4];
}
hevcasm_table_transform;

static inline hevcasm_transform **hevcasm_get_transform(hevcasm_table_transform *table, int trType, int log2TrafoSize)
{
	if (!trType) return &table->dct[log2TrafoSize - 2];
	assert(log2TrafoSize == 2);
	return &table->dst;
}

void HEVCASM_API hevcasm_populate_transform(hevcasm_table_transform *table, hevcasm_instruction_set mask);

void HEVCASM_API hevcasm_test_transform(int *error_count, hevcasm_instruction_set mask);

#ifdef __cplusplus
}
#endif

#endif
