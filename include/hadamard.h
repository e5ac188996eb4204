/*
 * hadamard.h -- Hadamard SATD (sum of absolute transformed differences) of 2x2, 4x4 and 8x8 blocks:
 * the distortion metric an encoder uses next to SAD in sub-pel motion search and mode decision.
 *
 * API-compatible with the reference's src/lib/hadamard.h: typedef (:53), table satd[3] indexed
 * by log2 size - 1 (:55-59), getter (:61-64), populate / self-test (:66-68).
 * Result = (sum |H (A - B) H'| + n/4) / (n/2), integer division (reference hadamard.c:98-110).
 * Batched GPU entry point: hevcasm_hip.h.
 */
#ifndef INCLUDED_hadamard_h
#define INCLUDED_hadamard_h

#include "hevcasm.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef int hevcasm_hadamard_satd(const uint8_t *srcA, ptrdiff_t stride_srcA, const uint8_t *srcB, ptrdiff_t stride_srcB);

typedef struct
{
	hevcasm_hadamard_satd *satd[3];
}
hevcasm_table_hadamard_satd;

static inline hevcasm_hadamard_satd **hevcasm_get_hadamard_satd(hevcasm_table_hadamard_satd *table, int log2TrafoSize)
{
	return &table->satd[log2TrafoSize - 1];
}

void HEVCASM_API hevcasm_populate_hadamard_satd(hevcasm_table_hadamard_satd *table, hevcasm_instruction_set mask);

void HEVCASM_API hevcasm_test_hadamard_satd(int *error_count, hevcasm_instruction_set mask);

#ifdef __cplusplus
}
#endif

#endif
