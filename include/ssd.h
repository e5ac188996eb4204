/*
 * ssd.h -- Sum of Squared Differences over a block (e.g. for PSNR / RDO distortion).
 *
 * API-compatible with the reference's src/lib/ssd.h: typedef (:53), table (:55-59; the field
 * really is called `satd` there and its name is ABI for code that pokes the struct), getter
 * (:61-64), populate / self-test (:66-68).  Batched GPU entry point: hevcasm_hip.h.
 */
#ifndef INCLUDED_ssd_h
#define INCLUDED_ssd_h

#include "hevcasm.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef int hevcasm_ssd(const uint8_t *srcA, ptrdiff_t stride_srcA, const uint8_t *srcB, ptrdiff_t stride_srcB, int w, int h);

/* One slot per square block size 4, 8, 16, 32, 64. */
typedef struct
{
	hevcasm_ssd *satd[5];
}
hevcasm_table_ssd;

static inline hevcasm_ssd **hevcasm_get_ssd(hevcasm_table_ssd *table, int log2TrafoSize)
{
	return &table->satd[log2TrafoSize - 2];
}

void HEVCASM_API hevcasm_populate_ssd(hevcasm_table_ssd *table, hevcasm_instruction_set mask);

void HEVCASM_API hevcasm_test_ssd(int *error_count, hevcasm_instruction_set mask);

#ifdef __cplusplus
}
#endif

#endif
