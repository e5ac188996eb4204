/*
 * diff.h -- linear SSD: sum of squared differences of two runs of `size` contiguous bytes (the
 * building block of whole-frame SSD / PSNR).
 *
 * API-compatible with the reference's src/lib/diff.h: function type (:52), the selector
 * hevcasm_get_ssd_linear(size, mask) (:54) -- there is no table struct for this one -- and the
 * self-test (:56).  C body: diff.c:45-54.  Batched GPU entry point: hevcasm_hip.h.
 */
#ifndef INCLUDED_diff_h
#define INCLUDED_diff_h

#include "hevcasm.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef int hevcasm_ssd_linear(const uint8_t *src0, const uint8_t *src1, int size);

/* Best implementation in `mask` for runs of `size` bytes, or NULL. */
hevcasm_ssd_linear *HEVCASM_API hevcasm_get_ssd_linear(int size, hevcasm_instruction_set mask);

void HEVCASM_API hevcasm_test_ssd_linear(int *error_count, hevcasm_instruction_set mask);

#ifdef __cplusplus
}
#endif

#endif
