/*
 * sad.h -- Sum of Absolute Differences, single reference and 4-way multi-reference.
 *
 * API-compatible with the reference's src/lib/sad.h: the function typedefs (:50, :95), the
 * caller-owned tables whose field order is ABI (:52-67, :97-102), the inline slot getters
 * (:69-87, :104-108) and the populate / self-test entry points (:89-91, :110-112).
 * The batched GPU entry points for this family are in hevcasm_hip.h.
 */
#ifndef INCLUDED_sad_h
#define INCLUDED_sad_h

#include "hevcasm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* SAD of a width x height block; rect = HEVCASM_RECT(width, height).  Returns the sum. */
typedef int hevcasm_sad(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref, uint32_t rect);

/* The block sizes that own a table slot, in slot order (the order is ABI: reference sad.h:54-64). */
#define HEVCASM_SAD_NAMED_SIZES(X) \
	X(64, 64) X(64, 32) X(32, 64) X(32, 32) X(32, 16) X(16, 32) X(16, 16) X(16, 8) X(8, 16) X(8, 8) X(8, 4)

/* One pointer per named size -- fields sad64x64, sad64x32, ... sad8x4 -- then one slot shared by every other size. */
typedef struct
{
#define X(w, h) hevcasm_sad *sad ## w ## x ## h;
	HEVCASM_SAD_NAMED_SIZES(X)
#undef X
	hevcasm_sad *sadGeneric;
}
hevcasm_table_sad;

/* Address of the table slot serving a width x height block. */
static inline hevcasm_sad **hevcasm_get_sad(hevcasm_table_sad *table, int width, int height)
{
#define X(w, h) if (width == w && height == h) return &table->sad ## w ## x ## h;
	HEVCASM_SAD_NAMED_SIZES(X)
#undef X
	return &table->sadGeneric;
}

void HEVCASM_API hevcasm_populate_sad(hevcasm_table_sad *table, hevcasm_instruction_set mask);

hevcasm_test_function hevcasm_test_sad;

/* SAD of one source block against several candidate references that share a stride;
 * sad[i] receives the result for ref[i]. */
typedef void hevcasm_sad_multiref(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref[], ptrdiff_t stride_ref, int sad[], uint32_t rect);

/* One slot per (width, height) in multiples of 4 up to 64. */
typedef struct
{
	hevcasm_sad_multiref *lookup[16][16];
	hevcasm_sad_multiref *sadGeneric_4;
}
hevcasm_table_sad_multiref;

/* Only 4-way tables exist: returns NULL for any other `ways`. */
static inline hevcasm_sad_multiref **hevcasm_get_sad_multiref(hevcasm_table_sad_multiref *table, int ways, int width, int height)
{
	return ways == 4 ? &table->lookup[(width >> 2) - 1][(height >> 2) - 1] : 0;
}

void HEVCASM_API hevcasm_populate_sad_multiref(hevcasm_table_sad_multiref *table, hevcasm_instruction_set mask);

hevcasm_test_function hevcasm_test_sad_multiref;

#ifdef __cplusplus
}
#endif

#endif
