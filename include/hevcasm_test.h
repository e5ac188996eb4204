/*
 * hevcasm_test.h -- differential self-test / timing harness.
 *
 * Mirrors the reference's src/lib/hevcasm_test.h:59-82: a test binds two copies of its
 * arguments ("bound" closures), one to HEVCASM_C_REF and one to each instruction set under
 * test, times the latter and compares their results.
 */
#ifndef INCLUDED_hevcasm_test_h
#define INCLUDED_hevcasm_test_h

#include "hevcasm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Bit index of each instruction set (reference hevcasm_test.h:44-48). */
typedef enum
{
#define X(value, name, description) name = value,
	HEVCASM_INSTRUCTION_SET_XMACRO
#undef X
} hevcasm_instruction_set_idx_t;

/* Name of a single-bit instruction-set value, or NULL (reference hevcasm_test.h:50-56). */
static inline const char *hevcasm_instruction_set_as_text(int set)
{
#define X(value, name, description) if (set == (1 << value)) return #name;
	HEVCASM_INSTRUCTION_SET_XMACRO
#undef X
	return 0;
}

/* invoke: run the bound function n times.  mismatch: non-zero when two bounds disagree.
 * get: bind to the implementation selected by `set`; zero when there is none. */
typedef void hevcasm_bound_invoke(void *bound, int n);
typedef int hevcasm_bound_mismatch(void *boundRef, void *boundTest);
typedef int hevcasm_bound_get(void *bound, hevcasm_instruction_set set);

/* Time boundTest (average timestamp ticks per call, printed as " NAME:ticks(xSpeedup)"), then
 * compare with boundRef; prints "-MISMATCH" and returns 1 on disagreement
 * (reference hevcasm_test.c:40-107). */
int hevcasm_count_average_cycles(
	void *boundRef, void *boundTest,
	hevcasm_bound_invoke *f,
	hevcasm_bound_mismatch *m,
	double *first_result,
	hevcasm_instruction_set set,
	int iterations);

/* Run one differential test over every instruction set in `mask`
 * (reference hevcasm_test.c:110-137).  Returns the number of mismatching sets. */
int hevcasm_test(
	void *ref,
	void *test,
	hevcasm_bound_get *get,
	hevcasm_bound_invoke *invoke,
	hevcasm_bound_mismatch *mismatch,
	hevcasm_instruction_set mask,
	int iterations);

#ifdef __cplusplus
}
#endif

#endif
