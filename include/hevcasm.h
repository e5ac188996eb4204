/*
 * hevcasm.h -- core definitions of the MI355X-native HEVCasm replacement.
 *
 * Drop-in for the reference's src/lib/hevcasm.h (reference lines cited per item): same
 * instruction-set bit flags, same HEVCASM_RECT packing, same three entry points, so a codec
 * that includes "hevcasm.h" and links the reference library can link libhevcasm_hip.so
 * instead.  One bit is added to the instruction-set mask: HEVCASM_HIP (bit 9) = "batched
 * HIP kernels for gfx950 are available".
 */
#ifndef INCLUDED_hevcasm_h
#define INCLUDED_hevcasm_h

#include <stddef.h>
#include <stdint.h>
#include <inttypes.h>
#include <stdio.h>

#if defined(__x86_64__)
#define HEVCASM_X64
#endif

/* Symbols are exported with default visibility; the reference's Windows DLL decoration
 * (reference hevcasm.h:59-69) has no counterpart on the ROCm platforms this library targets. */
#define HEVCASM_API

/* Declare `T v` aligned to n bytes (reference hevcasm.h:71-72, 104-105). */
#define HEVCASM_ALIGN(n, T, v) T v __attribute__((aligned(n)))

/* Cycle-resolution timestamp for the self-test harness (reference hevcasm.h:86-100 reads the
 * x86 TSC).  Hosts without a TSC fall back to CLOCK_MONOTONIC nanoseconds. */
typedef uint64_t hevcasm_timestamp;
#if defined(__x86_64__) || defined(__i386__)
static inline hevcasm_timestamp hevcasm_get_timestamp(void)
{
	uint32_t lo, hi;
	__asm__ __volatile__("rdtsc" : "=a"(lo), "=d"(hi));
	return ((hevcasm_timestamp)hi << 32) | lo;
}
#else
#include <time.h>
static inline hevcasm_timestamp hevcasm_get_timestamp(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (hevcasm_timestamp)ts.tv_sec * 1000000000ull + (hevcasm_timestamp)ts.tv_nsec;
}
#endif

/*
 * Instruction sets, as X(bit, NAME, description).  Bits 0-8 are the reference's
 * (hevcasm.h:113-122) and keep their values so existing masks stay valid.  The x86 SIMD bits
 * are reported by hevcasm_instruction_set_support() for information but select nothing in
 * this library: the accelerated path is bit 9.
 */
#define HEVCASM_INSTRUCTION_SET_XMACRO \
	X(0, C_REF, "C - reference; may be slow") \
	X(1, C_OPT, "C - optimised") \
	X(2, SSE2, "SSE2") \
	X(3, SSE3, "SSE3") \
	X(4, SSSE3, "Supplementary SSE3") \
	X(5, SSE41, "SSE4.1") \
	X(6, SSE42, "SSE4.2") \
	X(7, AVX, "AVX") \
	X(8, AVX2, "AVX2") \
	X(9, HIP, "HIP - AMD Instinct MI355X (gfx950) batched kernels")

#define HEVCASM_INSTRUCTION_SET_COUNT 10

#ifdef __cplusplus
extern "C" {
#endif

typedef enum
{
#define X(value, name, description) HEVCASM_##name = 1 << value,
	HEVCASM_INSTRUCTION_SET_XMACRO
#undef X
} hevcasm_instruction_set;

/* Bitmask of usable instruction sets: C_REF | C_OPT always and the x86 bits from cpuid
 * (reference hevcasm.c:99-138) -- what the reference returns, so an unchanged codec that populates its
 * tables from this mask gets host code, as before.  HEVCASM_HIP is opt-in: it is included only when
 * HEVCASM_ENABLE_HIP=1 is set in the environment (and a gfx950 device is visible, and
 * HEVCASM_DISABLE_HIP is not set), because the highest bit in a mask overwrites the table entries
 * (reference sad.c:63-98) and a per-block entry behind HEVCASM_HIP costs a kernel launch per block. */
hevcasm_instruction_set HEVCASM_API hevcasm_instruction_set_support(void);

/* HEVCASM_HIP if a gfx950 device is visible to the HIP runtime (and HEVCASM_DISABLE_HIP is not set),
 * else 0: for callers that want the per-block HIP table entries on purpose -- the self-test does,
 * hevcasm_main() ORs this into its mask -- `hevcasm_instruction_set_support() | hevcasm_instruction_set_hip()`.
 * (Not in the reference.) */
hevcasm_instruction_set HEVCASM_API hevcasm_instruction_set_hip(void);

/* One "[x] NAME (description)" line per instruction set (reference hevcasm.c:141-149).
 * Unlike the reference, output really goes to `f`. */
void HEVCASM_API hevcasm_print_instruction_set_support(FILE *f, hevcasm_instruction_set mask);

/* Library self-test: runs every hevcasm_test_* in the reference's order (hevcasm.c:169-180)
 * and returns the number of mismatches. */
int HEVCASM_API hevcasm_main(int argc, const char *argv[]);

/* Block dimensions packed into one word: width in bits 8.., height in bits 0-7
 * (reference hevcasm.h:156). */
#define HEVCASM_RECT(width, height) (((width) << 8) | (height))

/* Signature shared by all hevcasm_test_* self-tests (reference hevcasm.h:159). */
typedef void HEVCASM_API hevcasm_test_function(int *error_count, hevcasm_instruction_set mask);

#ifdef __cplusplus
}
#endif

#endif
