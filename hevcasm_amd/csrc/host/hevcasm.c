/*
 * hevcasm.c -- library entry points: instruction-set discovery and the self-test main.
 *
 * Counterpart of the reference's src/lib/hevcasm.c.  hevcasm_instruction_set_support()
 * (:99-138 there) reports the host's instruction sets, as the reference does.  HEVCASM_HIP is OPT-IN:
 * the reference's convention is that the highest instruction set in the mask overwrites the table
 * entries (sad.c:63-98), and a per-block table entry behind the HIP bit is a kernel launch per block --
 * right for the self-test's HIP column, wrong for an unchanged codec's inner loop.  So the bit is only
 * reported when the application asks for it (HEVCASM_ENABLE_HIP=1 in the environment), and
 * hevcasm_main() adds it itself for the self-test.  The throughput path (hevcasm_hip_*_batch) never
 * depends on the bit.
 * hevcasm_main() (:152-186 there) runs the reference's twelve self-tests in the reference's order
 * (the ten hot-path families, intra DC and Hadamard SATD), then the linear-SSD test, which the
 * reference has (diff.c:113) but does not call from its main.
 */
#include "hevcasm.h"
#include "sad.h"
#include "ssd.h"
#include "pred_inter.h"
#include "residual_decode.h"
#include "quantize.h"
#include "hadamard.h"
#include "pred_intra.h"
#include "diff.h"
#include "hevcasm_hip.h"

#include <stdlib.h>
#include <string.h>

#if defined(__x86_64__) || defined(__i386__)
#include <cpuid.h>

static uint64_t read_xcr0(void)
{
	uint32_t lo, hi;
	__asm__ __volatile__("xgetbv" : "=a"(lo), "=d"(hi) : "c"(0));
	return ((uint64_t)hi << 32) | lo;
}

static int x86_mask(void)
{
	unsigned a, b, c, d;
	int mask = 0;
	if (!__get_cpuid(1, &a, &b, &c, &d)) return 0;
	if (d & (1u << 26)) mask |= HEVCASM_SSE2;
	if (c & (1u << 0)) mask |= HEVCASM_SSE3;
	if (c & (1u << 9)) mask |= HEVCASM_SSSE3;
	if (c & (1u << 19)) mask |= HEVCASM_SSE41;
	if (c & (1u << 20)) mask |= HEVCASM_SSE42;
	/* AVX needs the CPU flag, OSXSAVE, and the OS saving XMM+YMM state */
	if ((c & (1u << 28)) && (c & (1u << 27)) && (read_xcr0() & 6) == 6)
	{
		mask |= HEVCASM_AVX;
		if (__get_cpuid_count(7, 0, &a, &b, &c, &d) && (b & (1u << 5))) mask |= HEVCASM_AVX2;
	}
	return mask;
}
#else
static int x86_mask(void) { return 0; }
#endif

static int env_set(const char *name)
{
	const char *v = getenv(name);
	return v && *v && strcmp(v, "0");
}

hevcasm_instruction_set hevcasm_instruction_set_support(void)
{
	int mask = HEVCASM_C_REF | HEVCASM_C_OPT | x86_mask();
	if (env_set("HEVCASM_ENABLE_HIP") && !env_set("HEVCASM_DISABLE_HIP") && hevcasm_hip_available() && !hevcasm_hip_latched()) mask |= HEVCASM_HIP;
	return (hevcasm_instruction_set)mask;
}

hevcasm_instruction_set hevcasm_instruction_set_hip(void)
{
	return (hevcasm_instruction_set)((!env_set("HEVCASM_DISABLE_HIP") && hevcasm_hip_available() && !hevcasm_hip_latched()) ? HEVCASM_HIP : 0);
}

void hevcasm_print_instruction_set_support(FILE *f, hevcasm_instruction_set mask)
{
	fprintf(f, "HEVCasm processor instruction set support:\n");
#define X(value, name, description) fprintf(f, "[%c] " #name " (" description ")\n", (mask & (1 << value)) ? 'x' : ' ');
	HEVCASM_INSTRUCTION_SET_XMACRO
#undef X
	fprintf(f, "\n");
}

int hevcasm_main(int argc, const char *argv[])
{
	(void)argc; (void)argv;
	/* the self-test wants the HIP column whenever a device is there (HEVCASM_DISABLE_HIP=1 leaves it out) */
	const hevcasm_instruction_set mask = (hevcasm_instruction_set)(hevcasm_instruction_set_support() | hevcasm_instruction_set_hip());

	printf("HEVCasm self test\n\n");
	hevcasm_print_instruction_set_support(stdout, mask);

	int error_count = 0;
	hevcasm_test_sad_multiref(&error_count, mask);
	hevcasm_test_sad(&error_count, mask);
	hevcasm_test_ssd(&error_count, mask);
	hevcasm_test_pred_intra(&error_count, mask);
	hevcasm_test_hadamard_satd(&error_count, mask);
	hevcasm_test_quantize_inverse(&error_count, mask);
	hevcasm_test_quantize(&error_count, mask);
	hevcasm_test_quantize_reconstruct(&error_count, mask);
	hevcasm_test_pred_uni(&error_count, mask);
	hevcasm_test_pred_bi(&error_count, mask);
	hevcasm_test_inverse_transform_add(&error_count, mask);
	hevcasm_test_transform(&error_count, mask);
	hevcasm_test_ssd_linear(&error_count, mask);   /* not in the reference's main (hevcasm.c:168-180); its test exists (diff.c:113) */

	printf("\nHEVCasm self test: %d errors\n", error_count);
	return error_count;
}
