/* sad_x86.h -- host SIMD SAD kernels (SSE2 / AVX2 intrinsics): see sad_x86.c.  Only referenced on x86 builds. */
#ifndef HEVCASM_SAD_X86_H
#define HEVCASM_SAD_X86_H

#include "sad.h"

int hevcasm_x86_simd_built(void);     /* 1 when the kernels below exist in this build */

#if defined(__x86_64__) || defined(__i386__)
hevcasm_sad hevcasm_sse2_sad;                          /* widths 8, 16, 32, 48, 64 */
hevcasm_sad_multiref hevcasm_sse2_sad_multiref_4;      /* the same widths */
hevcasm_sad hevcasm_avx2_sad;                          /* widths 32, 64 */
hevcasm_sad_multiref hevcasm_avx2_sad_multiref_4;      /* every multiple of 4 up to 64 */
#endif

#endif
