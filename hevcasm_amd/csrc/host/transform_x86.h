/* transform_x86.h -- host SIMD transforms (SSSE3 / AVX2 intrinsics): see transform_x86.c.  Only referenced on x86 builds. */
#ifndef HEVCASM_TRANSFORM_X86_H
#define HEVCASM_TRANSFORM_X86_H

#include "residual_decode.h"

#if defined(__x86_64__) || defined(__i386__)
hevcasm_inverse_transform_add hevcasm_ssse3_idct_8, hevcasm_ssse3_idct_16;
hevcasm_transform hevcasm_ssse3_dct_16;
hevcasm_inverse_transform_add hevcasm_avx2_idst_4, hevcasm_avx2_idct_4, hevcasm_avx2_idct_8, hevcasm_avx2_idct_16, hevcasm_avx2_idct_32;
#endif

#endif
