/* quant_x86.h -- host SIMD (de)quantisation, reconstruction and SSD (SSE4.1 / AVX intrinsics): see quant_x86.c.  x86 builds only. */
#ifndef HEVCASM_QUANT_X86_H
#define HEVCASM_QUANT_X86_H

#include "quantize.h"
#include "ssd.h"

#if defined(__x86_64__) || defined(__i386__)
hevcasm_quantize_inverse hevcasm_sse4_quantize_inverse;
hevcasm_quantize hevcasm_sse4_quantize;
hevcasm_quantize_reconstruct hevcasm_sse4_quantize_reconstruct;
hevcasm_ssd hevcasm_avx_ssd;
#endif

#endif
