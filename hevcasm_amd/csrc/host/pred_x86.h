/* pred_x86.h -- host SIMD inter prediction (SSE2 / SSE4.1 intrinsics): see pred_x86.c.  Only referenced on x86 builds. */
#ifndef HEVCASM_PRED_X86_H
#define HEVCASM_PRED_X86_H

#include "pred_inter.h"

#if defined(__x86_64__) || defined(__i386__)
hevcasm_pred_uni_8to8 hevcasm_sse2_pred_uni_copy;                            /* integer vectors, any width */
hevcasm_pred_bi_8to8 hevcasm_sse2_pred_bi_copy;
hevcasm_pred_uni_8to8 hevcasm_sse4_pred_uni_8tap, hevcasm_sse4_pred_uni_4tap; /* every fraction pair, any width */
hevcasm_pred_bi_8to8 hevcasm_sse4_pred_bi_8tap, hevcasm_sse4_pred_bi_4tap;
#endif

#endif
