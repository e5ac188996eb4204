/*
 * cpath.h -- the library's portable C implementations, selected by the HEVCASM_C_REF and
 * HEVCASM_C_OPT mask bits exactly as in the reference (whose C_OPT column is the same
 * function as C_REF for every hot-path family).  They are what hevcasm_test() compares the HIP
 * column against, and what a caller gets when it masks HEVCASM_HIP off.  They are never used
 * behind the HEVCASM_HIP bit or by the hevcasm_hip_* entry points.
 */
#ifndef HEVCASM_CPATH_H
#define HEVCASM_CPATH_H

#include "sad.h"
#include "ssd.h"
#include "pred_inter.h"
#include "residual_decode.h"
#include "quantize.h"
#include "hadamard.h"
#include "pred_intra.h"
#include "diff.h"

hevcasm_sad hevcasm_c_sad;
hevcasm_sad_multiref hevcasm_c_sad_multiref_4;
hevcasm_ssd hevcasm_c_ssd;
hevcasm_hadamard_satd hevcasm_c_satd_2, hevcasm_c_satd_4, hevcasm_c_satd_8;
hevcasm_pred_uni_8to8 hevcasm_c_pred_uni_8tap, hevcasm_c_pred_uni_4tap;
hevcasm_pred_bi_8to8 hevcasm_c_pred_bi_8tap, hevcasm_c_pred_bi_4tap;
hevcasm_inverse_transform_add hevcasm_c_idst_4, hevcasm_c_idct_4, hevcasm_c_idct_8, hevcasm_c_idct_16, hevcasm_c_idct_32;
hevcasm_transform hevcasm_c_dst_4, hevcasm_c_dct_4, hevcasm_c_dct_8, hevcasm_c_dct_16, hevcasm_c_dct_32;
hevcasm_quantize_inverse hevcasm_c_quantize_inverse;
hevcasm_quantize hevcasm_c_quantize;
hevcasm_quantize_reconstruct hevcasm_c_quantize_reconstruct;
hevcasm_pred_intra hevcasm_c_pred_intra_dc, hevcasm_c_pred_intra_planar, hevcasm_c_pred_intra_angular;
hevcasm_ssd_linear hevcasm_c_ssd_linear;

/* HEVC interpolation filter tap k of an n-tap filter at fractional position frac. */
int hevcasm_c_filter_tap(int n, int frac, int k);
/* Entry (k, i) of the n-point HEVC core transform matrix (n = 4, 8, 16, 32). */
int hevcasm_c_transform_coefficient(int n, int k, int i);

#endif
