/*
 * perblock.h -- per-block entry points behind the HEVCASM_HIP mask bit.
 *
 * These have the reference's per-block function types so that they can sit in the
 * hevcasm_table_* structs.  Each call stages one block through pinned, device-mapped host
 * memory, launches the SAME batched kernel as the hevcasm_hip_*_batch API with n_blocks = 1 and
 * waits for it (hip/perblock.hip).  They exist so that hevcasm_test() gains a HIP column that
 * checks the GPU kernels block by block against the C path; they are correct but pay a launch
 * per block -- throughput callers use hevcasm_hip.h.  The reference's function types have no error
 * channel and a silent CPU result would be a lie: a HIP failure latches the device "unavailable" for the
 * process, goes to the application's failure handler (or aborts), and later hevcasm_populate_*() calls
 * drop the HEVCASM_HIP bit (hevcasm_hip1_effective_mask; include/hevcasm_hip.h, hevcasm_hip_latched).
 */
#ifndef HEVCASM_PERBLOCK_H
#define HEVCASM_PERBLOCK_H

#include "sad.h"
#include "ssd.h"
#include "pred_inter.h"
#include "residual_decode.h"
#include "quantize.h"
#include "hadamard.h"
#include "pred_intra.h"
#include "diff.h"

#ifdef __cplusplus
extern "C" {
#endif

hevcasm_sad hevcasm_hip1_sad;
hevcasm_sad_multiref hevcasm_hip1_sad_multiref_4;
hevcasm_ssd hevcasm_hip1_ssd;
hevcasm_hadamard_satd hevcasm_hip1_satd_2, hevcasm_hip1_satd_4, hevcasm_hip1_satd_8;
hevcasm_pred_uni_8to8 hevcasm_hip1_pred_uni_8tap, hevcasm_hip1_pred_uni_4tap;
hevcasm_pred_bi_8to8 hevcasm_hip1_pred_bi_8tap, hevcasm_hip1_pred_bi_4tap;
hevcasm_inverse_transform_add hevcasm_hip1_idst_4, hevcasm_hip1_idct_4, hevcasm_hip1_idct_8, hevcasm_hip1_idct_16, hevcasm_hip1_idct_32;
hevcasm_transform hevcasm_hip1_dst_4, hevcasm_hip1_dct_4, hevcasm_hip1_dct_8, hevcasm_hip1_dct_16, hevcasm_hip1_dct_32;
hevcasm_quantize_inverse hevcasm_hip1_quantize_inverse;
hevcasm_quantize hevcasm_hip1_quantize;
hevcasm_quantize_reconstruct hevcasm_hip1_quantize_reconstruct;
hevcasm_pred_intra hevcasm_hip1_pred_intra_dc, hevcasm_hip1_pred_intra_any;
hevcasm_ssd_linear hevcasm_hip1_ssd_linear;

/* shim.hip: the mask a populate function works with -- HEVCASM_HIP removed when the device is unusable or latched */
int hevcasm_hip1_effective_mask(int mask);
void hevcasm_hip1_latch(const char *reason);
void hevcasm_hip1_failed(const char *text);

#ifdef __cplusplus
}
#endif

#endif
