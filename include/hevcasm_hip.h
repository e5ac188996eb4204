/*
 * hevcasm_hip.h -- C-ABI of the batched MI355X (gfx950) kernels: the drop-in boundary.
 *
 * The reference library has no process or device boundary: a codec calls one function pointer
 * per block.  One 16x16 SAD moves 516 bytes, far below a kernel launch, so the
 * GPU path is exposed as one `*_batch` entry point per reference function type.  Each takes
 * the reference function's arguments with the per-block pointers replaced by
 *
 *     one base pointer per plane  +  a device array of per-block u32 offsets from that base
 *
 * and processes `n_blocks` independent blocks in one launch.  Everything else (strides, rect
 * packing, fraction and shift semantics, rounding, clipping) is the reference's, and results
 * are bit-identical to the reference's C path (HEVCASM_C_REF).
 *
 * Conventions
 *  - `extern "C"`, plain pointers and sizes only.
 *  - Every data pointer is a DEVICE pointer (hipMalloc or the helpers below); the caller keeps
 *    frames resident in HBM across calls.  Offsets are in bytes for uint8 planes and in
 *    elements for int16 planes; they must address memory inside the plane including the
 *    filter margins, and every byte a block touches must lie less than 4 GiB - 64 above its
 *    plane's base pointer (the kernels address with base + 32-bit offset).  A plane larger than
 *    that is handled by splitting the batch and advancing the base pointer.
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are
 *    asynchronous: they return once the launch is queued.  A batch call does nothing but enqueue
 *    kernels on that stream, so it may be recorded into a HIP graph (stream capture) and replayed.
 *  - Return value: 0 on success; HEVCASM_HIP_EINVAL for arguments outside the domain the
 *    reference function accepts (same conditions as the reference's asserts / NULL table
 *    entries); otherwise the hipError_t of the failing runtime call.  There is no CPU
 *    fallback: without a usable gfx950 device every entry point fails with
 *    HEVCASM_HIP_ENODEV.  hevcasm_hip_last_error() returns text for the calling thread's last
 *    failure.
 *  - Side effect on the HIP runtime's error state: every entry point starts by reading (and thereby CLEARING)
 *    the calling thread's pending non-sticky HIP error (hipGetLastError), so that the status it returns is this
 *    call's own; an application that polls hipGetLastError() for its own launches must do so before calling in.
 *  - Thread safety: entry points may be called concurrently from several host threads (use one
 *    stream per thread to overlap).
 */
#ifndef INCLUDED_hevcasm_hip_h
#define INCLUDED_hevcasm_hip_h

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HEVCASM_HIP_EINVAL (-1)
#define HEVCASM_HIP_ENODEV (-2)

/* ---- runtime ------------------------------------------------------------------------- */

/* Number of visible gfx950 devices (0 when the HIP runtime finds none). */
int hevcasm_hip_device_count(void);
/* 1 when the current device is a usable gfx950, else 0.  Never throws, never falls back. */
int hevcasm_hip_available(void);
/* Select the device used by subsequent calls of this thread (hipSetDevice). */
int hevcasm_hip_set_device(int device);
const char *hevcasm_hip_last_error(void);

/* The per-block table entries behind the HEVCASM_HIP mask bit (hevcasm.h) have the reference's function types, which have no error return
 * (sad.h:50, pred_inter.h:56 ...).  A HIP failure inside one of them -- or asking hevcasm_populate_*() for the bit when no gfx950 device is usable --
 * LATCHES the device "unavailable" for the process: the text of the first failure is kept (hevcasm_hip_latch_reason(), also printed to stderr when
 * no handler is installed) and every later hevcasm_populate_*() / hevcasm_get_ssd_linear() treats HEVCASM_HIP as not set, i.e. fills its slots from
 * the other bits of the mask or with NULL, the reference's convention for "not available" (ssd.c:65-69).  The CALL THAT FAILED cannot deliver a
 * result: it calls the handler installed here with the text and, when the handler returns, returns with its outputs unspecified (0 for the
 * functions that return a value); without a handler it aborts the process, as the reference's asserts do.  There is never a CPU result from a HIP slot.
 * hevcasm_hip_latch_reset() clears the latch (e.g. after the application has reset the device); the batch entry points below are not affected by the
 * latch -- they report every failure through their return value. */
typedef void hevcasm_hip_failure_handler(const char *text);
hevcasm_hip_failure_handler *hevcasm_hip_set_failure_handler(hevcasm_hip_failure_handler *handler);   /* returns the previous handler (NULL = abort) */
int hevcasm_hip_latched(void);
const char *hevcasm_hip_latch_reason(void);
void hevcasm_hip_latch_reset(void);

/* Thin device-memory helpers so that a C codec needs no HIP headers. */
int hevcasm_hip_malloc(void **dev_ptr, size_t bytes);
int hevcasm_hip_free(void *dev_ptr);
int hevcasm_hip_upload(void *dev_dst, const void *host_src, size_t bytes, void *stream);
int hevcasm_hip_download(void *host_dst, const void *dev_src, size_t bytes, void *stream);
int hevcasm_hip_stream_create(void **stream);
int hevcasm_hip_stream_destroy(void *stream);
int hevcasm_hip_stream_synchronize(void *stream);

/* ---- SAD / SSD ------------------------------------------------------------------------ */

/* Batched hevcasm_sad (sad.h; reference sad.h:50, C body sad.c:47-60).
 * sad[i] = SAD(src + src_off[i], ref + ref_off[i]); width, height multiples of 4 in 4..64.
 * Memory touched: the block rows themselves, and for the reference plane at most the 4-byte-aligned words that hold
 * a byte of a block row (the kernels load reference rows word-aligned and realign in registers) -- for 8- and 12-wide
 * blocks (one lane per row) also up to 4 bytes past a row's end, or before the top row's start, that lie between two
 * rows of the same block (rows at least a block width apart, |stride| >= w; closer rows take byte-exact kernels); never
 * anything below the lowest or above the highest word that holds a byte of the block.  Nothing is written but sad[]. */
int hevcasm_hip_sad_batch(
	const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref,
	const uint32_t *src_off, const uint32_t *ref_off, int *sad,
	uint32_t rect, size_t n_blocks, void *stream);

/* Batched hevcasm_sad_multiref, 4 ways (reference sad.h:95, C body sad.c:101-121).
 * ref_off and sad hold 4 consecutive entries per block; all four candidates share `ref`
 * and stride_ref, as the reference's ref[] pointers share stride_ref.
 * Memory touched in the reference plane: whole 4-byte-aligned words, never below the lowest nor above the highest word that
 * holds a byte of any of the block's four candidates (when the candidates lie within 16 pixels of each other the kernel reads
 * their common window once, including words between them that belong to none). */
int hevcasm_hip_sad_multiref_batch(
	const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref,
	const uint32_t *src_off, const uint32_t *ref_off, int *sad,
	uint32_t rect, size_t n_blocks, void *stream);

/* Batched hevcasm_ssd (reference ssd.h:53, C body ssd.c:43-55); w, h multiples of 4 in 4..64
 * (the reference's table holds the square sizes 4..64, its C body takes any w x h).
 * Memory touched: as hevcasm_hip_sad_batch (the same kernel): for BOTH planes whole 4-byte-aligned words between the lowest
 * and the highest word that hold a byte of the block -- with one lane per row (8- and 12-wide blocks) that includes up to
 * 4 bytes past a row's end, or before the top row's start, in the gap between two rows of the block; this needs rows at
 * least a block width apart (|stride| >= w, which the entry point checks: closer rows take byte-exact kernels). */
int hevcasm_hip_ssd_batch(
	const uint8_t *srcA, ptrdiff_t stride_srcA, const uint8_t *srcB, ptrdiff_t stride_srcB,
	const uint32_t *offA, const uint32_t *offB, int *ssd,
	int w, int h, size_t n_blocks, void *stream);

/* ONE launch for a frame's whole partition (round 6): the block costs of PUs of many sizes.  As the prediction `_runs` entry points below: run i = `count`
 * consecutive entries, from `first`, of the descriptor arrays, all of one size `rect` = HEVCASM_RECT(w, h); w one of HEVC's partition widths 4, 8, 12, 16, 24,
 * 32, 48, 64 (anything else: HEVCASM_HIP_EINVAL -- the one-size calls take every multiple of 4), h a multiple of 4 in 4..64.  `runs` is a HOST array, read
 * during the call (it travels in the kernel arguments, 96 runs per launch; more runs = more launches, made by the call).  Per block exactly the result of
 * hevcasm_hip_sad_batch / _sad_multiref_batch / _ssd_batch for the run's size: sad[i] (multiref: sad[4 i .. 4 i + 3], ref_off[4 i .. 4 i + 3]) for descriptor
 * i.  Memory touched: no more than the one-size calls touch. */
typedef struct
{
	uint32_t first, count;      /* descriptor entries [first, first + count) */
	uint32_t rect;              /* HEVCASM_RECT(w, h) */
} hevcasm_hip_cost_run;

int hevcasm_hip_sad_runs(
	const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref,
	const uint32_t *src_off, const uint32_t *ref_off, int *sad, const hevcasm_hip_cost_run *runs, int n_runs, void *stream);
int hevcasm_hip_sad_multiref_runs(
	const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref,
	const uint32_t *src_off, const uint32_t *ref_off, int *sad, const hevcasm_hip_cost_run *runs, int n_runs, void *stream);
int hevcasm_hip_ssd_runs(
	const uint8_t *srcA, ptrdiff_t stride_srcA, const uint8_t *srcB, ptrdiff_t stride_srcB,
	const uint32_t *offA, const uint32_t *offB, int *ssd, const hevcasm_hip_cost_run *runs, int n_runs, void *stream);

/* Batched hevcasm_hadamard_satd (reference hadamard.h:53; C body hadamard.c:76-111):
 * log2TrafoSize 1, 2 or 3 (2x2, 4x4, 8x8 blocks). */
int hevcasm_hip_hadamard_satd_batch(
	const uint8_t *srcA, ptrdiff_t stride_srcA, const uint8_t *srcB, ptrdiff_t stride_srcB,
	const uint32_t *offA, const uint32_t *offB, int *satd,
	int log2TrafoSize, size_t n_blocks, void *stream);

/* Batched hevcasm_ssd_linear (reference diff.h:52; C body diff.c:45-54): run i = `size`
 * contiguous bytes at src0 + off0[i] against src1 + off1[i]; any size >= 0.  The sum is kept in
 * 32 bits like the reference's int (runs longer than 33 024 bytes can wrap). */
int hevcasm_hip_ssd_linear_batch(
	const uint8_t *src0, const uint8_t *src1, const uint32_t *off0, const uint32_t *off1, int *ssd,
	int size, size_t n_blocks, void *stream);

/* Whole-frame sum of squared errors with 64-bit accumulation ("linear SSD / PSNR over whole frames"):
 * frame i is width x height bytes, rows stride_a / stride_b apart, at a + a_off[i] against b + b_off[i];
 * sse[i] = the sum over its rows of hevcasm_ssd_linear(row of a, row of b, width) (reference diff.h:52; C body diff.c:45-54) -- each
 * row sum is exact in the reference's int (width <= 32 768 < 33 025), the sum over rows is kept in 64 bits (a 3840 x 2160 frame of
 * 0 against 255 is 5.4e11).  width 0..32768, height 0..65536, any strides (negative = bottom-up), any alignment.  PSNR is a host
 * one-liner on top: 10 log10(255^2 * width * height / sse).  sse must be 8-byte aligned; the call zeroes it first
 * (hipMemsetAsync on `stream`) and then accumulates into it, so it belongs to the call until `stream` has passed the launch. */
int hevcasm_hip_frame_sse_batch(
	const uint8_t *a, ptrdiff_t stride_a, const uint8_t *b, ptrdiff_t stride_b,
	const uint32_t *a_off, const uint32_t *b_off, uint64_t *sse,
	int width, int height, size_t n_frames, void *stream);

/* ---- intra prediction ----------------------------------------------------------------- */

/* Batched hevcasm_pred_intra (reference pred_intra.h:55; C body pred_intra.c:61-101).  Block i
 * receives neighbours + i * neighbour_stride as its `neighbours` argument (left column ending at
 * [63] -- sample (-1, y) = [63 - y] --, top row starting at [64], corner (-1, -1) at [128]: pred_intra.c:52-58, f265/intra.asm:126-131)
 * and writes the N x N bytes at dst + i * N * N.  packed as hevcasm_pred_intra_pack(): log2 size 2..5, bit 0 = edge filter (luma below 32x32).
 * intraPredMode 1 (INTRA_DC) is the reference's C and bit-identical to it; it reads bytes [64 - N, 64 + N) only, any neighbour_stride is legal
 * and with 2N the records pack back to back.
 * intraPredMode 0 (planar) and 2..34 (angular) have NO C in the reference (only f265 assembly for 4x4 / 8x8 that cannot be assembled here):
 * they follow H.265 8.4.4.2.4 / 8.4.4.2.6 on the same layout, with bit 0 of `packed` also enabling the boundary smoothing of modes 10 and 26
 * (the text's disableIntraBoundaryFilter == 0 && cIdx == 0 && nTbS < 32), and are checked against this repository's restatement of that text
 * only -- parity UNPINNED.  They read bytes [64 - 2N, 64 + 2N) and [128] of a record: records of at least 129 bytes.  The neighbours are used
 * as given: substitution of unavailable samples and the smoothing of 8.4.4.2.3 are the caller's (hevcasm_hip_intra_filter_neighbours_batch). */
int hevcasm_hip_pred_intra_batch(
	uint8_t *dst, const uint8_t *neighbours, ptrdiff_t neighbour_stride,
	int intraPredMode, int packed, size_t n_blocks, void *stream);

/* Batch-API extension of `packed` (not part of the reference's packing; bits 0..15 stay f265's): COMPACT RECORDS.  The reference's layout puts the corner
 * at [128] whatever the block size, so a 4x4 block drags a 129-byte record behind its 17 samples and the batch moves 5x its algorithmic bytes.  With
 * this flag the corner is read at [64 + 2N] -- directly behind the top row -- so a record is the 4N + 1 contiguous bytes [64 - 2N, 64 + 2N] and
 * neighbour_stride may be as small as 4N + 1.  `neighbours` still points 64 - 2N bytes BEFORE block 0's first sample (it may point before the allocation:
 * nothing below [64 - 2N] is ever read).  Accepted in `packed` by the two intra prediction entry points (INTRA_DC never reads the corner and accepts it
 * too) and, or-ed into log2TrafoSize, by hevcasm_hip_intra_filter_neighbours_batch (input and output records alike). */
#define HEVCASM_HIP_INTRA_COMPACT 0x10000

/* The same with one mode PER BLOCK (modes[i] in 0..34): what a frame's intra blocks of one size look like to an encoder's mode decision or a
 * decoder's reconstruction wavefront.  Parity as above: mode 1 pinned, the others unpinned.  `modes` is device memory, so the entry point cannot
 * refuse a bad byte as hevcasm_hip_pred_intra_batch refuses a bad intraPredMode: a byte above 34 is treated as mode 34 (the block's pixels are
 * then those of mode 34 -- defined, never an out-of-range access). */
int hevcasm_hip_pred_intra_modes_batch(
	uint8_t *dst, const uint8_t *neighbours, ptrdiff_t neighbour_stride,
	const uint8_t *modes, int packed, size_t n_blocks, void *stream);

/* H.265 8.4.4.2.3 for a batch: filtered[i] = the smoothed copy of record i's 4N + 1 samples, same layout (bytes [64 - 2N, 64 + 2N) and [128] of the
 * output record are written; filtered_stride >= 129): [1 2 1] / 4 along both edges through the corner, end samples kept; with
 * strong_intra_smoothing != 0, 32x32 blocks whose two edges are flat (|corner + end - 2 middle| < 8 both ways) take the bi-linear form instead.
 * Whether a block uses the filtered or the plain record is the caller's decision (the text's filterFlag: mode and size; f265 passes a filter_flag
 * the same way, f265/intra.asm:748-768).  UNPINNED like the modes above.  `filtered` must not overlap `neighbours`; |filtered_stride| below 129
 * (4N + 1 with compact records) is refused with HEVCASM_HIP_EINVAL: the output records would overwrite each other. */
int hevcasm_hip_intra_filter_neighbours_batch(
	uint8_t *filtered, ptrdiff_t filtered_stride, const uint8_t *neighbours, ptrdiff_t neighbour_stride,
	int log2TrafoSize, int strong_intra_smoothing, size_t n_blocks, void *stream);

/* ---- intra frames: on-device reconstruction in wavefront order (second half) ----------------------------------------
 *
 * A whole intra-coded plane, TU after TU, without leaving the device: for every transform unit the kernel builds the neighbour record from the
 * RECONSTRUCTED plane (the reference's layout, pred_intra.c:59-65: left at [63 - y], top at [64 + x]; corner as f265/intra.asm:126-131), substitutes
 * unavailable samples (H.265 8.4.4.2.2), decides on and applies the neighbour smoothing (8.4.4.2.3), predicts (hevcasm_pred_intra, pred_intra.h:55),
 * and then runs the reconstruction loop of an encoder --
 *     residual = src - prediction -> hevcasm_transform (residual_decode.h:82) -> hevcasm_quantize (quantize.h:78)
 *              -> hevcasm_quantize_inverse (quantize.h:57) -> hevcasm_inverse_transform_add (residual_decode.h:54)
 * -- or, with src == NULL, of a decoder (levels in, the last two steps only).  4x4 luma TUs take the DST (trType 1), as intra CUs do.
 * Every arithmetic step is bit-identical to the reference's C function named; the steps the reference does not have (record construction,
 * substitution, the smoothing decision) follow the text.  PINNED: frames whose TUs all use INTRA_DC (the one mode the reference wires,
 * pred_intra.c:113-139) equal the composition of the reference's own C functions (tests/golden: iframe_dc_*).  Other modes:
 * UNPINNED, as hevcasm_hip_pred_intra_modes_batch.
 *
 * Scheduling.  A TU needs the reconstructed samples left of, above and above-right of it, so the frame cannot be one flat batch.  CTUs are handed out
 * in wavefront order -- CTU (cx, cy) after (cx - 1, cy) and (cx + 1, cy - 1), the two-CTU lag of HEVC's wavefront parallel processing -- by ONE
 * persistent launch: workgroups draw tickets from a counter in `workspace`; ticket order is a topological order of the dependencies, so a workgroup
 * only ever waits for CTUs that are already held by running workgroups (no co-residency assumption, no deadlock); a finished CTU publishes its bottom
 * row and right column as self-tagged 8-byte granules in `workspace`, which its neighbours poll.  Inside a CTU the TUs run in list order on one
 * wavefront.  One frame has at most min(rows, cols / 2) CTUs in flight; n_frames independent frames (an all-intra sequence, or the intra pictures of
 * several streams) fill the device: frames are interleaved inside every wavefront step.
 *
 * Geometry.  frame_w, frame_h: multiples of 4 (partial CTUs at the right / bottom edge are fine).  ctu: 16, 32 or 64 (a 4:2:0 chroma plane of a 64x64-CTU
 * stream is a plane with ctu = 32, cIdx = 1).  Frame f's planes start at recon + f * frame_pitch_recon, src + f * frame_pitch_src (bytes).
 * TU list: tus[ctu_first[f * n_ctus + c] .. ctu_first[f * n_ctus + c + 1]) are the TUs of CTU c (raster order) of frame f, in decoding order (z-scan);
 * x, y in plane samples; every TU aligned to its size and inside the picture; the TUs of a frame must TILE it (a sample of a neighbouring CTU counts as
 * available whenever that CTU precedes in the wavefront).  Levels: frame f at levels + f * n_ctus * ctu * ctu; inside, CTU c at c * ctu * ctu, and the
 * TU at (x, y) at 16 * z, z = the z-scan index of its first 4x4 cell in the CTU (bits of (x % ctu) / 4 and (y % ctu) / 4 interleaved, x lowest) -- a TU's
 * N * N coefficients are contiguous, row-major, as the reference's coeffs[N * N] (residual_decode.h:54); how HM keeps a CTU's coefficients.
 * cbf[i] (optional; encoder only) = hevcasm_quantize's return value for TU i of `tus` (the OR of its levels: zero / non-zero is what it means).
 * qp: HOST pointer to 4 parameter sets, indexed log2TrafoSize - 2 (read during the call).  cIdx: 0 = luma (edge filters below 32x32, neighbour
 * smoothing, DST at 4x4), else chroma.  flags: bit 0 = strong_intra_smoothing_enabled_flag.
 * pred_out (optional): the prediction of every TU, same geometry as recon (what an encoder's mode decision wants to look at).
 * workspace: hevcasm_hip_intra_frames_workspace_bytes(...) bytes of device memory, 16-byte aligned, owned by the call until `stream` has passed it; the
 * call enqueues a memset of it, a small set-up launch and the persistent launch (graph-capturable).  Status: word 1 of the workspace is 0 after a good
 * run; hevcasm_hip_intra_frames_status() waits for the stream and reports it (a malformed TU is skipped and flagged; a wait that times out -- only
 * possible on a malformed list or a lost device -- makes every workgroup leave). */
typedef struct
{
	uint16_t x, y;             /* top-left sample of the TU in its plane */
	uint8_t log2TrafoSize;     /* 2..5 */
	uint8_t intraPredMode;     /* 0..34 */
	uint16_t reserved;         /* 0 */
} hevcasm_hip_intra_tu;

typedef struct
{
	int q_scale, q_shift, q_offset;      /* hevcasm_quantize's scale, shift, offset (quantize.c:160-186: 0 <= scale, offset < 0x8000, 16 <= shift <= 27) */
	int dq_scale, dq_shift;              /* hevcasm_quantize_inverse's scale, shift (quantize.c:53-62; scale fits int16, 1 <= shift <= 31) */
} hevcasm_hip_intra_qp;

#define HEVCASM_HIP_INTRA_STRONG_SMOOTHING 1

size_t hevcasm_hip_intra_frames_workspace_bytes(int frame_w, int frame_h, int ctu, size_t n_frames);

int hevcasm_hip_intra_reconstruct_frames(
	uint8_t *recon, ptrdiff_t stride_recon, size_t frame_pitch_recon,
	const uint8_t *src, ptrdiff_t stride_src, size_t frame_pitch_src,
	int16_t *levels, int *cbf, uint8_t *pred_out,
	const hevcasm_hip_intra_tu *tus, const uint32_t *ctu_first, const hevcasm_hip_intra_qp *qp,
	int frame_w, int frame_h, int ctu, int cIdx, int flags, size_t n_frames,
	void *workspace, void *stream);

/* Synchronises `stream` and returns 0 when the last hevcasm_hip_intra_reconstruct_frames on `workspace` ran clean, else HEVCASM_HIP_EINVAL with the reason
 * in hevcasm_hip_last_error() (bit 0 of the status word: a malformed TU descriptor was skipped; bit 1: a dependency wait timed out). */
int hevcasm_hip_intra_frames_status(const void *workspace, void *stream);

/* ---- inter prediction ----------------------------------------------------------------- */

/* Batched hevcasm_pred_uni_8to8 (reference pred_inter.h:56; C bodies pred_inter.c:141-228).
 * taps = 8 (luma, fractions 0..3, nPbW/nPbH <= 64) or 4 (chroma, fractions 0..7, <= 32).
 * frac[i] = xFrac | (yFrac << 4).  ref + ref_off[i] is the integer-pel top-left sample.
 * Memory read per block is the reference's footprint: the filter margins (taps/2 - 1 samples before, taps/2 after) only in a
 * direction whose fraction is non-zero -- nPbW x nPbH bytes for an integer motion vector, (nPbW + taps - 1) x nPbH for
 * a horizontal-only, nPbW x (nPbH + taps - 1) for a vertical-only and the full window for a two-dimensional fraction.
 * 8 taps: exactly those bytes.  4 taps, PUs up to 8 wide or 16 wide: the kernel loads whole 4-byte-aligned words, 16 bytes at a
 * time, never below the lowest nor above the highest word that holds a byte of the footprints of the (at most 64)
 * consecutive PUs of the batch that one wavefront handles -- i.e. it may read bytes between those footprints, and up to 3
 * bytes beside the first / last of them inside their words.  PRECONDITION of that 4-tap path (not checked by the launcher, which
 * sees only offsets): every address between the lowest and the highest footprint byte of 64 consecutive PUs of the batch must be
 * mapped -- true whenever `ref` and all offsets lie in ONE contiguous allocation (a frame plane, a frame stack), which is what the
 * base + u32 offset addressing of this API presumes anyway; a batch gathered from separately mapped regions must be split per region.
 * Writes exactly nPbW x nPbH bytes per block. */
int hevcasm_hip_pred_uni_8to8_batch(
	uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *ref, ptrdiff_t stride_ref,
	const uint32_t *dst_off, const uint32_t *ref_off, const uint8_t *frac,
	int taps, int nPbW, int nPbH, size_t n_blocks, void *stream);

/* Batched hevcasm_pred_bi_8to8 (reference pred_inter.h:76; C body pred_inter.c:490-530).  Memory read: as above, per
 * reference and its own fraction pair (what the reference's x86 kernels read; its C body reads the full window always). */
int hevcasm_hip_pred_bi_8to8_batch(
	uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *ref0, const uint8_t *ref1, ptrdiff_t stride_ref,
	const uint32_t *dst_off, const uint32_t *ref0_off, const uint32_t *ref1_off,
	const uint8_t *frac0, const uint8_t *frac1,
	int taps, int nPbW, int nPbH, size_t n_blocks, void *stream);

/* ONE launch for a frame's whole partition (round 6).  A frame is never one PU size -- HEVC has 24 (pred_inter.c:438-444) on three planes -- and a launch per
 * (size, plane) is latency, not bandwidth, at one frame per launch.  The caller lists the frame as RUNS: run i = `count` consecutive entries, from `first`, of
 * the descriptor arrays (dst_off / ref_off / frac, as above), all of one size nPbW x nPbH, one filter (taps 8 or 4) and one plane set planes[plane]; sort the
 * frame's PUs by (plane, size) and there are at most 72 runs.  Cb and Cr of a 4:2:0 frame are two runs over the SAME entries with different `plane`.
 * `planes` and `runs` are HOST arrays, read during the call (they travel in the kernel arguments, 96 runs per launch; more runs = more launches, made by the
 * call).  Per PU exactly the result of hevcasm_hip_pred_uni_8to8_batch / _bi_ with the run's geometry; memory read: the stream kernel's footprint (exactly
 * the reference's, see above -- the 4-tap word-aligned strip form is not used here).  offsets of a run are relative to ITS plane's dst / ref pointers. */
typedef struct
{
	uint8_t *dst; ptrdiff_t stride_dst;
	const uint8_t *ref0, *ref1; ptrdiff_t stride_ref;      /* uni-prediction reads ref0 only */
} hevcasm_hip_pred_plane;

typedef struct
{
	uint32_t first, count;               /* descriptor entries [first, first + count) */
	uint8_t taps, nPbW, nPbH, plane;     /* 8 | 4; PU size; index into planes[] */
} hevcasm_hip_pred_run;

int hevcasm_hip_pred_uni_8to8_runs(
	const hevcasm_hip_pred_plane *planes, int n_planes, const hevcasm_hip_pred_run *runs, int n_runs,
	const uint32_t *dst_off, const uint32_t *ref_off, const uint8_t *frac, void *stream);
int hevcasm_hip_pred_bi_8to8_runs(
	const hevcasm_hip_pred_plane *planes, int n_planes, const hevcasm_hip_pred_run *runs, int n_runs,
	const uint32_t *dst_off, const uint32_t *ref0_off, const uint32_t *ref1_off, const uint8_t *frac0, const uint8_t *frac1, void *stream);

/* ---- transforms ------------------------------------------------------------------------ */

/* Batched hevcasm_inverse_transform_add (reference residual_decode.h:54; C path
 * residual_decode.c:69-413).  coeffs: n_blocks consecutive N x N int16 blocks (N = 1 <<
 * log2TrafoSize, 2..5); trType 1 = 4x4 DST.  dst may equal pred with the same offsets and stride (in-place
 * reconstruction); partially overlapping dst and pred blocks are not supported. */
int hevcasm_hip_inverse_transform_add_batch(
	uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *pred, ptrdiff_t stride_pred,
	const uint32_t *dst_off, const uint32_t *pred_off, const int16_t *coeffs,
	int trType, int log2TrafoSize, size_t n_blocks, void *stream);

/* Fused hevcasm_quantize_inverse -> hevcasm_inverse_transform_add:
 * levels are dequantised with (scale, shift) exactly as quantize.c:53-62, including the clip
 * to int16, then transformed and added; the dequantised coefficients never touch HBM. */
int hevcasm_hip_dequant_inverse_transform_add_batch(
	uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *pred, ptrdiff_t stride_pred,
	const uint32_t *dst_off, const uint32_t *pred_off, const int16_t *levels,
	int scale, int shift,
	int trType, int log2TrafoSize, size_t n_blocks, void *stream);

/* Batched hevcasm_transform (reference residual_decode.h:82; C path residual_decode.c:592-892).
 * src_off in int16 elements; coeffs: n_blocks consecutive N x N blocks. */
int hevcasm_hip_transform_batch(
	int16_t *coeffs, const int16_t *src, ptrdiff_t src_stride, const uint32_t *src_off,
	int trType, int log2TrafoSize, size_t n_blocks, void *stream);

/* ---- quantisation ----------------------------------------------------------------------- */

/* hevcasm_quantize_inverse over n coefficients (reference quantize.h:57; quantize.c:53-62).
 * Already a flat array operation: a batch of blocks is one call with n = blocks * N * N.
 * scale must fit int16 (as the reference's SIMD path requires), shift in 1..31. */
int hevcasm_hip_quantize_inverse(
	int16_t *dst, const int16_t *src, int scale, int shift, size_t n, void *stream);

/* Batched hevcasm_quantize (reference quantize.h:78; quantize.c:160-186): n_blocks consecutive
 * blocks of n_per_block coefficients (16, 64, 256 or 1024); cbf[i] = OR of block i's levels.
 * Domain: 0 <= scale < 0x8000, 0 <= offset < 0x8000, 16 <= shift <= 27 (the reference asserts the upper bounds,
 * quantize.c:162-168; negative scale / offset are refused with HEVCASM_HIP_EINVAL). */
int hevcasm_hip_quantize_batch(
	int16_t *dst, const int16_t *src, int scale, int shift, int offset,
	int n_per_block, size_t n_blocks, int *cbf, void *stream);

/* Batched hevcasm_quantize_reconstruct (reference quantize.h:99; quantize.c:292-302):
 * res holds n_blocks consecutive nCbS x nCbS int16 blocks (nCbS = 4, 8, 16 or 32).  rec may equal pred with the
 * same offsets and stride (in-place reconstruction). */
int hevcasm_hip_quantize_reconstruct_batch(
	uint8_t *rec, ptrdiff_t stride_rec, const uint8_t *pred, ptrdiff_t stride_pred,
	const uint32_t *rec_off, const uint32_t *pred_off, const int16_t *res,
	int nCbS, size_t n_blocks, void *stream);

/* ONE launch per stage for a frame's transform units (round 6): TUs of the four sizes the reference's tables hold (residual_decode.h:54, :82: 4x4 DST or DCT,
 * 8x8, 16x16, 32x32) listed as RUNS, as the prediction and block-cost `_runs` entry points list PUs.  Run i = `count` consecutive entries, from `first`, of the
 * offset arrays (and of cbf[]), all of one size; its coefficients are `count` consecutive N x N int16 blocks starting at element `coeff_first` of the
 * coefficient array (for TUs that tile a frame, runs laid end to end: frame_w x frame_h elements in all).  scale / shift / offset are the run's (de)quantiser
 * parameters -- HEVC's shifts depend on the TU size -- read by hevcasm_hip_quantize_runs (all three) and hevcasm_hip_dequant_inverse_transform_add_runs (scale,
 * shift), ignored by the other two.  `runs` is a HOST array read during the call (32 runs per launch; more = more launches, made by the call).  Per TU exactly
 * the result of the one-size call (hevcasm_hip_transform_batch, _quantize_batch, _inverse_transform_add_batch, _dequant_inverse_transform_add_batch,
 * _quantize_reconstruct_batch) with the
 * run's size and parameters, which also gives the domain of every field (`plane` must be 0); a bad run refuses the whole call (HEVCASM_HIP_EINVAL) before anything is launched.
 * Offsets are relative to ONE base pointer and stride per call: planes that share a stride (Cb and Cr; or all three of a frame kept at the luma stride) go in one
 * call when they lie in one allocation. */
typedef struct
{
	uint32_t first, count;       /* descriptor entries [first, first + count) */
	uint32_t coeff_first;        /* first int16 element of the run's coefficient / level blocks */
	uint8_t log2TrafoSize;       /* 2..5 */
	uint8_t trType;              /* 1 = the 4x4 DST (log2TrafoSize 2 only) */
	uint16_t plane;              /* hevcasm_hip_transform_quantize_reconstruct_planes: index into planes[]; every other entry point: 0 */
	int32_t scale, shift, offset;
} hevcasm_hip_tu_run;

int hevcasm_hip_transform_runs(
	int16_t *coeffs, const int16_t *src, ptrdiff_t src_stride, const uint32_t *src_off,
	const hevcasm_hip_tu_run *runs, int n_runs, void *stream);
int hevcasm_hip_quantize_runs(
	int16_t *dst, const int16_t *src, int *cbf,
	const hevcasm_hip_tu_run *runs, int n_runs, void *stream);
int hevcasm_hip_inverse_transform_add_runs(
	uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *pred, ptrdiff_t stride_pred,
	const uint32_t *dst_off, const uint32_t *pred_off, const int16_t *coeffs,
	const hevcasm_hip_tu_run *runs, int n_runs, void *stream);
int hevcasm_hip_dequant_inverse_transform_add_runs(
	uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *pred, ptrdiff_t stride_pred,
	const uint32_t *dst_off, const uint32_t *pred_off, const int16_t *levels,
	const hevcasm_hip_tu_run *runs, int n_runs, void *stream);
/* hevcasm_quantize_reconstruct (quantize.h:99) for units of every size in one launch: per unit exactly hevcasm_hip_quantize_reconstruct_batch with nCbS =
 * 1 << log2TrafoSize; the run's residual blocks start at element coeff_first of `res`; trType and the quantiser fields are not read. */
int hevcasm_hip_quantize_reconstruct_runs(
	uint8_t *rec, ptrdiff_t stride_rec, const uint8_t *pred, ptrdiff_t stride_pred,
	const uint32_t *rec_off, const uint32_t *pred_off, const int16_t *res,
	const hevcasm_hip_tu_run *runs, int n_runs, void *stream);

/* An encoder's residual loop for a frame's TUs in ONE launch (round 6): per TU  src - pred  ->  hevcasm_transform  ->  hevcasm_quantize  (levels and cbf out)
 * ->  hevcasm_quantize_inverse  ->  hevcasm_inverse_transform_add  (residual_decode.h:82, quantize.h:78, :57, residual_decode.h:54), exactly the results of
 * hevcasm_hip_transform_runs on the int16 plane src - pred, hevcasm_hip_quantize_runs with runs_quantize, and hevcasm_hip_dequant_inverse_transform_add_runs
 * with runs_dequantize -- without the residual, the coefficients or the dequantised coefficients ever reaching memory (5 bytes per pixel cross HBM instead of
 * 16).  runs_quantize[i] and runs_dequantize[i] describe the same units (first, count, coeff_first, size, type: checked) and carry the quantiser's (scale,
 * shift, offset) and the dequantiser's (scale, shift); levels and cbf are laid out as for hevcasm_hip_quantize_runs.  rec may equal pred with the same offsets
 * and stride (in-place reconstruction); offsets are in bytes of their planes. */
int hevcasm_hip_transform_quantize_reconstruct_runs(
	uint8_t *rec, ptrdiff_t stride_rec, const uint8_t *src, ptrdiff_t stride_src, const uint8_t *pred, ptrdiff_t stride_pred,
	const uint32_t *rec_off, const uint32_t *src_off, const uint32_t *pred_off, int16_t *levels, int *cbf,
	const hevcasm_hip_tu_run *runs_quantize, const hevcasm_hip_tu_run *runs_dequantize, int n_runs, void *stream);

/* The same for the planes of a frame in ONE launch -- luma, Cb and Cr have their own pointers and strides: run i works on planes[runs_quantize[i].plane]; the
 * offset arrays, levels and cbf are shared (a run's `first` / `coeff_first` say where its entries lie), n_planes <= 4.  Per unit exactly the single-plane call. */
typedef struct
{
	uint8_t *rec;
	const uint8_t *src, *pred;
	ptrdiff_t stride_rec, stride_src, stride_pred;
} hevcasm_hip_tu_plane;

int hevcasm_hip_transform_quantize_reconstruct_planes(
	const hevcasm_hip_tu_plane *planes, int n_planes,
	const uint32_t *rec_off, const uint32_t *src_off, const uint32_t *pred_off, int16_t *levels, int *cbf,
	const hevcasm_hip_tu_run *runs_quantize, const hevcasm_hip_tu_run *runs_dequantize, int n_runs, void *stream);

/* The same two calls for TWO planes that share one PU list -- Cb and Cr of a 4:2:0 frame: same geometry, same motion
 * vectors, same fractions -- in one launch: plane a and plane b of the destination and of each reference have the same
 * stride and take the same offsets.  Results are exactly those of two single-plane calls; a frame's chroma launches are
 * short (a 1080p plane of 8x8 PUs is ~10 us), so halving their number is worth having. */
int hevcasm_hip_pred_uni_8to8_batch_2planes(
	uint8_t *dst_a, uint8_t *dst_b, ptrdiff_t stride_dst, const uint8_t *ref_a, const uint8_t *ref_b, ptrdiff_t stride_ref,
	const uint32_t *dst_off, const uint32_t *ref_off, const uint8_t *frac,
	int taps, int nPbW, int nPbH, size_t n_blocks, void *stream);
int hevcasm_hip_pred_bi_8to8_batch_2planes(
	uint8_t *dst_a, uint8_t *dst_b, ptrdiff_t stride_dst,
	const uint8_t *ref0_a, const uint8_t *ref0_b, const uint8_t *ref1_a, const uint8_t *ref1_b, ptrdiff_t stride_ref,
	const uint32_t *dst_off, const uint32_t *ref0_off, const uint32_t *ref1_off,
	const uint8_t *frac0, const uint8_t *frac1,
	int taps, int nPbW, int nPbH, size_t n_blocks, void *stream);

/* ---- full-search motion estimation (composition of hevcasm_sad) -- */

/* For each block: SAD at every displacement (dx, dy) in [-range, range]^2 around
 * ref + ref_off[i]; result[3i..3i+2] = { dx, dy, cost } of the minimum, ties resolved to the
 * first candidate in raster order of (dy, dx).
 * OWNERSHIP OF result (this entry point and the two frame_* ones): the 12 * n bytes of `result` belong to the call from the
 * moment it is issued until `stream` has passed its last launch.  A launch too small to fill the device cuts every block's
 * candidates over several workgroups and then uses the result slots as scratch: they are preset to 0xff bytes
 * (hipMemsetAsync), hold running 64-bit minimum keys, and are rewritten as (dx, dy, cost) by a final decode launch -- so a
 * reader on another stream, or unrelated data aliased inside those bytes, sees or loses intermediate contents, and after a
 * failed call (non-zero return) the slots hold no answers at all.  result must be 4-byte aligned; nothing else is assumed. */
int hevcasm_hip_me_full_search_batch(
	const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref,
	const uint32_t *src_off, const uint32_t *ref_off, int *result,
	uint32_t rect, int range, size_t n_blocks, void *stream);

/* The same search over rows [row0, row1) of a frame's CTU grid, with no descriptor arrays: the unit BASELINE.json
 * config #5 shards across GPUs (contiguous CTU rows per GPU, each with its +-range halo of the
 * reference frame).  The frame is frame_w x frame_h pixels (multiples of 4), cut into ctu x ctu CTUs in raster
 * order; CTUs on the right / bottom edge are as wide / tall as the frame leaves them (2160 rows: 33 rows of 64x64
 * and one of 64x48, the reference's partitions sad.c:231-232).  `cur` and `ref` point at the pixel in column 0 of
 * CTU row row0 -- so a rank passes pointers into whatever slab it holds -- and the reference must be readable
 * `range` pixels beyond the slab on every side (the halo; the caller pads frame edges, HEVC-style).
 * result[3k..3k+2] = { dx, dy, cost } for the k-th CTU of those rows in raster order; per CTU exactly
 * hevcasm_hip_me_full_search_batch's answer (hevcasm_sad, sad.c:47-60, per candidate; raster-first minimum). */
int hevcasm_hip_me_full_search_frame_rows(
	const uint8_t *cur, ptrdiff_t stride_cur, const uint8_t *ref, ptrdiff_t stride_ref,
	int frame_w, int frame_h, int ctu, int range, int row0, int row1, int *result, void *stream);

/* The same for a run of n_ctus CTUs in raster order starting at CTU first_ctu of the frame (0 = top left), which need not
 * begin or end on a row: a split by CTUs balances ranks to one CTU where a split by rows balances them to one row (34 rows
 * over 8 GPUs are 5 or 4 rows each; 2040 CTUs are 255 each).  `cur` and `ref` point at the pixel in column 0 of the CTU row
 * that holds first_ctu; the caller holds every row the run touches, and the halo around them. */
int hevcasm_hip_me_full_search_frame_ctus(
	const uint8_t *cur, ptrdiff_t stride_cur, const uint8_t *ref, ptrdiff_t stride_ref,
	int frame_w, int frame_h, int ctu, int range, int first_ctu, int n_ctus, int *result, void *stream);

/* The same run of CTUs in n_frames slabs of identical geometry -- consecutive frames, one rank's rows of each -- in ONE launch (round 6): slab f's planes lie
 * f * frame_pitch_cur / f * frame_pitch_ref bytes after slab 0's (any layout the caller keeps its slabs in: a stack, or whole frames with the pointers at the rank's
 * first row), its answers are result[3 (f * n_ctus + k) ...].  A frame split over 8 GPUs leaves each launch 255 CTUs -- a quarter of the device's workgroup slots;
 * F frames per launch give every launch F times the work at no change to what a rank holds.  Per CTU and frame exactly hevcasm_hip_me_full_search_frame_ctus. */
int hevcasm_hip_me_full_search_frames_ctus(
	const uint8_t *cur, ptrdiff_t stride_cur, ptrdiff_t frame_pitch_cur, const uint8_t *ref, ptrdiff_t stride_ref, ptrdiff_t frame_pitch_ref, size_t n_frames,
	int frame_w, int frame_h, int ctu, int range, int first_ctu, int n_ctus, int *result, void *stream);

/* The search of a frame's whole partition in ONE launch (round 6): run i = `count` consecutive descriptors from `first`, all of size `rect`
 * (hevcasm_hip_cost_run, above; any multiple of 4 in 4..64 per dimension), one `range` for the call.  `runs` is a HOST array read during the call (96 runs per
 * launch; more = more launches, made by the call).  Per block exactly hevcasm_hip_me_full_search_batch's answer; one workgroup per block (no candidate-row
 * split), so `result` is written once and never used as scratch. */
int hevcasm_hip_me_full_search_runs(
	const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref,
	const uint32_t *src_off, const uint32_t *ref_off, int *result,
	const hevcasm_hip_cost_run *runs, int n_runs, int range, void *stream);

/* ---- fused quarter-pel refinement ------------------------------------- */

/* For each block: the 49 luma quarter-pel displacements (qx, qy) in [-3, 3]^2 around the
 * integer-pel position ref + ref_off[i].  Candidate prediction = hevcasm_pred_uni_8to8 with the
 * 8-tap filter (reference pred_inter.h:56) at integer offset (qx >> 2, qy >> 2) and fraction
 * (qx & 3, qy & 3); cost = hevcasm_sad (reference sad.h:50) of it against src + src_off[i].
 * result[3i..3i+2] = { qx, qy, cost } of the minimum, ties resolved to the first candidate in
 * raster order of (qy, qx).  The candidates are never written to memory.  metric: 0 = SAD;
 * 1 = hevcasm_hadamard_satd (reference hadamard.h:53) summed over the block's 8x8 tiles, or over its 4x4
 * tiles when a dimension is not a multiple of 8.
 * The reference plane must be readable 4 pixels around each block (the filter margins). */
int hevcasm_hip_subpel_refine_batch(
	const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref,
	const uint32_t *src_off, const uint32_t *ref_off, int *result,
	uint32_t rect, int metric, size_t n_blocks, void *stream);

/* The refinement of a frame's whole partition in ONE launch (round 6): runs as for hevcasm_hip_me_full_search_runs; per block exactly
 * hevcasm_hip_subpel_refine_batch's answer for the run's size and `metric`. */
int hevcasm_hip_subpel_refine_runs(
	const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref,
	const uint32_t *src_off, const uint32_t *ref_off, int *result,
	const hevcasm_hip_cost_run *runs, int n_runs, int metric, void *stream);

#ifdef __cplusplus
}
#endif

#endif
