// quant.hip -- inverse quantisation, forward quantisation and reconstruction for gfx950.
//
// Replaces hevcasm_quantize_inverse (reference quantize.h:57, C body quantize.c:53-62),
// hevcasm_quantize (quantize.h:78, quantize.c:160-186) and hevcasm_quantize_reconstruct
// (quantize.h:99, quantize.c:292-302) for whole batches.
//
// All three are streaming kernels, HBM-bound at 4 bytes per coefficient / 4 N^2 bytes per block:
// every lane moves 16-byte vectors (8 coefficients), consecutive lanes touch consecutive
// addresses, nothing is staged through LDS.  The quantiser's per-block coded-block flag is an OR
// across the lanes that share a block, done with wave shuffles.

#include "common.hpp"
#include "tu_runs.hpp"

namespace hevc {

__device__ __forceinline__ int sext_lo(uint32_t d) { return (int)(short)(d & 0xffff); }
__device__ __forceinline__ int sext_hi(uint32_t d) { return (int)d >> 16; }
__device__ __forceinline__ uint32_t pack16(int lo, int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }

__device__ __forceinline__ int dequant1(int v, int scale, int round, int shift)
{
	return clip3(-32768, 32767, (v * scale + round) >> shift);
}

// quantize.c:53-62, 8 coefficients per lane per trip, grid-stride
__global__ __launch_bounds__(kThreads) void quantize_inverse_kernel(
	int16_t *__restrict__ dst, const int16_t *__restrict__ src, int scale, int shift, size_t n)
{
	const int round = 1 << (shift - 1);
	scale = (scale << 8) >> 8;   // |scale| < 2^15 (entry point): 24-bit operands let the multiply be a full-rate v_mad_i32_i24
	const size_t n8 = n / 8;
	const size_t step = (size_t)gridDim.x * kThreads;
	for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n8; i += step)
	{
		Chunk<16> c;
		c.load(reinterpret_cast<const uint8_t *>(src + 8 * i));
#pragma unroll
		for (int k = 0; k < 4; ++k)
			c.d[k] = pack16(dequant1(sext_lo(c.d[k]), scale, round, shift), dequant1(sext_hi(c.d[k]), scale, round, shift));
		c.store_stream(reinterpret_cast<uint8_t *>(dst + 8 * i));
	}
	// tail (n not a multiple of 8): the reference's C body accepts any n
	const size_t t = 8 * n8 + (size_t)blockIdx.x * kThreads + threadIdx.x;
	if (t < n) dst[t] = (int16_t)dequant1(src[t], scale, round, shift);
}

__device__ __forceinline__ int quant1(int v, int scale, int round, int shift)
{
	// quantize.c:173-180: abs() in int, so -32768 becomes 32768
	const int mag = ((v < 0 ? -v : v) * scale + round) >> shift;
	return clip3(-32768, 32767, v < 0 ? -mag : mag);
}

// One block of NPB coefficients is shared by NPB / 8 lanes (NPB = 16, 64, 256) or by a whole
// wave doing two trips (NPB = 1024).
template <int NPB>
__global__ __launch_bounds__(kThreads) void quantize_kernel(
	int16_t *__restrict__ dst, const int16_t *__restrict__ src, int scale, int shift, int offset,
	int *__restrict__ cbf, uint32_t n_blocks)
{
#define HEVC_WG blockIdx.x
#include "quant_quantize.inc"
#undef HEVC_WG
}

// quantize.c:292-302.  A lane owns PX = min(8, N) horizontally adjacent pixels; 256 / (N * N / PX) blocks per workgroup.
template <int N>
__global__ __launch_bounds__(kThreads) void reconstruct_kernel(
	uint8_t *rec, ptrdiff_t stride_rec, const uint8_t *pred, ptrdiff_t stride_pred,   // rec may alias pred (in-place reconstruction): no __restrict__
	const uint32_t *__restrict__ rec_off, const uint32_t *__restrict__ pred_off,
	const int16_t *__restrict__ res, uint32_t n_blocks)
{
#define HEVC_WG xcd_contiguous(blockIdx.x, gridDim.x)
#include "quant_reconstruct.inc"
#undef HEVC_WG
}

// 4x4 blocks: one lane per block (32 bytes of residual in two 16-byte loads, four row dwords in and
// out); with one lane per ROW the address unit spent a full wave-instruction on 4 bytes per lane.
__global__ __launch_bounds__(kThreads) void reconstruct4_lane_kernel(
	uint8_t *rec, ptrdiff_t stride_rec, const uint8_t *pred, ptrdiff_t stride_pred,   // rec may alias pred (in-place reconstruction): no __restrict__
	const uint32_t *__restrict__ rec_off, const uint32_t *__restrict__ pred_off,
	const int16_t *__restrict__ res, uint32_t n_blocks)
{
#define HEVC_WG xcd_contiguous(blockIdx.x, gridDim.x)
#include "quant_reconstruct4_lane.inc"
#undef HEVC_WG
}

// ---- one launch for the quantiser of a frame's transform units (round 6; tu_runs.hpp): the body above per run, with the run's parameters ----------------
template <int NPB>
__device__ __forceinline__ void quantize_workgroup(
	int16_t *__restrict__ dst, const int16_t *__restrict__ src, int scale, int shift, int offset, int *__restrict__ cbf, uint32_t n_blocks, const uint32_t wg_in_batch)
{
#define HEVC_WG wg_in_batch
#include "quant_quantize.inc"
#undef HEVC_WG
}

// blocks per workgroup by variant (= log2TrafoSize - 2): NPB / 8 lanes share a block, a whole wave above 512 coefficients
constexpr uint32_t kQuantizeRunBlocksPerWg[4] = { kThreads / 2, kThreads / 8, kThreads / 32, kThreads / 64 };

struct QuantizeRunsArgs
{
	int16_t *dst;
	const int16_t *src;
	int *cbf;
	TuRunTable table;
};

__global__ __launch_bounds__(kThreads) void quantize_runs_kernel(const QuantizeRunsArgs a)
{
	uint32_t wg;
	const TuRunDev r = find_tu_run(a.table, wg);
	int16_t *d = a.dst + r.coeff_first;
	const int16_t *s = a.src + r.coeff_first;
	int *cbf = a.cbf + r.first;
	switch (r.variant)
	{
	case 0: quantize_workgroup<16>(d, s, r.scale, r.shift, r.offset, cbf, r.count, wg); break;
	case 1: quantize_workgroup<64>(d, s, r.scale, r.shift, r.offset, cbf, r.count, wg); break;
	case 2: quantize_workgroup<256>(d, s, r.scale, r.shift, r.offset, cbf, r.count, wg); break;
	case 3: quantize_workgroup<1024>(d, s, r.scale, r.shift, r.offset, cbf, r.count, wg); break;
	default: break;
	}
}

// ---- hevcasm_quantize_reconstruct for a frame's units of every size in one launch (round 6): the bodies above per run -----------------------------------
template <int N>
__device__ __forceinline__ void reconstruct_workgroup(
	uint8_t *rec, ptrdiff_t stride_rec, const uint8_t *pred, ptrdiff_t stride_pred, const uint32_t *__restrict__ rec_off, const uint32_t *__restrict__ pred_off,
	const int16_t *__restrict__ res, uint32_t n_blocks, const uint32_t wg_in_batch)
{
#define HEVC_WG wg_in_batch
#include "quant_reconstruct.inc"
#undef HEVC_WG
}
__device__ __forceinline__ void reconstruct4_lane_workgroup(
	uint8_t *rec, ptrdiff_t stride_rec, const uint8_t *pred, ptrdiff_t stride_pred, const uint32_t *__restrict__ rec_off, const uint32_t *__restrict__ pred_off,
	const int16_t *__restrict__ res, uint32_t n_blocks, const uint32_t wg_in_batch)
{
#define HEVC_WG wg_in_batch
#include "quant_reconstruct4_lane.inc"
#undef HEVC_WG
}

// blocks per workgroup by variant (= log2 size - 2): a lane per 4x4 block; above, a lane per 8 pixels of a row
constexpr uint32_t kReconstructRunBlocksPerWg[4] = { kThreads, kThreads / 8, kThreads / 32, kThreads / 128 };

struct ReconstructRunsArgs
{
	uint8_t *rec;
	const uint8_t *pred;
	ptrdiff_t stride_rec, stride_pred;
	const uint32_t *rec_off, *pred_off;
	const int16_t *res;
	TuRunTable table;
};

__global__ __launch_bounds__(kThreads) void reconstruct_runs_kernel(const ReconstructRunsArgs a)
{
	uint32_t wg;
	const TuRunDev r = find_tu_run(a.table, wg);
	const uint32_t *ro = a.rec_off + r.first, *po = a.pred_off + r.first;
	const int16_t *res = a.res + r.coeff_first;
	switch (r.variant)
	{
	case 0: reconstruct4_lane_workgroup(a.rec, a.stride_rec, a.pred, a.stride_pred, ro, po, res, r.count, wg); break;
	case 1: reconstruct_workgroup<8>(a.rec, a.stride_rec, a.pred, a.stride_pred, ro, po, res, r.count, wg); break;
	case 2: reconstruct_workgroup<16>(a.rec, a.stride_rec, a.pred, a.stride_pred, ro, po, res, r.count, wg); break;
	case 3: reconstruct_workgroup<32>(a.rec, a.stride_rec, a.pred, a.stride_pred, ro, po, res, r.count, wg); break;
	default: break;
	}
}

}  // namespace hevc

using namespace hevc;

extern "C" int hevcasm_hip_quantize_reconstruct_runs(uint8_t *rec, ptrdiff_t stride_rec, const uint8_t *pred, ptrdiff_t stride_pred,
		const uint32_t *rec_off, const uint32_t *pred_off, const int16_t *res, const hevcasm_hip_tu_run *runs, int n_runs, void *stream)
{
	if (int e = check_device()) return e;
	if (n_runs < 0) return set_error(HEVCASM_HIP_EINVAL, "negative run count");
	if (n_runs == 0) return 0;
	if (!runs || !rec || !pred || !rec_off || !pred_off || !res) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	std::vector<TuRunTable> tables;
	std::vector<uint32_t> grids;
	if (int e = build_tu_tables(runs, n_runs, [](const hevcasm_hip_tu_run &r, uint32_t &variant, uint32_t &bpwg) {
			if (r.log2TrafoSize < 2 || r.log2TrafoSize > 5 || r.plane) return set_error(HEVCASM_HIP_EINVAL, "run: nCbS must be 4, 8, 16 or 32 (log2TrafoSize 2..5), plane 0");
			variant = (uint32_t)r.log2TrafoSize - 2u;
			bpwg = kReconstructRunBlocksPerWg[variant];
			return 0; }, tables, grids)) return e;
	for (size_t i = 0; i < tables.size(); ++i)
	{
		ReconstructRunsArgs a = { rec, pred, stride_rec, stride_pred, rec_off, pred_off, res, tables[i] };
		hipLaunchKernelGGL(reconstruct_runs_kernel, dim3(grids[i]), dim3(kThreads), 0, as_stream(stream), a);
		if (int e = launch_status("reconstruct_runs_kernel")) return e;
	}
	return 0;
}

extern "C" int hevcasm_hip_quantize_runs(int16_t *dst, const int16_t *src, int *cbf, const hevcasm_hip_tu_run *runs, int n_runs, void *stream)
{
	if (int e = check_device()) return e;
	if (n_runs < 0) return set_error(HEVCASM_HIP_EINVAL, "negative run count");
	if (n_runs == 0) return 0;
	if (!runs || !dst || !src || !cbf) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	std::vector<TuRunTable> tables;
	std::vector<uint32_t> grids;
	if (int e = build_tu_tables(runs, n_runs, [](const hevcasm_hip_tu_run &r, uint32_t &variant, uint32_t &bpwg) {
			if (r.log2TrafoSize < 2 || r.log2TrafoSize > 5 || r.plane) return set_error(HEVCASM_HIP_EINVAL, "run: log2TrafoSize must be 2..5 (16, 64, 256 or 1024 coefficients per block), plane 0");
			// the domain of hevcasm_hip_quantize_batch (quantize.c:162-168)
			if (r.scale < 0 || r.scale >= 0x8000 || r.offset < 0 || r.offset >= 0x8000 || r.shift < 16 || r.shift > 27)
				return set_error(HEVCASM_HIP_EINVAL, "run: scale/offset/shift outside the reference's domain (0 <= scale, offset < 0x8000; 16 <= shift <= 27)");
			variant = (uint32_t)r.log2TrafoSize - 2u;
			bpwg = kQuantizeRunBlocksPerWg[variant];
			return 0; }, tables, grids)) return e;
	for (size_t i = 0; i < tables.size(); ++i)
	{
		QuantizeRunsArgs a = { dst, src, cbf, tables[i] };
		hipLaunchKernelGGL(quantize_runs_kernel, dim3(grids[i]), dim3(kThreads), 0, as_stream(stream), a);
		if (int e = launch_status("quantize_runs_kernel")) return e;
	}
	return 0;
}

extern "C" int hevcasm_hip_quantize_inverse(int16_t *dst, const int16_t *src, int scale, int shift, size_t n, void *stream)
{
	if (int e = check_device()) return e;
	if (shift < 1 || shift > 31) return set_error(HEVCASM_HIP_EINVAL, "shift out of range");
	if (scale < -32768 || scale > 32767) return set_error(HEVCASM_HIP_EINVAL, "scale must fit int16 (the reference's SIMD path multiplies with pmaddwd, quantize_a.asm:53-109)");
	if (n == 0) return 0;
	if (!dst || !src) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	// memory-bound: one 16-byte vector per lane and no loop -- one-shot workgroups stream faster than a capped,
	// grid-striding grid on this part (profiles/r01_hbm_stream.txt: 6.3 against 5.0 TB/s for a plain copy); the
	// kernel's loop only runs when the batch exceeds 2^31 lanes
	size_t wgs = (n / 8 + kThreads - 1) / kThreads;
	if (wgs < 1) wgs = 1;
	if (wgs > 0x7fffffffull / kThreads) wgs = 0x7fffffffull / kThreads;
	hipLaunchKernelGGL(quantize_inverse_kernel, dim3((uint32_t)wgs), dim3(kThreads), 0, as_stream(stream), dst, src, scale, shift, n);
	return launch_status("quantize_inverse_kernel");
}

extern "C" int hevcasm_hip_quantize_batch(int16_t *dst, const int16_t *src, int scale, int shift, int offset,
		int n_per_block, size_t n_blocks, int *cbf, void *stream)
{
	if (int e = check_device()) return e;
	// the reference asserts the upper bounds (quantize.c:162-168); a quantiser's scale and rounding offset are non-negative
	// by construction (the kernel's 24-bit multiply assumes 0 <= scale < 2^15), so negative ones are refused rather than
	// computed differently from the reference's C body
	if (scale < 0 || scale >= 0x8000 || offset < 0 || offset >= 0x8000 || shift < 16 || shift > 27)
		return set_error(HEVCASM_HIP_EINVAL, "scale/offset/shift outside the reference's domain (0 <= scale, offset < 0x8000; 16 <= shift <= 27)");
	if (n_blocks == 0) return 0;
	if (n_blocks > 0xffffffffull) return set_error(HEVCASM_HIP_EINVAL, "n_blocks exceeds 2^32-1");
	if (!dst || !src || !cbf) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	hipStream_t st = as_stream(stream);
#define LAUNCH(NPB) { \
		constexpr int lpb = NPB / 8 < kWave ? NPB / 8 : kWave; \
		const uint32_t grid = ceil_div(n_blocks * lpb, kThreads); \
		hipLaunchKernelGGL((quantize_kernel<NPB>), dim3(grid), dim3(kThreads), 0, st, dst, src, scale, shift, offset, cbf, (uint32_t)n_blocks); \
		return launch_status("quantize_kernel"); }
	switch (n_per_block)
	{
	case 16: LAUNCH(16)
	case 64: LAUNCH(64)
	case 256: LAUNCH(256)
	case 1024: LAUNCH(1024)
	}
#undef LAUNCH
	return set_error(HEVCASM_HIP_EINVAL, "n_per_block must be 16, 64, 256 or 1024");
}

extern "C" int hevcasm_hip_quantize_reconstruct_batch(uint8_t *rec, ptrdiff_t stride_rec, const uint8_t *pred, ptrdiff_t stride_pred,
		const uint32_t *rec_off, const uint32_t *pred_off, const int16_t *res, int nCbS, size_t n_blocks, void *stream)
{
	if (int e = check_device()) return e;
	if (n_blocks == 0) return 0;
	if (n_blocks > 0xffffffffull) return set_error(HEVCASM_HIP_EINVAL, "n_blocks exceeds 2^32-1");
	if (!rec || !pred || !rec_off || !pred_off || !res) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	hipStream_t st = as_stream(stream);
#define LAUNCH(N) { \
		constexpr int lpb = (N / (N < 8 ? N : 8)) * N; \
		const uint32_t grid = ceil_div(n_blocks, kThreads / lpb); \
		hipLaunchKernelGGL((reconstruct_kernel<N>), dim3(grid), dim3(kThreads), 0, st, rec, stride_rec, pred, stride_pred, rec_off, pred_off, res, (uint32_t)n_blocks); \
		return launch_status("reconstruct_kernel"); }
	if (nCbS == 4)
	{
		hipLaunchKernelGGL(reconstruct4_lane_kernel, dim3(ceil_div(n_blocks, kThreads)), dim3(kThreads), 0, st, rec, stride_rec, pred, stride_pred, rec_off, pred_off, res, (uint32_t)n_blocks);
		return launch_status("reconstruct4_lane_kernel");
	}
	switch (nCbS)
	{
	case 8: LAUNCH(8)
	case 16: LAUNCH(16)
	case 32: LAUNCH(32)
	}
#undef LAUNCH
	return set_error(HEVCASM_HIP_EINVAL, "nCbS must be 4, 8, 16 or 32");
}
