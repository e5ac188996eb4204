// diff.hip -- batched linear SSD: hevcasm_ssd_linear (reference diff.h:52, C
// body diff.c:45-54) over n_blocks independent runs of `size` contiguous bytes.
//
// LPS lanes per run (64 / 16 / 4 by run length); a lane takes every LPS-th 16-byte chunk, the
// size % 16 tail is taken byte-wise.  sum (a-b)^2 = sum a^2 + sum b^2 - 2 sum ab with v_dot4_u32_u8,
// in 32-bit wrap-around arithmetic -- equal to the reference's int sum whenever that is defined
// (size < 33 025 bytes cannot overflow) and its two's-complement wrap beyond.  HBM-bound:
// algorithmic bytes per run = 2 * size + 4 + 8.

#include "common.hpp"

#include <algorithm>

namespace hevc {

template <int LPS>
__global__ __launch_bounds__(kThreads) void ssd_linear_kernel(
	const uint8_t *__restrict__ src0, const uint8_t *__restrict__ src1,
	const uint32_t *__restrict__ off0, const uint32_t *__restrict__ off1, int *__restrict__ ssd,
	uint32_t size, uint32_t n_blocks)
{
	const uint32_t t = xcd_contiguous(blockIdx.x, gridDim.x) * kThreads + threadIdx.x;
	uint32_t run = t / LPS;
	const uint32_t j = t % LPS;
	const bool live = run < n_blocks;
	if (!live) run = n_blocks - 1;
	const uint8_t *a = src0 + off0[run], *b = src1 + off1[run];

	uint32_t sq = 0, ab = 0;
	const uint32_t whole = size & ~15u;
	for (uint32_t o = 16 * j; o < whole; o += 16 * LPS)
	{
		Chunk<16> x, y;
		x.load_stream(a + o);   // both runs are read exactly once
		y.load_stream(b + o);
#pragma unroll
		for (int k = 0; k < 4; ++k)
		{
			sq = __builtin_amdgcn_udot4(x.d[k], x.d[k], sq, false);
			sq = __builtin_amdgcn_udot4(y.d[k], y.d[k], sq, false);
			ab = __builtin_amdgcn_udot4(x.d[k], y.d[k], ab, false);
		}
	}
	for (uint32_t o = whole + j; o < size; o += LPS)
	{
		const uint32_t x = a[o], y = b[o];
		sq += x * x + y * y;
		ab += x * y;
	}
	uint32_t v = sq - 2u * ab;
#pragma unroll
	for (int d = LPS / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, kWave);
	if (live && j == 0) ssd[run] = (int)v;
}

// Whole-frame SSE (second half): frame i = `width` x `height` bytes, rows `stride` apart, at
// a + a_off[i] against b + b_off[i]; sse[i] = sum over the rows of hevcasm_ssd_linear(row a, row b, width) (reference
// diff.c:45-54), accumulated in 64 bits -- a row is at most 32 768 bytes here so every row sum is exact in the reference's
// int, and a 3840 x 2160 frame of 0 against 255 (5.4e11) needs the wide sum.
//
// A workgroup owns kThreads / LPR row slots and walks `rows_per_wg` rows of one frame; LPR lanes share a row, 16 bytes each
// and LPR * 16 bytes apart, tail bytes one per lane.  Per chunk sum (a-b)^2 = sum a^2 + sum b^2 - 2 sum ab <= 16 * 255^2
// (three v_dot4_u32_u8 per dword), added to a 64-bit lane total; wave totals by __shfl_xor, the four waves' through LDS, then ONE
// 64-bit atomic add per workgroup into sse[i] (integer: the order does not matter).  The launcher keeps a frame to at most 128
// workgroups: adds to one address are serialised at the memory side (4 320 of them -- a wave each, two rows a workgroup -- made a
// single 4K frame take 64 us, profiles/r05_frame_sse.txt).  HBM-bound: algorithmic bytes per frame = 2 * width * height + 8 + 8.
template <int LPR, bool SINGLE /* a row is at most LPR chunks: one per lane, so that the unrolled rows' loads are all issued before the first is used */>
__global__ __launch_bounds__(kThreads) void frame_sse_kernel(
	const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, const uint32_t *__restrict__ a_off, const uint32_t *__restrict__ b_off,
	unsigned long long *__restrict__ sse, int64_t stride_a, int64_t stride_b, uint32_t width, uint32_t height, uint32_t rows_per_wg, uint32_t wgs_per_frame)
{
	constexpr uint32_t kSlots = kThreads / LPR;
	__shared__ unsigned long long wave_total[kThreads / kWave];
	const uint32_t wg = xcd_contiguous(blockIdx.x, gridDim.x);
	const uint32_t frame = wg / wgs_per_frame;
	const uint32_t row0 = (wg % wgs_per_frame) * rows_per_wg;
	const uint32_t row1 = min(row0 + rows_per_wg, height);
	const uint32_t lane = threadIdx.x % LPR, slot = threadIdx.x / LPR;
	const uint8_t *pa = a + a_off[frame], *pb = b + b_off[frame];
	const uint32_t whole = width & ~15u;

	unsigned long long total = 0;
#pragma unroll 4
	for (uint32_t y = row0 + slot; y < row1; y += kSlots)
	{
		const uint8_t *ra = pa + (int64_t)y * stride_a, *rb = pb + (int64_t)y * stride_b;
		auto chunk = [&](uint32_t o)
		{
			Chunk<16> x, z;
			x.load_stream(ra + o);   // both frames are read exactly once
			z.load_stream(rb + o);
			uint32_t sq = 0, ab = 0;
#pragma unroll
			for (int k = 0; k < 4; ++k)
			{
				sq = __builtin_amdgcn_udot4(x.d[k], x.d[k], sq, false);
				sq = __builtin_amdgcn_udot4(z.d[k], z.d[k], sq, false);
				ab = __builtin_amdgcn_udot4(x.d[k], z.d[k], ab, false);
			}
			total += sq - 2u * ab;
		};
		if constexpr (SINGLE) { if (16 * lane < whole) chunk(16 * lane); }
		else for (uint32_t o = 16 * lane; o < whole; o += 16 * LPR) chunk(o);
		if (whole + lane < width)    // LPR >= 16: one tail byte per lane
		{
			const int d = (int)ra[whole + lane] - (int)rb[whole + lane];
			total += (uint32_t)(d * d);
		}
	}
	uint32_t lo = (uint32_t)total, hi = (uint32_t)(total >> 32);
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1)
	{
		const unsigned long long other = ((unsigned long long)(uint32_t)__shfl_xor((int)hi, d, kWave) << 32) | (uint32_t)__shfl_xor((int)lo, d, kWave);
		total += other;
		lo = (uint32_t)total; hi = (uint32_t)(total >> 32);
	}
	if (threadIdx.x % kWave == 0) wave_total[threadIdx.x / kWave] = total;
	__syncthreads();
	if (threadIdx.x == 0)
	{
		unsigned long long t = 0;
#pragma unroll
		for (int w = 0; w < kThreads / kWave; ++w) t += wave_total[w];
		if (t) atomicAdd(&sse[frame], t);
	}
}

}  // namespace hevc

using namespace hevc;

extern "C" int hevcasm_hip_ssd_linear_batch(
	const uint8_t *src0, const uint8_t *src1, const uint32_t *off0, const uint32_t *off1, int *ssd,
	int size, size_t n_blocks, void *stream)
{
	if (int e = check_device()) return e;
	if (size < 0) return set_error(HEVCASM_HIP_EINVAL, "size must be >= 0");
	if (n_blocks == 0) return 0;
	if (n_blocks > 0x3ffffffull) return set_error(HEVCASM_HIP_EINVAL, "n_blocks too large");
	if (!src0 || !src1 || !off0 || !off1 || !ssd) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	hipStream_t st = as_stream(stream);
	if (size >= 1024)
		hipLaunchKernelGGL(ssd_linear_kernel<64>, dim3(ceil_div(n_blocks * 64, kThreads)), dim3(kThreads), 0, st, src0, src1, off0, off1, ssd, (uint32_t)size, (uint32_t)n_blocks);
	else if (size >= 128)
		hipLaunchKernelGGL(ssd_linear_kernel<16>, dim3(ceil_div(n_blocks * 16, kThreads)), dim3(kThreads), 0, st, src0, src1, off0, off1, ssd, (uint32_t)size, (uint32_t)n_blocks);
	else
		hipLaunchKernelGGL(ssd_linear_kernel<4>, dim3(ceil_div(n_blocks * 4, kThreads)), dim3(kThreads), 0, st, src0, src1, off0, off1, ssd, (uint32_t)size, (uint32_t)n_blocks);
	return launch_status("ssd_linear_kernel");
}

extern "C" int hevcasm_hip_frame_sse_batch(
	const uint8_t *a, ptrdiff_t stride_a, const uint8_t *b, ptrdiff_t stride_b, const uint32_t *a_off, const uint32_t *b_off,
	uint64_t *sse, int width, int height, size_t n_frames, void *stream)
{
	if (int e = check_device()) return e;
	if (width < 0 || height < 0 || width > 32768 || height > 65536) return set_error(HEVCASM_HIP_EINVAL, "width must be 0..32768, height 0..65536");
	if (n_frames == 0) return 0;
	if (n_frames > 0xfffffull) return set_error(HEVCASM_HIP_EINVAL, "n_frames too large");
	if (!a || !b || !a_off || !b_off || !sse) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	if (reinterpret_cast<uintptr_t>(sse) & 7u) return set_error(HEVCASM_HIP_EINVAL, "sse must be 8-byte aligned (64-bit atomic adds)");
	hipStream_t st = as_stream(stream);
	HEVC_HIP_TRY(hipMemsetAsync(sse, 0, 8 * n_frames, st));
	if (width == 0 || height == 0) return 0;
	// lanes per row: the smallest of 16 / 64 / 128 / 256 that covers the row's 16-byte chunks (a 4K row: 240 chunks on 256 lanes; 1080p: 120 on 128)
	const uint32_t chunks = ((uint32_t)width + 15) / 16;
	const uint32_t lpr = chunks <= 16 ? 16 : chunks <= 64 ? 64 : chunks <= 128 ? 128 : 256;
	const uint32_t slots = kThreads / lpr;
	// workgroups per frame: about 8 per CU over the whole batch (one resident generation), never more than 128 for one frame (one atomic add each, all to sse[i]) nor
	// fewer rows than one round of the slots; a workgroup's rows are whole rounds of its slots
	const size_t want = std::max<size_t>(1, std::min<size_t>(128, ((size_t)device_cus() * 8 + n_frames - 1) / n_frames));
	const uint32_t rounds = std::max<uint32_t>(1, ceil_div(ceil_div((size_t)height, slots), want));
	const uint32_t rows_per_wg = rounds * slots;
	const uint32_t wgs_per_frame = ceil_div((size_t)height, rows_per_wg);
	const size_t grid = n_frames * (size_t)wgs_per_frame;
	if (grid > 0x7fffffffull) return set_error(HEVCASM_HIP_EINVAL, "batch too large for one launch");
	unsigned long long *out = reinterpret_cast<unsigned long long *>(sse);
#define HEVC_LAUNCH_SSE(L, S) hipLaunchKernelGGL((frame_sse_kernel<L, S>), dim3((uint32_t)grid), dim3(kThreads), 0, st, a, b, a_off, b_off, out, \
	(int64_t)stride_a, (int64_t)stride_b, (uint32_t)width, (uint32_t)height, rows_per_wg, wgs_per_frame)
	if (lpr == 16) HEVC_LAUNCH_SSE(16, true);
	else if (lpr == 64) HEVC_LAUNCH_SSE(64, true);
	else if (lpr == 128) HEVC_LAUNCH_SSE(128, true);
	else if (chunks <= 256) HEVC_LAUNCH_SSE(256, true);
	else HEVC_LAUNCH_SSE(256, false);
#undef HEVC_LAUNCH_SSE
	return launch_status("frame_sse_kernel");
}
