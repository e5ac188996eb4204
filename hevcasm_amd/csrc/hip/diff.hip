// diff.hip -- batched linear SSD: hevcasm_ssd_linear (reference diff.h:52, C
// body diff.c:45-54) over n_blocks independent runs of `size` contiguous bytes.
//
// LPS lanes per run (64 / 16 / 4 by run length); a lane takes every LPS-th 16-byte chunk, the
// size % 16 tail is taken byte-wise.  sum (a-b)^2 = sum a^2 + sum b^2 - 2 sum ab with v_dot4_u32_u8,
// in 32-bit wrap-around arithmetic -- equal to the reference's int sum whenever that is defined
// (size < 33 025 bytes cannot overflow) and its two's-complement wrap beyond.  HBM-bound:
// algorithmic bytes per run = 2 * size + 4 + 8.

#include "common.hpp"

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
