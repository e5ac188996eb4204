// pred_intra.hip -- batched intra DC prediction (as far as the reference goes:
// its table wires only INTRA_DC, with and without the edge filter -- pred_intra.c:61-101, lookup
// pred_intra.c:157).
//
// Block i: neighbours_i = neighbours + i * neighbour_stride is what the reference function receives:
// the left column bottom-up ends at neighbours_i[63], the top row starts at neighbours_i[64], so the
// 2N samples DC needs are the contiguous bytes [64 - N, 64 + N).  dst holds n_blocks consecutive
// N x N blocks (the reference writes with stride N).
// One lane per 16 output bytes.  The lanes of a block load the 2N neighbour bytes once between them,
// sum them with v_sad_u8 against zero, and exchange partial sums with wave shuffles.  HBM-bound:
// algorithmic bytes per block = N * N + 2 N.

#include "common.hpp"

namespace hevc {

template <int N>
__global__ __launch_bounds__(kThreads) void pred_intra_dc_kernel(
	uint8_t *__restrict__ dst, const uint8_t *__restrict__ neighbours, ptrdiff_t neighbour_stride,
	int filter_edge, uint32_t n_blocks)
{
	constexpr int LPB = N * N / 16;            // lanes per block: 1, 4, 16, 64
	constexpr int LOG2 = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
	const uint32_t t = xcd_contiguous(blockIdx.x, gridDim.x) * kThreads + threadIdx.x;
	uint32_t blk = t / LPB;
	const uint32_t j = t % LPB;
	const bool live = blk < n_blocks;
	if (!live) blk = n_blocks - 1;
	const uint8_t *nb = neighbours + (ptrdiff_t)blk * neighbour_stride;

	// DC value, pred_intra.c:66-76: (N + sum of the 2N samples) >> (log2 N + 1)
	uint32_t sum = 0;
	if (N == 4)
	{
		Chunk<8> c;
		c.load(nb + 60);
		sum = __builtin_amdgcn_sad_u8(c.d[0], 0, __builtin_amdgcn_sad_u8(c.d[1], 0, 0));
	}
	else if ((int)j < 2 * N / 16)
	{
		Chunk<16> c;
		c.load(nb + 64 - N + 16 * j);
#pragma unroll
		for (int k = 0; k < 4; ++k) sum = __builtin_amdgcn_sad_u8(c.d[k], 0, sum);
	}
	if (N >= 8)
	{
		constexpr int LOADERS = 2 * N / 16;
#pragma unroll
		for (int d = LOADERS / 2; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, kWave);
		sum = __shfl(sum, (int)((threadIdx.x & 63) - j), kWave);      // the block's first lane holds the total
	}
	const int dc = (int)(sum + N) >> (LOG2 + 1);

	Chunk<16> out;
#pragma unroll
	for (int k = 0; k < 4; ++k) out.d[k] = (uint32_t)dc * 0x01010101u;
	if (filter_edge)
	{
		// pred_intra.c:80-93: first row and first column blend the neighbours into the DC value.
		// A lane owns 16 consecutive output bytes = ROWS rows of min(N, 16) columns.
		constexpr int ROWS = N >= 16 ? 1 : 16 / N, COLS = N >= 16 ? 16 : N;
		const int y0 = 16 * (int)j / N, x0 = 16 * (int)j % N;
		uint8_t px[16];
#pragma unroll
		for (int b = 0; b < 16; ++b) px[b] = (uint8_t)dc;
		if (y0 == 0)
		{
			Chunk<COLS> top;                       // samples (x0 .. x0 + COLS - 1, -1)
			top.load(nb + 64 + x0);
#pragma unroll
			for (int b = 0; b < COLS; ++b) px[b] = (uint8_t)((((top.d[b / 4] >> (8 * (b % 4))) & 0xff) + 3 * dc + 2) >> 2);
		}
		if (x0 == 0)
		{
#pragma unroll
			for (int r = 0; r < ROWS; ++r)
				if (y0 + r > 0) px[r * COLS] = (uint8_t)((nb[63 - (y0 + r)] + 3 * dc + 2) >> 2);
			if (y0 == 0) px[0] = (uint8_t)((nb[63] + 2 * dc + nb[64] + 2) >> 2);
		}
#pragma unroll
		for (int k = 0; k < 4; ++k)
			out.d[k] = px[4 * k] | ((uint32_t)px[4 * k + 1] << 8) | ((uint32_t)px[4 * k + 2] << 16) | ((uint32_t)px[4 * k + 3] << 24);
	}
	if (live) out.store(dst + (size_t)blk * (N * N) + 16 * j);
}

}  // namespace hevc

using namespace hevc;

extern "C" int hevcasm_hip_pred_intra_batch(
	uint8_t *dst, const uint8_t *neighbours, ptrdiff_t neighbour_stride,
	int intraPredMode, int packed, size_t n_blocks, void *stream)
{
	if (int e = check_device()) return e;
	const int log2n = (packed >> 1) & 0x7f, filter_edge = packed & 1;
	if (intraPredMode != 1) return set_error(HEVCASM_HIP_EINVAL, "only INTRA_DC (mode 1) has an implementation in the reference");
	if (log2n < 2 || log2n > 5) return set_error(HEVCASM_HIP_EINVAL, "block size must be 4, 8, 16 or 32");
	if (filter_edge && log2n == 5) return set_error(HEVCASM_HIP_EINVAL, "the edge filter applies to blocks below 32x32");
	if (n_blocks == 0) return 0;
	if (n_blocks > (0xffffffffull >> (2 * log2n - 4)) - kThreads) return set_error(HEVCASM_HIP_EINVAL, "n_blocks too large");
	if (!dst || !neighbours) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	hipStream_t st = as_stream(stream);
	const int n = 1 << log2n;
	const uint32_t grid = ceil_div(n_blocks * (size_t)(n * n / 16), kThreads);
	switch (log2n)
	{
	case 2: hipLaunchKernelGGL(pred_intra_dc_kernel<4>, dim3(grid), dim3(kThreads), 0, st, dst, neighbours, neighbour_stride, filter_edge, (uint32_t)n_blocks); break;
	case 3: hipLaunchKernelGGL(pred_intra_dc_kernel<8>, dim3(grid), dim3(kThreads), 0, st, dst, neighbours, neighbour_stride, filter_edge, (uint32_t)n_blocks); break;
	case 4: hipLaunchKernelGGL(pred_intra_dc_kernel<16>, dim3(grid), dim3(kThreads), 0, st, dst, neighbours, neighbour_stride, filter_edge, (uint32_t)n_blocks); break;
	default: hipLaunchKernelGGL(pred_intra_dc_kernel<32>, dim3(grid), dim3(kThreads), 0, st, dst, neighbours, neighbour_stride, filter_edge, (uint32_t)n_blocks); break;
	}
	return launch_status("pred_intra_dc_kernel");
}
