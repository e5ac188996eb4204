// satd.hip -- batched Hadamard SATD of 2x2, 4x4 and 8x8 blocks for gfx950.
//
// Replaces hevcasm_hadamard_satd (reference hadamard.h:53; C body hadamard.c:76-111) for whole
// batches.  The sum of |H D H'| does not depend on the ordering of the Hadamard basis, so any
// complete set of butterfly stages gives the reference's value; the rounding (+ n/4) and the
// integer division by n/2 are the reference's (hadamard.c:98, :110).
//
// Mapping: one lane per block (the whole 2-D transform stays in that lane's registers), lanes of a
// wave = consecutive blocks, so for raster-ordered blocks each row load of a wave is one contiguous
// run.  Values are held as int32 in this first version (|coefficient| <= 64 * 255 also fits packed
// int16: the next step).
// Bound: integer VALU for 8x8 (about 450 lane-ops per 132 bytes), vector L1 for 2x2 / 4x4.
// Algorithmic bytes per block: 2 n^2 + 4 (+ 8 of offsets).

#include "common.hpp"

namespace hevc {

template <int N> __device__ __forceinline__ void hadamard_inplace(int (&v)[N])
{
#pragma unroll
	for (int span = N / 2; span >= 1; span /= 2)
#pragma unroll
		for (int i = 0; i < N; i += 2 * span)
#pragma unroll
			for (int j = 0; j < span; ++j)
			{
				const int a = v[i + j], b = v[i + span + j];
				v[i + j] = a + b;
				v[i + span + j] = a - b;
			}
}

template <int N>
__global__ __launch_bounds__(kThreads) void hadamard_satd_kernel(
	const uint8_t *__restrict__ srcA, ptrdiff_t stride_a, const uint8_t *__restrict__ srcB, ptrdiff_t stride_b,
	const uint32_t *__restrict__ offA, const uint32_t *__restrict__ offB, int *__restrict__ out, uint32_t n_blocks)
{
	const uint32_t blk = xcd_contiguous(blockIdx.x, gridDim.x) * kThreads + threadIdx.x;
	if (blk >= n_blocks) return;
	const uint8_t *pa = srcA + offA[blk], *pb = srcB + offB[blk];

	// all rows of the first block, then all rows of the second (not alternating: see the load-order note in sad.hip)
	uint32_t ra[N][(N + 3) / 4], rb[N][(N + 3) / 4];
#pragma unroll
	for (int y = 0; y < N; ++y)
	{
		if (N == 2) ra[y][0] = reinterpret_cast<const H1 *>(pa + y * stride_a)->v;
		else
		{
			Chunk<(N >= 4 ? N : 4)> c;
			c.load(pa + y * stride_a);
#pragma unroll
			for (int i = 0; i < N / 4; ++i) ra[y][i] = c.d[i];
		}
	}
	__builtin_amdgcn_sched_barrier(0);
#pragma unroll
	for (int y = 0; y < N; ++y)
	{
		if (N == 2) rb[y][0] = reinterpret_cast<const H1 *>(pb + y * stride_b)->v;
		else
		{
			Chunk<(N >= 4 ? N : 4)> c;
			c.load(pb + y * stride_b);
#pragma unroll
			for (int i = 0; i < N / 4; ++i) rb[y][i] = c.d[i];
		}
	}
	__builtin_amdgcn_sched_barrier(0);
	int d[N][N];
#pragma unroll
	for (int y = 0; y < N; ++y)
	{
		uint32_t a[(N + 3) / 4], b[(N + 3) / 4];
#pragma unroll
		for (int i = 0; i < (N + 3) / 4; ++i) { a[i] = ra[y][i]; b[i] = rb[y][i]; }
#pragma unroll
		for (int x = 0; x < N; ++x)
			d[y][x] = (int)((a[x / 4] >> (8 * (x & 3))) & 0xff) - (int)((b[x / 4] >> (8 * (x & 3))) & 0xff);
		hadamard_inplace<N>(d[y]);
	}
	int sum = N / 4;
#pragma unroll
	for (int x = 0; x < N; ++x)
	{
		int col[N];
#pragma unroll
		for (int y = 0; y < N; ++y) col[y] = d[y][x];
		hadamard_inplace<N>(col);
#pragma unroll
		for (int y = 0; y < N; ++y) sum += col[y] < 0 ? -col[y] : col[y];
	}
	out[blk] = sum / (N / 2);
}

}  // namespace hevc

using namespace hevc;

extern "C" int hevcasm_hip_hadamard_satd_batch(
	const uint8_t *srcA, ptrdiff_t stride_srcA, const uint8_t *srcB, ptrdiff_t stride_srcB,
	const uint32_t *offA, const uint32_t *offB, int *satd,
	int log2TrafoSize, size_t n_blocks, void *stream)
{
	if (int e = check_device()) return e;
	if (log2TrafoSize < 1 || log2TrafoSize > 3) return set_error(HEVCASM_HIP_EINVAL, "SATD size must be 2x2, 4x4 or 8x8");
	if (n_blocks == 0) return 0;
	if (n_blocks > 0xffffffffull) return set_error(HEVCASM_HIP_EINVAL, "n_blocks exceeds 2^32-1");
	if (!srcA || !srcB || !offA || !offB || !satd) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	const uint32_t grid = ceil_div(n_blocks, kThreads);
	hipStream_t st = as_stream(stream);
	switch (log2TrafoSize)
	{
	case 1: hipLaunchKernelGGL((hadamard_satd_kernel<2>), dim3(grid), dim3(kThreads), 0, st, srcA, stride_srcA, srcB, stride_srcB, offA, offB, satd, (uint32_t)n_blocks); break;
	case 2: hipLaunchKernelGGL((hadamard_satd_kernel<4>), dim3(grid), dim3(kThreads), 0, st, srcA, stride_srcA, srcB, stride_srcB, offA, offB, satd, (uint32_t)n_blocks); break;
	default: hipLaunchKernelGGL((hadamard_satd_kernel<8>), dim3(grid), dim3(kThreads), 0, st, srcA, stride_srcA, srcB, stride_srcB, offA, offB, satd, (uint32_t)n_blocks); break;
	}
	return launch_status("hadamard_satd_kernel");
}
