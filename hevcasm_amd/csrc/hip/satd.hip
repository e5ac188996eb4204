// satd.hip -- batched Hadamard SATD of 2x2, 4x4 and 8x8 blocks for gfx950.
//
// Replaces hevcasm_hadamard_satd (reference hadamard.h:53; C body hadamard.c:76-111) for whole
// batches.  The sum of |H D H'| does not depend on the ordering of the Hadamard basis, so any
// complete set of butterfly stages gives the reference's value; the rounding (+ n/4) and the
// integer division by n/2 are the reference's (hadamard.c:98, :110).
//
// Mapping: one lane per block (the whole 2-D transform stays in that lane's registers), lanes of a
// wave = consecutive blocks, so for raster-ordered blocks each row load of a wave is one contiguous
// run.  8x8 blocks are transformed as packed int16 pairs (satd8_packed); 2x2 and 4x4 as int32.
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

typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_add16(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (s16x2)(__builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b))); }
__device__ __forceinline__ uint32_t pk_sub16(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (s16x2)(__builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b))); }
// (lo, hi) -> (lo + hi, lo - hi): one v_pk_mad_i16 whose first operand reads the high half for both results and whose
// addend reads the low half for both (op_sel), times (+1, -1)
__device__ __forceinline__ uint32_t pk_sum_diff(uint32_t d)
{
	uint32_t r;
	const uint32_t plus_minus = 0xffff0001u;     // (+1, -1)
	asm("v_pk_mad_i16 %0, %1, %2, %1 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(d), "v"(plus_minus));
	return r;
}

// 8x8 with rows packed two to a register as int16 pairs: R[x][p] = (d[2p][x], d[2p+1][x]).  Every value of the transform
// fits int16 (|.| <= 64 * 255), so both passes run on v_pk_add/sub_i16: the stage between the two rows of a pair is one
// v_pk_mad_i16 (pk_sum_diff), the other two vertical stages and all three horizontal ones act on whole registers.
// About 0.55 of the int32 form's instructions.
__device__ __forceinline__ int satd8_packed(const uint32_t (&ra)[8][2], const uint32_t (&rb)[8][2])
{
	uint32_t R[8][4];
#pragma unroll
	for (int p = 0; p < 4; ++p)
#pragma unroll
		for (int x = 0; x < 8; ++x)
		{
			const uint32_t sel = 0x0c000c00u | ((4u + (x & 3)) << 16) | (uint32_t)(x & 3);    // (byte x of row 2p, byte x of row 2p + 1) as int16
			const uint32_t pa = __builtin_amdgcn_perm(ra[2 * p + 1][x >> 2], ra[2 * p][x >> 2], sel);
			const uint32_t pb = __builtin_amdgcn_perm(rb[2 * p + 1][x >> 2], rb[2 * p][x >> 2], sel);
			R[x][p] = pk_sum_diff(pk_sub16(pa, pb));       // vertical stage inside the pair
		}
	// vertical stages between pairs
#pragma unroll
	for (int x = 0; x < 8; ++x)
	{
		const uint32_t a0 = pk_add16(R[x][0], R[x][1]), a1 = pk_sub16(R[x][0], R[x][1]);
		const uint32_t a2 = pk_add16(R[x][2], R[x][3]), a3 = pk_sub16(R[x][2], R[x][3]);
		R[x][0] = pk_add16(a0, a2); R[x][2] = pk_sub16(a0, a2);
		R[x][1] = pk_add16(a1, a3); R[x][3] = pk_sub16(a1, a3);
	}
	// horizontal transform over x, then sum of absolute values
	int sum = 2;      // + n / 4 (hadamard.c:98)
#pragma unroll
	for (int p = 0; p < 4; ++p)
	{
		uint32_t v[8];
#pragma unroll
		for (int x = 0; x < 8; ++x) v[x] = R[x][p];
#pragma unroll
		for (int span = 4; span >= 1; span /= 2)
#pragma unroll
			for (int i = 0; i < 8; i += 2 * span)
#pragma unroll
				for (int j = 0; j < span; ++j)
				{
					const uint32_t a = v[i + j], b = v[i + span + j];
					v[i + j] = pk_add16(a, b);
					v[i + span + j] = pk_sub16(a, b);
				}
#pragma unroll
		for (int x = 0; x < 8; ++x)
		{
			const s16x2 a = __builtin_bit_cast(s16x2, v[x]);
			const s16x2 m = __builtin_elementwise_max(a, (s16x2)(-a));
			sum = __builtin_amdgcn_sdot2(m, (s16x2){ 1, 1 }, sum, false);
		}
	}
	return sum / 4;
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
	if constexpr (N == 8)
	{
		out[blk] = satd8_packed(ra, rb);
		return;
	}
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
