// sad.hip -- batched SAD, 4-way multi-reference SAD and SSD for gfx950.
//
// Replaces, for a whole batch per launch: hevcasm_sad (reference sad.h:50, C body sad.c:47-60),
// hevcasm_sad_multiref 4-way (sad.h:95, sad.c:101-121) and hevcasm_ssd (ssd.h:53, ssd.c:43-55).
//
// Layout: frames stay as the codec has them (row-strided u8 planes in HBM); a block is a base
// pointer + u32 byte offset.  Mapping: each block row of W bytes is cut into chunks of
// C = 16 / 8 / 4 bytes (the widest that divides W); LPB = W / C adjacent lanes own one block and
// every lane walks its chunk down all H rows.  A wave therefore covers 64 / LPB blocks, and when
// consecutive blocks are horizontal neighbours in the frame (the usual case) one wave-wide load
// instruction reads one contiguous run of up to 1 KiB of a frame row: fully coalesced, no LDS.
// Reference blocks sit at arbitrary byte alignment (motion vectors); the loads carry no
// alignment promise and the hardware splits the few that straddle a line.
// Arithmetic: v_sad_u8 accumulates 4 byte differences per instruction; SSD uses
// sum(a-b)^2 = sum a^2 + sum b^2 - 2 sum ab with three v_dot4_u32_u8 (exact: every partial sum
// is below 2^30 for blocks up to 64x64).  The LPB partial sums are combined with wave shuffles.
// HBM-bound: algorithmic bytes per block = (1 + WAYS) * W * H + 4 * WAYS (+ 4 * (1 + WAYS) of offsets).

#include "common.hpp"

namespace hevc {

template <int W> struct Tiling
{
	static constexpr int C = (W % 16 == 0) ? 16 : ((W % 8 == 0) ? 8 : 4);  // chunk bytes per lane per row
	static constexpr int LPB = W / C;                                       // lanes per block
	static constexpr int BPW = kWave / LPB;                                 // blocks per wave
};

enum { OP_SAD = 0, OP_SSD = 1 };

template <int C, int OP> struct Acc
{
	uint32_t a, b;  // SAD: a only.  SSD: a = sum x^2 + sum y^2, b = sum x*y
	__device__ __forceinline__ Acc() : a(0), b(0) {}
	__device__ __forceinline__ void add(const Chunk<C> &x, const Chunk<C> &y)
	{
#pragma unroll
		for (int i = 0; i < C / 4; ++i)
		{
			if (OP == OP_SAD)
			{
				a = __builtin_amdgcn_sad_u8(x.d[i], y.d[i], a);
			}
			else
			{
				a = __builtin_amdgcn_udot4(x.d[i], x.d[i], a, false);
				a = __builtin_amdgcn_udot4(y.d[i], y.d[i], a, false);
				b = __builtin_amdgcn_udot4(x.d[i], y.d[i], b, false);
			}
		}
	}
	__device__ __forceinline__ uint32_t value() const { return OP == OP_SAD ? a : a - 2u * b; }
};

// Sum `v` over the LPB consecutive lanes of a block; the result is valid in the block's first lane.
template <int LPB> __device__ __forceinline__ uint32_t reduce_lanes(uint32_t v)
{
	if (LPB == 1) return v;
	if ((LPB & (LPB - 1)) == 0)
	{
#pragma unroll
		for (int d = LPB / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, kWave);
		return v;
	}
	uint32_t t = v;
#pragma unroll
	for (int j = 1; j < LPB; ++j) t += __shfl_down(v, j, kWave);
	return t;
}

template <int W, int WAYS, int OP>
__global__ __launch_bounds__(kThreads) void block_cost_kernel(
	const uint8_t *__restrict__ src, ptrdiff_t stride_src,
	const uint8_t *__restrict__ ref, ptrdiff_t stride_ref,
	const uint32_t *__restrict__ src_off, const uint32_t *__restrict__ ref_off,
	int *__restrict__ out, int H, uint32_t n_blocks)
{
	typedef Tiling<W> T;
	constexpr int C = T::C, LPB = T::LPB, BPW = T::BPW;

	const uint32_t wg = xcd_contiguous(blockIdx.x, gridDim.x);
	const uint32_t wave = wg * (kThreads / kWave) + (threadIdx.x >> 6);
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t sub = lane / LPB, chunk = lane - sub * LPB;

	// Lanes past the batch (or the spare lanes of a wave when 64 % LPB != 0) redo the last valid
	// block so that every lane executes the same loads and shuffles; only real leaders store.
	uint32_t blk = wave * BPW + (sub < (uint32_t)BPW ? sub : BPW - 1);
	const bool live = sub < (uint32_t)BPW && blk < n_blocks;
	if (blk >= n_blocks) blk = n_blocks - 1;

	const uint8_t *ps = src + src_off[blk] + chunk * C;
	const uint8_t *pr[WAYS];
#pragma unroll
	for (int k = 0; k < WAYS; ++k) pr[k] = ref + ref_off[(size_t)blk * WAYS + k] + chunk * C;

	Acc<C, OP> acc[WAYS];

	// H is a multiple of 4: four rows per trip keeps 4 * (1 + WAYS) independent loads in flight per lane.
	for (int y = 0; y < H; y += 4)
	{
		Chunk<C> s[4], r[4][WAYS];
#pragma unroll
		for (int j = 0; j < 4; ++j)
		{
			s[j].load(ps + j * stride_src);
#pragma unroll
			for (int k = 0; k < WAYS; ++k) r[j][k].load(pr[k] + j * stride_ref);
		}
#pragma unroll
		for (int j = 0; j < 4; ++j)
#pragma unroll
			for (int k = 0; k < WAYS; ++k) acc[k].add(s[j], r[j][k]);
		ps += 4 * stride_src;
#pragma unroll
		for (int k = 0; k < WAYS; ++k) pr[k] += 4 * stride_ref;
	}

	uint32_t total[WAYS];
#pragma unroll
	for (int k = 0; k < WAYS; ++k) total[k] = reduce_lanes<LPB>(acc[k].value());

	if (live && chunk == 0)
	{
		if (WAYS == 4)
		{
			// one 16-byte store per block; `out` only needs int alignment
			u32x4 v;
			v.x = total[0]; v.y = total[WAYS > 1 ? 1 : 0]; v.z = total[WAYS > 2 ? 2 : 0]; v.w = total[WAYS > 3 ? 3 : 0];
			reinterpret_cast<U4 *>(out + (size_t)blk * 4)->v = v;
		}
		else
		{
			out[blk] = (int)total[0];
		}
	}
}

template <int W, int WAYS, int OP>
static int launch_w(const uint8_t *src, ptrdiff_t ss, const uint8_t *ref, ptrdiff_t sr,
		const uint32_t *so, const uint32_t *ro, int *out, int H, size_t n, hipStream_t st)
{
	constexpr int blocks_per_wg = Tiling<W>::BPW * (kThreads / kWave);
	const uint32_t grid = ceil_div(n, blocks_per_wg);
	hipLaunchKernelGGL((block_cost_kernel<W, WAYS, OP>), dim3(grid), dim3(kThreads), 0, st,
			src, ss, ref, sr, so, ro, out, H, (uint32_t)n);
	return launch_status("block_cost_kernel");
}

template <int WAYS, int OP>
static int launch(const uint8_t *src, ptrdiff_t ss, const uint8_t *ref, ptrdiff_t sr,
		const uint32_t *so, const uint32_t *ro, int *out, int W, int H, size_t n, void *stream)
{
	if (int e = check_device()) return e;
	if (W < 4 || W > 64 || H < 4 || H > 64 || (W & 3) || (H & 3))
		return set_error(HEVCASM_HIP_EINVAL, "block size must be a multiple of 4 in 4..64");
	if (n == 0) return 0;
	if (n > 0xffffffffull) return set_error(HEVCASM_HIP_EINVAL, "n_blocks exceeds 2^32-1");
	if (!src || !ref || !so || !ro || !out) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	hipStream_t st = as_stream(stream);
	switch (W)
	{
#define CASE(w) case w: return launch_w<w, WAYS, OP>(src, ss, ref, sr, so, ro, out, H, n, st);
		CASE(4) CASE(8) CASE(12) CASE(16) CASE(20) CASE(24) CASE(28) CASE(32)
		CASE(36) CASE(40) CASE(44) CASE(48) CASE(52) CASE(56) CASE(60) CASE(64)
#undef CASE
	}
	return set_error(HEVCASM_HIP_EINVAL, "unsupported width");
}

}  // namespace hevc

extern "C" int hevcasm_hip_sad_batch(
	const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref,
	const uint32_t *src_off, const uint32_t *ref_off, int *sad,
	uint32_t rect, size_t n_blocks, void *stream)
{
	return hevc::launch<1, hevc::OP_SAD>(src, stride_src, ref, stride_ref, src_off, ref_off, sad,
			(int)(rect >> 8), (int)(rect & 0xff), n_blocks, stream);
}

extern "C" int hevcasm_hip_sad_multiref_batch(
	const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref,
	const uint32_t *src_off, const uint32_t *ref_off, int *sad,
	uint32_t rect, size_t n_blocks, void *stream)
{
	return hevc::launch<4, hevc::OP_SAD>(src, stride_src, ref, stride_ref, src_off, ref_off, sad,
			(int)(rect >> 8), (int)(rect & 0xff), n_blocks, stream);
}

extern "C" int hevcasm_hip_ssd_batch(
	const uint8_t *srcA, ptrdiff_t stride_srcA, const uint8_t *srcB, ptrdiff_t stride_srcB,
	const uint32_t *offA, const uint32_t *offB, int *ssd,
	int w, int h, size_t n_blocks, void *stream)
{
	return hevc::launch<1, hevc::OP_SSD>(srcA, stride_srcA, srcB, stride_srcB, offA, offB, ssd, w, h, n_blocks, stream);
}
