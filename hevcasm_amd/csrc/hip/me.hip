// me.hip -- full-search motion estimation: hevcasm_sad evaluated at every displacement of a
// +-range window, arg-min per block.
//
// One workgroup per block.  The source block and the (W + 2R) x (H + 2R) reference window are
// staged once in LDS (64x64, R = 64: 4 KiB + 36 KiB); each thread then walks its share of the
// (2R + 1)^2 candidates.  A candidate row starts at an arbitrary byte of the window, so each
// 4-byte group is assembled from two aligned LDS dwords with v_alignbyte_b32 and fed to
// v_sad_u8.  Consecutive threads take consecutive dx, i.e. consecutive LDS bytes: conflict-free.
// The arg-min key is (cost << 16) | candidate index, so ties go to the first candidate in raster
// order of (dy, dx), the order of the nested loops a CPU full search uses.
// VALU/LDS-bound (about 1.7 k byte operations per byte fetched), not HBM-bound.

#include "common.hpp"

namespace hevc {

__global__ __launch_bounds__(kThreads) void me_full_search_kernel(
	const uint8_t *__restrict__ src, ptrdiff_t stride_src, const uint8_t *__restrict__ ref, ptrdiff_t stride_ref,
	const uint32_t *__restrict__ src_off, const uint32_t *__restrict__ ref_off, int *__restrict__ result,
	int W, int H, int R, int win_stride /* bytes, multiple of 4 */)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	__shared__ unsigned long long best_of_wave[kThreads / kWave];

	const uint32_t blk = blockIdx.x;
	uint32_t *s_src = reinterpret_cast<uint32_t *>(lds);                 // H rows of W / 4 dwords
	uint8_t *s_win = lds + W * H;                                          // (H + 2R) rows of win_stride bytes
	const int wd = W / 4;
	const int win_w = W + 2 * R, win_h = H + 2 * R;

	const uint8_t *sp = src + src_off[blk];
	for (int i = threadIdx.x; i < wd * H; i += kThreads)
	{
		const int y = i / wd, x = i - y * wd;
		s_src[i] = reinterpret_cast<const U1 *>(sp + (ptrdiff_t)y * stride_src + 4 * x)->v;
	}
	const uint8_t *rp = ref + ref_off[blk] - (ptrdiff_t)R * stride_ref - R;
	const int row_dwords = (win_w + 3) / 4;
	for (int i = threadIdx.x; i < row_dwords * win_h; i += kThreads)
	{
		const int y = i / row_dwords, x = i - y * row_dwords;
		const uint8_t *p = rp + (ptrdiff_t)y * stride_ref + 4 * x;
		uint32_t v;
		if (4 * x + 4 <= win_w) v = reinterpret_cast<const U1 *>(p)->v;
		else
		{
			v = 0;  // last partial dword of a row: never read past the window the caller guarantees
			for (int b = 0; 4 * x + b < win_w; ++b) v |= (uint32_t)p[b] << (8 * b);
		}
		*reinterpret_cast<uint32_t *>(s_win + y * win_stride + 4 * x) = v;
	}
	__syncthreads();

	const int side = 2 * R + 1;
	const int n_cand = side * side;
	unsigned long long best = ~0ull;
	for (int c = threadIdx.x; c < n_cand; c += kThreads)
	{
		const int cy = c / side, cx = c - cy * side;   // = dy + R, dx + R
		uint32_t acc = 0;
		for (int y = 0; y < H; ++y)
		{
			const uint32_t a = (uint32_t)((y + cy) * win_stride + cx);
			const uint32_t *q = reinterpret_cast<const uint32_t *>(s_win + (a & ~3u));
			const uint32_t sh = a & 3u;
			uint32_t lo = q[0];
			const uint32_t *s = s_src + y * wd;
			for (int j = 0; j < wd; ++j)
			{
				const uint32_t hi = q[j + 1];
				acc = __builtin_amdgcn_sad_u8(s[j], __builtin_amdgcn_alignbyte(hi, lo, sh), acc);
				lo = hi;
			}
		}
		const unsigned long long key = ((unsigned long long)acc << 16) | (unsigned)c;
		best = key < best ? key : best;
	}

#pragma unroll
	for (int d = 32; d >= 1; d >>= 1)
	{
		const unsigned long long o = __shfl_xor(best, d, kWave);
		best = o < best ? o : best;
	}
	if ((threadIdx.x & 63) == 0) best_of_wave[threadIdx.x >> 6] = best;
	__syncthreads();
	if (threadIdx.x == 0)
	{
		for (int w = 1; w < kThreads / kWave; ++w) best = best_of_wave[w] < best ? best_of_wave[w] : best;
		const int c = (int)(best & 0xffff);
		result[3 * blk + 0] = c % side - R;
		result[3 * blk + 1] = c / side - R;
		result[3 * blk + 2] = (int)(best >> 16);
	}
}

}  // namespace hevc

using namespace hevc;

extern "C" int hevcasm_hip_me_full_search_batch(
	const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref,
	const uint32_t *src_off, const uint32_t *ref_off, int *result,
	uint32_t rect, int range, size_t n_blocks, void *stream)
{
	if (int e = check_device()) return e;
	const int W = (int)(rect >> 8), H = (int)(rect & 0xff);
	if (W < 4 || W > 64 || H < 4 || H > 64 || (W & 3) || (H & 3)) return set_error(HEVCASM_HIP_EINVAL, "block size must be a multiple of 4 in 4..64");
	if (range < 0 || range > 127) return set_error(HEVCASM_HIP_EINVAL, "range must be 0..127");
	if (n_blocks == 0) return 0;
	if (n_blocks > 0x7fffffffull) return set_error(HEVCASM_HIP_EINVAL, "n_blocks too large");
	if (!src || !ref || !src_off || !ref_off || !result) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	// one spare dword per row so that the (hi, lo) pair of the last group stays inside the row
	const int win_stride = ((W + 2 * range + 3) & ~3) + 4;
	const size_t shmem = (size_t)W * H + (size_t)win_stride * (H + 2 * range) + 16;
	if (shmem > 64 * 1024) return set_error(HEVCASM_HIP_EINVAL, "search window does not fit in 64 KiB of LDS");
	hipLaunchKernelGGL(me_full_search_kernel, dim3((uint32_t)n_blocks), dim3(kThreads), shmem, as_stream(stream),
			src, stride_src, ref, stride_ref, src_off, ref_off, result, W, H, range, win_stride);
	return launch_status("me_full_search_kernel");
}
