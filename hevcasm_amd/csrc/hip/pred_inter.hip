// pred_inter.hip -- HEVC uni- and bi-directional inter prediction (motion compensation), 8-bit.
//
// Replaces hevcasm_pred_uni_8to8 (reference pred_inter.h:56; C bodies pred_inter.c:141-228) and
// hevcasm_pred_bi_8to8 (pred_inter.h:76; C body pred_inter.c:490-530) for whole batches of
// equally sized prediction units (PUs) with per-PU motion-vector fractions.
//
// Arithmetic.  One code path serves every fraction pair: horizontal FIR at full precision into
// int16 (never overflows: |sum| <= 88 * 255), then vertical FIR in 32 bits.
//   uni: (v + 2048) >> 12, clip to 8 bits.  Identical to the reference's copy / H-only / V-only /
//        HV special cases because an unfiltered direction is the tap {64}, i.e. an exact factor
//        64 that the final shift removes (64 * (s + 32) == 64 s + 2048).
//   bi:  v >> 6 per reference (no rounding), then (a + b + 64) >> 7, clip.
//
// Mapping.  LPP lanes (8..256, chosen from the PU area) own one PU; a 256-thread workgroup
// carries 256 / LPP PUs.  Per PU the (W + T - 1) x (H + T - 1) reference window is copied to LDS
// (the line buffer of the separable filter), the horizontal pass writes an int16 plane to LDS,
// and the vertical pass reads it column-wise.  Lanes are laid out LX across by LPP / LX down so
// consecutive lanes touch consecutive pixels.
// Algorithmic bytes per PU: refs * (W + T - 1)(H + T - 1) + W * H.

#include "common.hpp"

namespace hevc {

__constant__ signed char kLumaTaps[4][8] = {
	{ 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 },
	{ -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
__constant__ signed char kChromaTaps[8][4] = {
	{ 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
	{ -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };

template <int TAPS> __device__ __forceinline__ void load_taps(int frac, int (&c)[TAPS])
{
#pragma unroll
	for (int k = 0; k < TAPS; ++k) c[k] = TAPS == 8 ? kLumaTaps[frac & 3][k] : kChromaTaps[frac & 7][k];
}

struct PredGeom
{
	int W, H;
	int log2_lx;      // lanes across = 1 << log2_lx
	int win_stride;   // bytes per window row in LDS
	int win_bytes;    // window plane, rounded up to 16
	int tmp_bytes;    // int16 (H + T - 1) x W plane, rounded up to 16
	int pu_bytes;     // LDS per PU
};

template <int TAPS, int LPP, bool BI>
__global__ __launch_bounds__(kThreads) void pred_kernel(
	uint8_t *__restrict__ dst, ptrdiff_t stride_dst,
	const uint8_t *__restrict__ ref0, const uint8_t *__restrict__ ref1, ptrdiff_t stride_ref,
	const uint32_t *__restrict__ dst_off, const uint32_t *__restrict__ ref0_off, const uint32_t *__restrict__ ref1_off,
	const uint8_t *__restrict__ frac0, const uint8_t *__restrict__ frac1,
	PredGeom g, uint32_t n_blocks)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	constexpr int PPWG = kThreads / LPP;
	constexpr int BEFORE = TAPS / 2 - 1;

	const uint32_t wg = xcd_contiguous(blockIdx.x, gridDim.x);
	const uint32_t local = threadIdx.x / LPP, l = threadIdx.x % LPP;
	uint32_t pu = wg * PPWG + local;
	const bool live = pu < n_blocks;
	if (!live) pu = n_blocks - 1;

	uint8_t *win = lds + local * g.pu_bytes;
	int16_t *tmp = reinterpret_cast<int16_t *>(win + g.win_bytes);
	int16_t *first = reinterpret_cast<int16_t *>(win + g.win_bytes + g.tmp_bytes);  // bi only: 14-bit prediction of ref 0

	const int W = g.W, H = g.H, WS = g.win_stride;
	const int LX = 1 << g.log2_lx, LY = LPP >> g.log2_lx;
	const int lx = l & (LX - 1), ly = l >> g.log2_lx;
	const int rows = H + TAPS - 1, cols = W + TAPS - 1;

#pragma unroll
	for (int r = 0; r < (BI ? 2 : 1); ++r)
	{
		const uint8_t *ref = (r ? ref1 : ref0) + (r ? ref1_off : ref0_off)[pu];
		const int fr = (r ? frac1 : frac0)[pu];
		int cx[TAPS], cy[TAPS];
		load_taps<TAPS>(fr & 15, cx);
		load_taps<TAPS>(fr >> 4, cy);

		// reference window -> LDS
		for (int y = ly; y < rows; y += LY)
		{
			const uint8_t *src_row = ref + (ptrdiff_t)(y - BEFORE) * stride_ref - BEFORE;
			for (int x = lx; x < cols; x += LX) win[y * WS + x] = src_row[x];
		}
		__syncthreads();

		// horizontal pass, full precision (pred_inter.c:175, shift 0)
		for (int y = ly; y < rows; y += LY)
			for (int x = lx; x < W; x += LX)
			{
				int a = 0;
#pragma unroll
				for (int k = 0; k < TAPS; ++k) a += cx[k] * win[y * WS + x + k];
				tmp[y * W + x] = (int16_t)a;
			}
		__syncthreads();

		// vertical pass
		for (int y = ly; y < H; y += LY)
			for (int x = lx; x < W; x += LX)
			{
				int a = 0;
#pragma unroll
				for (int k = 0; k < TAPS; ++k) a += cy[k] * tmp[(y + k) * W + x];
				if (!BI)
				{
					if (live) dst[dst_off[pu] + (ptrdiff_t)y * stride_dst + x] = (uint8_t)clip3(0, 255, (a + 2048) >> 12);
				}
				else if (r == 0)
				{
					first[y * W + x] = (int16_t)(a >> 6);  // pred_inter.c:515, shift 6, no rounding
				}
				else
				{
					const int v = ((int)first[y * W + x] + (int)(int16_t)(a >> 6) + 64) >> 7;  // pred_inter.c:490-501
					if (live) dst[dst_off[pu] + (ptrdiff_t)y * stride_dst + x] = (uint8_t)clip3(0, 255, v);
				}
			}
		__syncthreads();
	}
}

static int round16(int v) { return (v + 15) & ~15; }

template <int TAPS, bool BI>
static int launch_pred(uint8_t *dst, ptrdiff_t sd, const uint8_t *ref0, const uint8_t *ref1, ptrdiff_t sr,
		const uint32_t *dof, const uint32_t *r0, const uint32_t *r1, const uint8_t *f0, const uint8_t *f1,
		int W, int H, size_t n, hipStream_t st)
{
	PredGeom g;
	g.W = W; g.H = H;
	g.win_stride = (W + TAPS - 1 + 3) & ~3;
	g.win_bytes = round16(g.win_stride * (H + TAPS - 1));
	g.tmp_bytes = round16(2 * W * (H + TAPS - 1));
	g.pu_bytes = g.win_bytes + g.tmp_bytes + (BI ? round16(2 * W * H) : 0);
	int lx = 0;
	while ((1 << lx) < W) ++lx;

	const int area = W * H;
	const int lpp = area >= 2048 ? 256 : area >= 1024 ? 128 : area >= 512 ? 64 : area >= 256 ? 32 : area >= 128 ? 16 : 8;
	int lpp_log2 = 0;
	while ((1 << lpp_log2) < lpp) ++lpp_log2;
	g.log2_lx = lx < lpp_log2 ? lx : lpp_log2;

	const int ppwg = kThreads / lpp;
	const uint32_t grid = ceil_div(n, ppwg);
	const size_t shmem = (size_t)g.pu_bytes * ppwg;
#define LAUNCH(L) hipLaunchKernelGGL((pred_kernel<TAPS, L, BI>), dim3(grid), dim3(kThreads), shmem, st, \
		dst, sd, ref0, ref1, sr, dof, r0, r1, f0, f1, g, (uint32_t)n); break;
	switch (lpp)
	{
	case 8: LAUNCH(8)
	case 16: LAUNCH(16)
	case 32: LAUNCH(32)
	case 64: LAUNCH(64)
	case 128: LAUNCH(128)
	default: LAUNCH(256)
	}
#undef LAUNCH
	return launch_status("pred_kernel");
}

static int check_pred_args(int taps, int W, int H, size_t n)
{
	if (taps != 8 && taps != 4) return set_error(HEVCASM_HIP_EINVAL, "taps must be 8 (luma) or 4 (chroma)");
	const int maxdim = taps == 8 ? 64 : 32;  // the reference's intermediate buffers (pred_inter.c:172, :221)
	if (W < 1 || H < 1 || W > maxdim || H > maxdim) return set_error(HEVCASM_HIP_EINVAL, "PU size outside the reference's domain");
	if (n > 0xffffffffull) return set_error(HEVCASM_HIP_EINVAL, "n_blocks exceeds 2^32-1");
	return 0;
}

}  // namespace hevc

using namespace hevc;

extern "C" int hevcasm_hip_pred_uni_8to8_batch(
	uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *ref, ptrdiff_t stride_ref,
	const uint32_t *dst_off, const uint32_t *ref_off, const uint8_t *frac,
	int taps, int nPbW, int nPbH, size_t n_blocks, void *stream)
{
	if (int e = check_device()) return e;
	if (int e = check_pred_args(taps, nPbW, nPbH, n_blocks)) return e;
	if (n_blocks == 0) return 0;
	if (!dst || !ref || !dst_off || !ref_off || !frac) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	hipStream_t st = as_stream(stream);
	return taps == 8
		? launch_pred<8, false>(dst, stride_dst, ref, ref, stride_ref, dst_off, ref_off, ref_off, frac, frac, nPbW, nPbH, n_blocks, st)
		: launch_pred<4, false>(dst, stride_dst, ref, ref, stride_ref, dst_off, ref_off, ref_off, frac, frac, nPbW, nPbH, n_blocks, st);
}

extern "C" int hevcasm_hip_pred_bi_8to8_batch(
	uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *ref0, const uint8_t *ref1, ptrdiff_t stride_ref,
	const uint32_t *dst_off, const uint32_t *ref0_off, const uint32_t *ref1_off,
	const uint8_t *frac0, const uint8_t *frac1,
	int taps, int nPbW, int nPbH, size_t n_blocks, void *stream)
{
	if (int e = check_device()) return e;
	if (int e = check_pred_args(taps, nPbW, nPbH, n_blocks)) return e;
	if (n_blocks == 0) return 0;
	if (!dst || !ref0 || !ref1 || !dst_off || !ref0_off || !ref1_off || !frac0 || !frac1) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	hipStream_t st = as_stream(stream);
	return taps == 8
		? launch_pred<8, true>(dst, stride_dst, ref0, ref1, stride_ref, dst_off, ref0_off, ref1_off, frac0, frac1, nPbW, nPbH, n_blocks, st)
		: launch_pred<4, true>(dst, stride_dst, ref0, ref1, stride_ref, dst_off, ref0_off, ref1_off, frac0, frac1, nPbW, nPbH, n_blocks, st);
}
