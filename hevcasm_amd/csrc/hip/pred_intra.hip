// pred_intra.hip -- batched intra prediction.  INTRA_DC is the reference's C (its table
// wires only that mode, with and without the edge filter -- pred_intra.c:61-101, lookup pred_intra.c:157) and is
// PINNED to it.  Planar and the 33 angular modes, and the neighbour smoothing, exist in the reference only as f265
// assembly for 4x4 / 8x8 (f265/intra.asm:216-1394), which cannot be assembled here: those follow H.265 8.4.4.2.3 /
// .4 / .6 on the reference's neighbour layout and packed argument (second half of this file) and are checked against
// the oracle's restatement of the same text -- PARITY UNPINNED, stated in DESIGN.md section 0.
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


// ---- all 35 modes (round 5; parity unpinned beyond DC) -------------------------------------------------------------------
//
// Block i gets neighbours_i = neighbours + i * neighbour_stride: p[-1][y] = neighbours_i[63 - y], p[x][-1] = neighbours_i[64 + x],
// p[-1][-1] = neighbours_i[128] (f265/intra.asm:126-131), 2N samples each way.  One lane per 16 output bytes, as the DC kernel.
// The lanes of a block first build in LDS what the mode reads -- angular: the text's ref[-N .. 2N] array (8.4.4.2.6: the main side, the corner, and
// for negative angles the other side projected by invAngle), one spare entry behind it so that the iFact == 0 rows need no branch; planar: the top
// row 0..N and the left column 0..N; DC: both edges -- and then every pixel is two LDS bytes and a blend.  Modes may differ from block to block
// (`modes`), so everything below is per-block data; a batch with ONE mode takes the same kernel with the mode in an SGPR.
// HBM-bound in principle (N^2 + 4N + 1 bytes per block); in this first form the per-pixel index arithmetic makes it VALU-bound.
__constant__ int8_t kIntraAngle[35] = { 0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
__constant__ int16_t kIntraInvAngle[15] = { -4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096 };

template <int N>
__global__ __launch_bounds__(kThreads) void pred_intra_modes_kernel(
	uint8_t *__restrict__ dst, const uint8_t *__restrict__ neighbours, ptrdiff_t neighbour_stride,
	const uint8_t *__restrict__ modes, int uniform_mode, int edge, uint32_t n_blocks)
{
	constexpr int LPB = N * N / 16;            // lanes per block: 1, 4, 16, 64
	constexpr int BPWG = kThreads / LPB;
	constexpr int LOG2 = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
	constexpr int REF = 3 * N + 4;             // entries i = -N .. 2N + 1 at [N + i], rounded up to a dword
	__shared__ __attribute__((aligned(16))) uint8_t refs[BPWG * REF];

	const uint32_t t = xcd_contiguous(blockIdx.x, gridDim.x) * kThreads + threadIdx.x;
	uint32_t blk = t / LPB;
	const int j = (int)(t % LPB);
	const bool live = blk < n_blocks;
	if (!live) blk = n_blocks - 1;
	const uint8_t *nb = neighbours + (ptrdiff_t)blk * neighbour_stride;
	const int mode = modes ? (int)modes[blk] : uniform_mode;
	uint8_t *ref = refs + (threadIdx.x / LPB) * REF;

	const int angle = kIntraAngle[mode];
	const bool vertical = mode >= 18;
	if (mode >= 2)
	{
		const int inv = angle < 0 ? (int)kIntraInvAngle[mode - 11] : 0;
		for (int e = j; e < 3 * N + 2; e += LPB)
		{
			const int i = e - N;
			int v = 0;
			if (i > 2 * N) v = 0;
			else if (i == 0) v = nb[128];
			else if (i > 0) v = vertical ? nb[63 + i] : nb[64 - i];
			else if (angle < 0)
			{
				const int k = min((i * inv + 128) >> 8, 2 * N);     // entries below (N * angle) >> 5 are never read: any in-record byte will do
				v = vertical ? nb[64 - k] : nb[63 + k];
			}
			ref[e] = (uint8_t)v;
		}
	}
	else if (mode == 0)
	{
		for (int e = j; e < 2 * N + 2; e += LPB) ref[e] = e <= N ? nb[64 + e] : nb[63 - (e - N - 1)];      // top 0..N, then left 0..N
	}
	else
	{
		for (int e = j; e < 2 * N; e += LPB) ref[e] = e < N ? nb[64 + e] : nb[63 - (e - N)];               // top 0..N-1, then left 0..N-1
	}
	__syncthreads();

	uint8_t px[16];
	if (mode >= 2)
	{
		const int corner = ref[N], first = ref[N + 1];
#pragma unroll
		for (int b = 0; b < 16; ++b)
		{
			const int o = 16 * j + b, y = o / N, x = o % N;
			const int u = vertical ? x : y, v = vertical ? y : x;
			const int tt = (v + 1) * angle, idx = tt >> 5, f = tt & 31;
			const int a = ref[N + u + idx + 1], c = ref[N + u + idx + 2];
			int val = ((32 - f) * a + f * c + 16) >> 5;
			px[b] = (uint8_t)val;
		}
		if (edge && angle == 0)
		{
			// modes 26 / 10 with the boundary smoothing: the first column / row takes half the gradient along the other edge
#pragma unroll
			for (int b = 0; b < 16; ++b)
			{
				const int o = 16 * j + b, y = o / N, x = o % N;
				const int u = vertical ? x : y, v = vertical ? y : x;
				if (u == 0)
				{
					const int other = vertical ? nb[63 - v] : nb[64 + v];
					px[b] = (uint8_t)clip3(0, 255, first + ((other - corner) >> 1));
				}
			}
		}
	}
	else if (mode == 0)
	{
		const int tn = ref[N], ln = ref[2 * N + 1];
#pragma unroll
		for (int b = 0; b < 16; ++b)
		{
			const int o = 16 * j + b, y = o / N, x = o % N;
			px[b] = (uint8_t)(((N - 1 - x) * ref[N + 1 + y] + (x + 1) * tn + (N - 1 - y) * ref[x] + (y + 1) * ln + N) >> (LOG2 + 1));
		}
	}
	else
	{
		uint32_t sum = 0;
#pragma unroll
		for (int q = 0; q < 2 * N / 4; ++q) sum = __builtin_amdgcn_sad_u8(reinterpret_cast<const uint32_t *>(ref)[q], 0, sum);
		const int dc = (int)(sum + N) >> (LOG2 + 1);
#pragma unroll
		for (int b = 0; b < 16; ++b)
		{
			const int o = 16 * j + b, y = o / N, x = o % N;
			int val = dc;
			if (edge)
			{
				if (x == 0 && y == 0) val = (ref[N] + 2 * dc + ref[0] + 2) >> 2;
				else if (y == 0) val = (ref[x] + 3 * dc + 2) >> 2;
				else if (x == 0) val = (ref[N + y] + 3 * dc + 2) >> 2;
			}
			px[b] = (uint8_t)val;
		}
	}
	Chunk<16> out;
#pragma unroll
	for (int k = 0; k < 4; ++k)
		out.d[k] = px[4 * k] | ((uint32_t)px[4 * k + 1] << 8) | ((uint32_t)px[4 * k + 2] << 16) | ((uint32_t)px[4 * k + 3] << 24);
	if (live) out.store(dst + (size_t)blk * (N * N) + 16 * j);
}

// H.265 8.4.4.2.3: the filtered neighbour record of each block (same layout; only the 4N + 1 samples are written).  One lane per output sample:
// lane s < 2N: left sample y = s; 2N <= s < 4N: top sample x = s - 2N; s == 4N: the corner.  `strong`: strong_intra_smoothing_enabled_flag -- 32x32 blocks
// whose two edges are flat take the bi-linear form, decided per block.
__global__ __launch_bounds__(kThreads) void intra_filter_neighbours_kernel(
	uint8_t *__restrict__ out, ptrdiff_t out_stride, const uint8_t *__restrict__ neighbours, ptrdiff_t neighbour_stride, int N, int strong, uint32_t n_blocks)
{
	const uint32_t per = 4 * N + 1;
	const uint64_t t = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
	const uint32_t blk = (uint32_t)(t / per), s = (uint32_t)(t % per);
	if (blk >= n_blocks) return;
	const uint8_t *nb = neighbours + (ptrdiff_t)blk * neighbour_stride;
	uint8_t *o = out + (ptrdiff_t)blk * out_stride;
	const int corner = nb[128];
	bool bi = false;
	if (strong && N == 32)
	{
		const int a = corner + nb[64 + 63] - 2 * nb[64 + 31], b = corner + nb[63 - 63] - 2 * nb[63 - 31];
		bi = abs(a) < 8 && abs(b) < 8;
	}
	if (s == 4u * N) { o[128] = bi ? (uint8_t)corner : (uint8_t)((nb[63] + 2 * corner + nb[64] + 2) >> 2); return; }
	const bool top = s >= 2u * N;
	const int i = (int)(top ? s - 2 * N : s);
	auto smp = [&](int k) { return k < 0 ? corner : (int)(top ? nb[64 + k] : nb[63 - k]); };
	int v;
	if (i == 2 * N - 1) v = smp(i);
	else if (bi) v = ((63 - i) * corner + (i + 1) * smp(63) + 32) >> 6;
	else v = (smp(i + 1) + 2 * smp(i) + smp(i - 1) + 2) >> 2;
	o[top ? 64 + i : 63 - i] = (uint8_t)v;
}

}  // namespace hevc

using namespace hevc;

static int check_intra_args(int packed, size_t n_blocks, int *log2n)
{
	*log2n = (packed >> 1) & 0x7f;
	if (*log2n < 2 || *log2n > 5) return set_error(HEVCASM_HIP_EINVAL, "block size must be 4, 8, 16 or 32");
	if ((packed & 1) && *log2n == 5) return set_error(HEVCASM_HIP_EINVAL, "the edge filter applies to blocks below 32x32");
	if (n_blocks > (0xffffffffull >> (2 * *log2n - 4)) - kThreads) return set_error(HEVCASM_HIP_EINVAL, "n_blocks too large");
	return 0;
}

static int launch_intra_modes(uint8_t *dst, const uint8_t *neighbours, ptrdiff_t neighbour_stride, const uint8_t *modes, int mode, int packed, size_t n_blocks, hipStream_t st)
{
	const int log2n = (packed >> 1) & 0x7f, edge = packed & 1, n = 1 << log2n;
	const uint32_t grid = ceil_div(n_blocks * (size_t)(n * n / 16), kThreads);
	switch (log2n)
	{
	case 2: hipLaunchKernelGGL(pred_intra_modes_kernel<4>, dim3(grid), dim3(kThreads), 0, st, dst, neighbours, neighbour_stride, modes, mode, edge, (uint32_t)n_blocks); break;
	case 3: hipLaunchKernelGGL(pred_intra_modes_kernel<8>, dim3(grid), dim3(kThreads), 0, st, dst, neighbours, neighbour_stride, modes, mode, edge, (uint32_t)n_blocks); break;
	case 4: hipLaunchKernelGGL(pred_intra_modes_kernel<16>, dim3(grid), dim3(kThreads), 0, st, dst, neighbours, neighbour_stride, modes, mode, edge, (uint32_t)n_blocks); break;
	default: hipLaunchKernelGGL(pred_intra_modes_kernel<32>, dim3(grid), dim3(kThreads), 0, st, dst, neighbours, neighbour_stride, modes, mode, edge, (uint32_t)n_blocks); break;
	}
	return launch_status("pred_intra_modes_kernel");
}

extern "C" int hevcasm_hip_pred_intra_batch(
	uint8_t *dst, const uint8_t *neighbours, ptrdiff_t neighbour_stride,
	int intraPredMode, int packed, size_t n_blocks, void *stream)
{
	if (int e = check_device()) return e;
	int log2n;
	if (intraPredMode < 0 || intraPredMode > 34) return set_error(HEVCASM_HIP_EINVAL, "intraPredMode must be 0..34");
	if (int e = check_intra_args(packed, n_blocks, &log2n)) return e;
	const int filter_edge = packed & 1;
	if (n_blocks == 0) return 0;
	if (!dst || !neighbours) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	hipStream_t st = as_stream(stream);
	if (intraPredMode != 1) return launch_intra_modes(dst, neighbours, neighbour_stride, nullptr, intraPredMode, packed, n_blocks, st);
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

extern "C" int hevcasm_hip_pred_intra_modes_batch(
	uint8_t *dst, const uint8_t *neighbours, ptrdiff_t neighbour_stride,
	const uint8_t *modes, int packed, size_t n_blocks, void *stream)
{
	if (int e = check_device()) return e;
	int log2n;
	if (int e = check_intra_args(packed, n_blocks, &log2n)) return e;
	if (n_blocks == 0) return 0;
	if (!dst || !neighbours || !modes) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	return launch_intra_modes(dst, neighbours, neighbour_stride, modes, 0, packed, n_blocks, as_stream(stream));
}

extern "C" int hevcasm_hip_intra_filter_neighbours_batch(
	uint8_t *filtered, ptrdiff_t filtered_stride, const uint8_t *neighbours, ptrdiff_t neighbour_stride,
	int log2TrafoSize, int strong_intra_smoothing, size_t n_blocks, void *stream)
{
	if (int e = check_device()) return e;
	if (log2TrafoSize < 2 || log2TrafoSize > 5) return set_error(HEVCASM_HIP_EINVAL, "block size must be 4, 8, 16 or 32");
	if (n_blocks == 0) return 0;
	if (n_blocks > 0xffffffull) return set_error(HEVCASM_HIP_EINVAL, "n_blocks too large");
	if (!filtered || !neighbours) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	const int n = 1 << log2TrafoSize;
	const uint32_t grid = ceil_div(n_blocks * (size_t)(4 * n + 1), kThreads);
	hipLaunchKernelGGL(intra_filter_neighbours_kernel, dim3(grid), dim3(kThreads), 0, as_stream(stream), filtered, filtered_stride, neighbours, neighbour_stride, n, strong_intra_smoothing,
			(uint32_t)n_blocks);
	return launch_status("intra_filter_neighbours_kernel");
}
