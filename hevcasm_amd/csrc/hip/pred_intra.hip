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
#include "intra_predict.hpp"

namespace hevc {

// A lane's 16 output bytes are one global round trip (its share of the 2N neighbour bytes) and one store: a workgroup that does this once is all latency
// (eight of them per CU keep 3.4 TB/s of stores going, a fill kernel reaches 5-7).  So a workgroup takes DC_UNROLL consecutive groups of 256 lanes and
// issues every group's loads before the first group's arithmetic.  Same-box A/B at 2^28 pixels (profiles/r05_intra_dc_unroll_ab.txt): four groups against
// one: 32x32 -20 %, 16x16 -4 ... -7 %, 8x8 +0.5 ... +7 %, 4x4 -3 ... +7 % (the small blocks are bound by their records); eight groups: no better anywhere.
// So: four groups from 16x16 up, one below.
template <int N> struct DcUnroll { static constexpr int K = N >= 16 ? 4 : 1; };

template <int N>
__global__ __launch_bounds__(kThreads) void pred_intra_dc_kernel(
	uint8_t *__restrict__ dst, const uint8_t *__restrict__ neighbours, ptrdiff_t neighbour_stride,
	int filter_edge, uint32_t n_blocks)
{
	constexpr int LPB = N * N / 16;            // lanes per block: 1, 4, 16, 64
	constexpr int LOG2 = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
	constexpr int K = DcUnroll<N>::K;
	constexpr int LOADERS = N == 4 ? 1 : 2 * N / 16;
	constexpr int ROWS = N >= 16 ? 1 : 16 / N, COLS = N >= 16 ? 16 : N;
	const uint32_t t0 = xcd_contiguous(blockIdx.x, gridDim.x) * (kThreads * K) + threadIdx.x;

	// phase 1: every group's loads (no lane is switched off: a lane past the batch repeats the last block, a lane that is not one of its block's loaders
	// repeats a loader's chunk -- the same lines, no branch between the loads)
	Chunk<N == 4 ? 8 : 16> sums[K];
	Chunk<COLS> tops[K];
	uint32_t lefts[K][ROWS], corner_l[K], corner_t[K];
#pragma unroll
	for (int k = 0; k < K; ++k)
	{
		const uint32_t t = t0 + (uint32_t)k * kThreads;
		uint32_t blk = t / LPB;
		const uint32_t j = t % LPB;
		if (blk >= n_blocks) blk = n_blocks - 1;
		const uint8_t *nb = neighbours + (ptrdiff_t)blk * neighbour_stride;
		if (N == 4) sums[k].load(nb + 60);
		else sums[k].load(nb + 64 - N + 16 * (j % LOADERS));
		if (filter_edge)
		{
			const int y0 = 16 * (int)j / N, x0 = 16 * (int)j % N;
			tops[k].load(nb + 64 + x0);                       // samples (x0 .. x0 + COLS - 1, -1): used by the lanes of row 0
#pragma unroll
			for (int r = 0; r < ROWS; ++r) lefts[k][r] = nb[63 - (y0 + r)];
			corner_l[k] = nb[63];
			corner_t[k] = nb[64];
		}
	}
	// phase 2: per group, the DC value and the 16 bytes
#pragma unroll
	for (int k = 0; k < K; ++k)
	{
		const uint32_t t = t0 + (uint32_t)k * kThreads;
		const uint32_t blk = t / LPB, j = t % LPB;
		const bool live = blk < n_blocks;
		// DC value, pred_intra.c:66-76: (N + sum of the 2N samples) >> (log2 N + 1)
		uint32_t sum = 0;
		if (N == 4) sum = __builtin_amdgcn_sad_u8(sums[k].d[0], 0, __builtin_amdgcn_sad_u8(sums[k].d[1], 0, 0));
		else
		{
			if ((int)j < LOADERS)
			{
#pragma unroll
				for (int q = 0; q < 4; ++q) sum = __builtin_amdgcn_sad_u8(sums[k].d[q], 0, sum);
			}
#pragma unroll
			for (int d = LOADERS / 2; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, kWave);
			sum = __shfl(sum, (int)((threadIdx.x & 63) - j), kWave);      // the block's first lane holds the total
		}
		const int dc = (int)(sum + N) >> (LOG2 + 1);

		Chunk<16> out;
#pragma unroll
		for (int q = 0; q < 4; ++q) out.d[q] = (uint32_t)dc * 0x01010101u;
		if (filter_edge)
		{
			// pred_intra.c:80-93: first row and first column blend the neighbours into the DC value.
			// A lane owns 16 consecutive output bytes = ROWS rows of min(N, 16) columns.
			const int y0 = 16 * (int)j / N, x0 = 16 * (int)j % N;
			uint8_t px[16];
#pragma unroll
			for (int b = 0; b < 16; ++b) px[b] = (uint8_t)dc;
			if (y0 == 0)
			{
#pragma unroll
				for (int b = 0; b < COLS; ++b) px[b] = (uint8_t)((((tops[k].d[b / 4] >> (8 * (b % 4))) & 0xff) + 3 * dc + 2) >> 2);
			}
			if (x0 == 0)
			{
#pragma unroll
				for (int r = 0; r < ROWS; ++r)
					if (y0 + r > 0) px[r * COLS] = (uint8_t)((lefts[k][r] + 3 * dc + 2) >> 2);
				if (y0 == 0) px[0] = (uint8_t)((corner_l[k] + 2 * dc + corner_t[k] + 2) >> 2);
			}
#pragma unroll
			for (int q = 0; q < 4; ++q)
				out.d[q] = px[4 * q] | ((uint32_t)px[4 * q + 1] << 8) | ((uint32_t)px[4 * q + 2] << 16) | ((uint32_t)px[4 * q + 3] << 24);
		}
		if (live) out.store(dst + (size_t)blk * (N * N) + 16 * j);
	}
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
// Second form (round 5).  Everything angular is computed in FAMILY space: a horizontal mode m is the transpose of the vertical mode 36 - m on swapped
// edges, so the staging step puts the edge the mode runs along ("main": top for modes 18..34, left for 2..17) into ref[] and the kernel always filters
// along it; a lane owns 16 family-space bytes = R rows of L = min(N, 16) pixels, each row one (iIdx, iFact): L + 1 consecutive ref bytes come out of LDS
// as aligned dwords + v_alignbyte, and the blend ((32 - f) a + f b + 16) >> 5 runs on even / odd bytes with v_pk_mad_u16 (11 instructions per 4 pixels;
// f = 0 gives a exactly, so no branch).  Vertical modes store the 16 bytes as they are; horizontal modes scatter them transposed into an LDS tile and
// read their row-major 16 bytes back.  Planar and DC stay per pixel (few operations each).
template <int N>
__global__ __launch_bounds__(kThreads) void pred_intra_modes_kernel(
	uint8_t *__restrict__ dst, const uint8_t *__restrict__ neighbours, ptrdiff_t neighbour_stride,
	const uint8_t *__restrict__ modes, int uniform_mode, int edge, uint32_t n_blocks, uint32_t iterations, int corner_at /* 128, or 64 + 2N for compact records */)
{
	constexpr int PX = N < 16 ? N : 16;        // output bytes per lane = one family row (4x4, 8x8: a whole row; 16: a row; 32: half a row)
	constexpr int LPB = N * N / PX;            // lanes per block: 4, 8, 16, 64.  (A lane per 16 bytes at 4x4 / 8x8 -- 1 and 4 lanes per block -- measured the
	                                           // same: small blocks are bound by their RECORDS, >= 129 bytes fetched per 16 / 64 bytes written: 4x4 moves
	                                           // 5.0 TB/s of real traffic at "0.13" of algorithmic bytes, profiles/r05_intra_modes.txt)
	constexpr int BPWG = kThreads / LPB;
	constexpr int REF = 3 * N + 12;            // entries i = -N .. 2N + 1 at [N + i]; the dword reads of a row may run 8 bytes past the last entry
	constexpr int RAW = 4 * N + 4;
	constexpr int LOADS = (N + LPB - 1) / LPB; // record dwords per lane: 1 (the first 32 lanes at 32x32)
	__shared__ __attribute__((aligned(16))) uint8_t refs[BPWG * REF];
	__shared__ __attribute__((aligned(16))) uint8_t tiles[kThreads * PX];
	__shared__ __attribute__((aligned(16))) uint8_t raws[BPWG * RAW];

	// A workgroup takes `iterations` consecutive groups of BPWG blocks.  A group's life is one global round trip (its records) and three barriers for
	// 1 - 4 KiB of output: alone that is latency, not bandwidth (0.3 of the peak with eight workgroups per CU), so the NEXT group's records and modes are
	// requested before this group's barriers and land while it computes.
	// A block's lanes all sit in ONE wave (4, 8, 16 or 64 of them) and its LDS areas are its own, so the three hand-overs of an iteration (record -> edge
	// arrays -> rows -> transposed tile) need the wave's own LDS accesses in order, not a workgroup barrier: the four waves run free of each other.
	auto block_sync = []()
	{
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	};
	const uint32_t first_group = xcd_contiguous(blockIdx.x, gridDim.x) * iterations;
	const int j = (int)(threadIdx.x % LPB);
	uint8_t *ref = refs + (threadIdx.x / LPB) * REF;
	uint8_t *tile = tiles + (threadIdx.x / LPB) * (N * N);
	uint8_t *raw = raws + (threadIdx.x / LPB) * RAW;

	uint32_t rec[LOADS], rec_corner = 0;
	int rec_mode = uniform_mode;
	auto request = [&](uint32_t group)
	{
		uint32_t b = group * BPWG + threadIdx.x / LPB;
		if (b >= n_blocks) b = n_blocks - 1;
		const uint8_t *nb = neighbours + (ptrdiff_t)b * neighbour_stride;
#pragma unroll
		for (int q = 0; q < LOADS; ++q)
			if (j + q * LPB < N) rec[q] = reinterpret_cast<const U1 *>(nb + 64 - 2 * N + 4 * (j + q * LPB))->v;
		if (j == 0) rec_corner = nb[corner_at];
		if (modes) rec_mode = min((int)modes[b], 34);      // a byte above 34 is not a mode: clamped, so that the tables below are never indexed past their ends (the header calls the result unspecified)
	};
	request(first_group);

	for (uint32_t it = 0; it < iterations; ++it)
	{
	const uint32_t blk = (first_group + it) * BPWG + threadIdx.x / LPB;
	const bool live = blk < n_blocks;
	const int mode = rec_mode;      // (as a scalar where the wave's blocks share it -- one mode per batch, 32x32: measured level or slower, not kept)
#pragma unroll
	for (int q = 0; q < LOADS; ++q)
		if (j + q * LPB < N) reinterpret_cast<uint32_t *>(raw)[j + q * LPB] = rec[q];
	if (j == 0) raw[4 * N] = (uint8_t)rec_corner;
	if (it + 1 < iterations) request(first_group + it + 1);
	block_sync();

	Chunk<PX> out;
	intra_predict_block<N, LPB>(raw, ref, tile, j, mode, edge, block_sync, out);
	if (live) out.store(dst + (size_t)blk * (N * N) + PX * j);
	}
}

// H.265 8.4.4.2.3: the filtered neighbour record of each block (same layout; only the 4N + 1 samples are written).  In memory the samples are ONE run --
// left column upwards [64 - 2N, 64), then the top row [64, 64 + 2N) -- with the corner (stored at [128]) logically between bytes 63 and 64, and the
// [1 2 1] filter is symmetric, so a lane filters one DWORD of that run from itself and its two neighbours (v_alignbyte by one byte each way, the corner
// patched in at the junction, even / odd bytes on packed 16-bit adds); lane N of a block writes the corner.  `strong`: strong_intra_smoothing_enabled_flag
// -- 32x32 blocks whose two edges are flat take the bi-linear form, decided per block.  (First form, a lane per byte: 0.04-0.07 of the HBM peak.)
template <int N>      // (a template since round 5's second look: the lane -> (block, dword) split divides by N + 1, which as a runtime 64-bit division cost more than the filter)
__global__ __launch_bounds__(kThreads) void intra_filter_neighbours_kernel(
	uint8_t *__restrict__ out, ptrdiff_t out_stride, const uint8_t *__restrict__ neighbours, ptrdiff_t neighbour_stride, int strong, uint32_t n_blocks, int corner_at)
{
	constexpr uint32_t per = (uint32_t)N + 1;
	const uint32_t t = xcd_contiguous(blockIdx.x, gridDim.x) * kThreads + threadIdx.x;      // n_blocks * (N + 1) < 2^32: checked by the entry point
	const uint32_t blk = t / per, s = t % per;
	if (blk >= n_blocks) return;
	const uint8_t *nb = neighbours + (ptrdiff_t)blk * neighbour_stride;
	uint8_t *o = out + (ptrdiff_t)blk * out_stride;
	const int corner = nb[corner_at];
	bool bi = false;
	if (strong && N == 32)
	{
		const int a = corner + nb[64 + 63] - 2 * nb[64 + 31], b = corner + nb[63 - 63] - 2 * nb[63 - 31];
		bi = abs(a) < 8 && abs(b) < 8;
	}
	if (s == (uint32_t)N) { o[corner_at] = bi ? (uint8_t)corner : (uint8_t)((nb[63] + 2 * corner + nb[64] + 2) >> 2); return; }
	const int m = 64 - 2 * N + 4 * (int)s;                      // the dword's first byte in the record
	const uint32_t cur = reinterpret_cast<const U1 *>(nb + m)->v;
	uint32_t res;
	if (bi)
	{
		// sample i of an edge: ((63 - i) corner + (i + 1) end + 32) >> 6, the end sample itself kept; left sample i lies at byte 63 - i, top sample i at 64 + i
		const bool top = m >= 64;
		const int end = top ? nb[127] : nb[0];
		res = 0;
#pragma unroll
		for (int b = 0; b < 4; ++b)
		{
			const int i = top ? m + b - 64 : 63 - (m + b);
			const int v = i == 63 ? end : ((63 - i) * corner + (i + 1) * end + 32) >> 6;
			res |= (uint32_t)v << (8 * b);
		}
	}
	else
	{
		const uint32_t prev = s > 0 ? reinterpret_cast<const U1 *>(nb + m - 4)->v : 0u;
		const uint32_t next = s + 1 < (uint32_t)N ? reinterpret_cast<const U1 *>(nb + m + 4)->v : 0u;
		uint32_t before = __builtin_amdgcn_alignbyte(cur, prev, 3);      // the bytes one position below each of cur's
		uint32_t after = __builtin_amdgcn_alignbyte(next, cur, 1);       // ... and one above
		if (m == 64) before = (before & 0xffffff00u) | (uint32_t)corner;           // below the first top sample: the corner
		if (m == 60) after = (after & 0x00ffffffu) | ((uint32_t)corner << 24);     // above the first left sample (byte 63): the corner
		const uint32_t even = (before & 0x00ff00ffu) + 2 * (cur & 0x00ff00ffu) + (after & 0x00ff00ffu) + 0x00020002u;
		const uint32_t odd = ((before >> 8) & 0x00ff00ffu) + 2 * ((cur >> 8) & 0x00ff00ffu) + ((after >> 8) & 0x00ff00ffu) + 0x00020002u;
		res = ((even >> 2) & 0x00ff00ffu) | (((odd >> 2) & 0x00ff00ffu) << 8);
		if (s == 0) res = (res & 0xffffff00u) | (cur & 0xffu);                     // the two end samples are kept
		if (s + 1 == (uint32_t)N) res = (res & 0x00ffffffu) | (cur & 0xff000000u);
	}
	reinterpret_cast<U1 *>(o + m)->v = res;
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
	// groups of 256 lanes (a lane = one row of a 4x4 / 8x8 block, 16 bytes of a larger one); a workgroup takes up to 8 groups in a row, fewer while that
	// would leave the device under ~8 workgroups per CU
	const uint32_t groups = ceil_div(n_blocks * (size_t)(n < 16 ? n : n * n / 16), kThreads);
	uint32_t iterations = 8;
	while (iterations > 1 && groups / iterations < (uint32_t)device_cus() * 8) iterations >>= 1;
	const uint32_t grid = ceil_div(groups, iterations);
	const int corner_at = (packed & HEVCASM_HIP_INTRA_COMPACT) ? 64 + 2 * n : 128;
#define HEVC_INTRA(NN) hipLaunchKernelGGL((pred_intra_modes_kernel<NN>), dim3(grid), dim3(kThreads), 0, st, dst, neighbours, neighbour_stride, modes, mode, edge, (uint32_t)n_blocks, iterations, corner_at)
	switch (log2n)
	{
	case 2: HEVC_INTRA(4); break;
	case 3: HEVC_INTRA(8); break;
	case 4: HEVC_INTRA(16); break;
	default: HEVC_INTRA(32); break;
	}
#undef HEVC_INTRA
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
	const uint32_t grid = ceil_div(n_blocks * (size_t)(n * n / 16), (size_t)kThreads * (n >= 16 ? DcUnroll<16>::K : DcUnroll<4>::K));
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
	const bool compact = (log2TrafoSize & HEVCASM_HIP_INTRA_COMPACT) != 0;      // the same flag as in `packed`, on the size argument here
	log2TrafoSize &= ~HEVCASM_HIP_INTRA_COMPACT;
	if (log2TrafoSize < 2 || log2TrafoSize > 5) return set_error(HEVCASM_HIP_EINVAL, "block size must be 4, 8, 16 or 32");
	if (n_blocks == 0) return 0;
	if (!filtered || !neighbours) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	const int n = 1 << log2TrafoSize;
	{
		// a record is written at bytes [64 - 2N, 64 + 2N) and the corner byte: records closer than that overwrite each other
		const ptrdiff_t need = compact ? 4 * n + 1 : 129, have = filtered_stride < 0 ? -filtered_stride : filtered_stride;
		if (have < need) return set_error(HEVCASM_HIP_EINVAL, "filtered_stride too small: at least 129, or 4N + 1 for compact records");
	}
	if (n_blocks > (0xffffffffull - kThreads) / (size_t)(n + 1)) return set_error(HEVCASM_HIP_EINVAL, "n_blocks too large for one launch");
	const uint32_t grid = ceil_div(n_blocks * (size_t)(n + 1), kThreads);
	const int corner_at = compact ? 64 + 2 * n : 128;
#define HEVC_FILTER(NN) hipLaunchKernelGGL(intra_filter_neighbours_kernel<NN>, dim3(grid), dim3(kThreads), 0, as_stream(stream), filtered, filtered_stride, neighbours, neighbour_stride, \
	strong_intra_smoothing, (uint32_t)n_blocks, corner_at)
	switch (n)
	{
	case 4: HEVC_FILTER(4); break;
	case 8: HEVC_FILTER(8); break;
	case 16: HEVC_FILTER(16); break;
	default: HEVC_FILTER(32); break;
	}
#undef HEVC_FILTER
	return launch_status("intra_filter_neighbours_kernel");
}
