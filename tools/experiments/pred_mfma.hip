// EXPERIMENT, NOT BUILT INTO THE LIBRARY (round 1).  Inter prediction of 32-/64-wide PUs as two banded
// matrix products on v_mfma_i32_32x32x32_i8.  Bit-exact on every parity case when wired in behind
// launch_pred() of pred_inter.hip (W % 32 == 0), but not faster than the register-streaming VALU kernel:
//   luma uni 64x64: 2.1-2.25 TB/s algorithmic against 2.77 for the VALU kernel; luma bi 64x64: 1.9-2.3
//   against 1.9; chroma (32x32) 1.0-1.4 against 2.2.
// What the counters said: operand preparation (byte splits, accumulator initialisation and read-back,
// the output transpose) costs about 290 VALU instructions per 32x32 tile -- as many lane-ops per pixel
// as the VALU filter it replaces -- and the lane-per-row operand loads ran the vector L1 at 0.73
// accesses per clock (2.7 bytes per access); loading rows coalesced and transposing through LDS (this
// version) cured that without changing the total.  Kept as a record of the approach and its layouts
// (the accumulator-tile-as-next-operand trick is the one transform.hip's forward 32x32 kernel uses).
//
// pred_mfma.hip -- inter prediction of PUs whose width is a multiple of 32, on the matrix cores.
//
// Same functions as pred_inter.hip (hevcasm_pred_uni_8to8 / hevcasm_pred_bi_8to8, reference
// pred_inter.h:56 / :76), same exact integer results; chosen by the launcher for W = 32, 64.  The
// register-streaming kernel of pred_inter.hip is bound by the integer VALU (about 20 lane-ops per
// pixel, 0.25-0.35 of the HBM peak on large PUs).  A separable FIR filter over a 32 x 32 tile is two
// products with banded (Toeplitz) coefficient matrices, and v_mfma_i32_32x32x32_i8 has the
// throughput to do them with 3/4 of the band's zeros included:
//
//   H pass   Hres = P' Th          A operand = window rows as they lie in memory (lane = row,
//                                  16 consecutive pixels, pixel - 128 as int8), B operand = the
//                                  band matrix of this PU's horizontal fraction; 39 window columns =
//                                  two k-steps.  +128 * sum(taps) = 8192 rides in the accumulator.
//   V pass   Out = Tv Hres         Hres' accumulator tile has its column on the lane and its 16
//                                  rows in registers: a B operand that sums over the row index,
//                                  used in place (split exactly into high bytes and low bytes - 128);
//                                  the band matrix of the vertical fraction is laid out in the
//                                  accumulator's row order.  32 output rows need window rows
//                                  0 .. 38: the tile's own Hres plus the first rows of the next.
//
// With the fraction-0 filter taken as {.., 64, ..} all four of the reference's paths (copy, H, V, HV)
// are this one formula, exactly (pred_inter.hip has the argument).  One wave per 32-column strip of
// a PU, walking down its 32-row tiles; band matrices for every fraction live in LDS (built once per
// workgroup); output goes through a 1 KiB LDS tile per wave so that each lane stores 16 bytes of a
// row.  Reads exactly the reference's window, writes exactly W x H.

#include "common.hpp"

namespace hevc {

typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef int v16i32 __attribute__((ext_vector_type(16)));

struct FilterTaps { signed char luma[4][8], chroma[8][4]; };
__constant__ FilterTaps kTaps = {
	{ { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 }, { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } },
	{ { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 }, { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } } };

// row of the accumulator tile held in register j of lane half h
__host__ __device__ constexpr int acc_row(int j, int h) { return (j & 3) + 8 * (j >> 2) + 4 * h; }

// 8 int16 in 4 dwords -> their low bytes (minus 128, as int8) and high bytes, 2 dwords each
__device__ __forceinline__ void split16(const uint32_t (&w)[4], uint32_t (&lo)[2], uint32_t (&hi)[2])
{
	lo[0] = __builtin_amdgcn_perm(w[1], w[0], 0x06040200) ^ 0x80808080u;
	lo[1] = __builtin_amdgcn_perm(w[3], w[2], 0x06040200) ^ 0x80808080u;
	hi[0] = __builtin_amdgcn_perm(w[1], w[0], 0x07050301);
	hi[1] = __builtin_amdgcn_perm(w[3], w[2], 0x07050301);
}

// accumulator tile of int16-range values -> the two int8 B operands of the next product
__device__ __forceinline__ void tile_to_operands(const v16i32 &t, v4i32 &lo, v4i32 &hi)
{
	uint32_t pair[8];
#pragma unroll
	for (int i = 0; i < 16; i += 2) pair[i / 2] = __builtin_amdgcn_perm((uint32_t)t[i + 1], (uint32_t)t[i], 0x05040100);
	uint32_t l[2], g[2];
	split16({ pair[0], pair[1], pair[2], pair[3] }, l, g);
	lo.x = l[0]; lo.y = l[1]; hi.x = g[0]; hi.y = g[1];
	split16({ pair[4], pair[5], pair[6], pair[7] }, l, g);
	lo.z = l[0]; lo.w = l[1]; hi.z = g[0]; hi.w = g[1];
}

template <int TAPS, bool BI>
__global__ __launch_bounds__(kThreads) void pred_mfma_kernel(
	uint8_t *__restrict__ dst, ptrdiff_t stride_dst,
	const uint8_t *__restrict__ ref0, const uint8_t *__restrict__ ref1, ptrdiff_t stride_ref,
	const uint32_t *__restrict__ dst_off, const uint32_t *__restrict__ ref0_off, const uint32_t *__restrict__ ref1_off,
	const uint8_t *__restrict__ frac0, const uint8_t *__restrict__ frac1,
	int W, int H, uint32_t n_items /* PUs x strips */)
{
	constexpr int NREF = BI ? 2 : 1;
	constexpr int FRACS = TAPS == 8 ? 4 : 8;
	constexpr int BEFORE = TAPS / 2 - 1;
	constexpr int WAVES = kThreads / kWave;

	// band matrices, one 16-byte operand fragment per (fraction, k-step, lane):
	//   th: H pass B operand, lane (x = l & 31, h): byte j <-> window column k = 32 ks + 16 h + j, value tap[k - x]
	//   tv: V pass A operand, lane (y = l & 31, h): byte j <-> window row 32 ks + acc_row(j, h), value tap[row - y]
	__shared__ __attribute__((aligned(16))) uint8_t th[FRACS][2][kWave][16];
	__shared__ __attribute__((aligned(16))) uint8_t tv[FRACS][2][kWave][16];
	__shared__ __attribute__((aligned(16))) uint8_t out_tile[WAVES][32][32];
	constexpr int kStagePitch = 48;   // bytes per staged window row (32 + TAPS - 1 <= 39 used)
	__shared__ __attribute__((aligned(16))) uint8_t stage[WAVES][32 * kStagePitch];

	for (int idx = threadIdx.x; idx < FRACS * 2 * kWave * 16; idx += kThreads)
	{
		const int j = idx & 15, l = (idx >> 4) & 63, ks = (idx >> 10) & 1, f = idx >> 11;
		const int r = l & 31, h = l >> 5;
		const int tap_h = 32 * ks + 16 * h + j - r;
		const int tap_v = 32 * ks + acc_row(j, h) - r;
		const signed char *taps = TAPS == 8 ? kTaps.luma[f] : kTaps.chroma[f];
		th[f][ks][l][j] = (uint8_t)(tap_h >= 0 && tap_h < TAPS ? taps[tap_h] : 0);
		tv[f][ks][l][j] = (uint8_t)(tap_v >= 0 && tap_v < TAPS ? taps[tap_v] : 0);
	}
	__syncthreads();

	const int wave_in_wg = threadIdx.x >> 6;
	const int l = threadIdx.x & 63, r = l & 31, h = l >> 5;
	const int strips = W / 32;
	const int window_rows = H + TAPS - 1;
	const int mtiles = (window_rows + 31) / 32, ntiles = (H + 31) / 32;

	// what a wave needs to know about an item (a 32-column strip of a PU); everything here is wave-uniform
	struct Item
	{
		const uint8_t *base[NREF];   // window origin: TAPS / 2 - 1 rows above and columns left of the strip's first sample
		int frac[NREF];
		uint8_t *dp;
	};
	auto describe = [&](uint32_t item) -> Item
	{
		const uint32_t pu = item / strips;
		const int strip = (int)(item - pu * strips);
		Item d;
#pragma unroll
		for (int q = 0; q < NREF; ++q)
		{
			d.base[q] = (q ? ref1 : ref0) + (q ? ref1_off : ref0_off)[pu] + 32 * strip - BEFORE - (ptrdiff_t)BEFORE * stride_ref;
			d.frac[q] = (q ? frac1 : frac0)[pu];
		}
		d.dp = dst + dst_off[pu] + 32 * strip;
		return d;
	};
	// One 32-row tile of the window: 32 rows x (32 + TAPS - 1) bytes.  Loaded with consecutive lanes along a
	// row (4 bytes each; the last dword of a row starts 3 bytes early so that nothing right of the window
	// is touched) -- a lane-per-row load touches 32 cache lines per instruction and pinned the first
	// version of this kernel to the vector L1's access rate (2.7 bytes per access).  Rows past the window
	// repeat its last row: only output rows >= H, which are never stored, depend on them.
	constexpr int DPR = (32 + TAPS - 1 + 3) / 4;              // dwords per window row: 10 (luma) / 9 (chroma)
	constexpr int NLD = (32 * DPR + kWave - 1) / kWave;       // load instructions per tile: 5
	constexpr int LAST_AT = 32 + TAPS - 1 - 4;                // byte offset of a row's last dword: 35 / 31
	struct Raw { uint32_t d[NLD]; };
	auto issue = [&](const Item &it, int mt, Raw (&raw)[NREF])
	{
#pragma unroll
		for (int k = 0; k < NLD; ++k)
		{
			const int idx = min(l + kWave * k, 32 * DPR - 1);
			const int row = idx / DPR, j = idx - row * DPR;
			const int wr = min(32 * mt + row, window_rows - 1);
			const int at = j < DPR - 1 ? 4 * j : LAST_AT;
#pragma unroll
			for (int q = 0; q < NREF; ++q)
				raw[q].d[k] = reinterpret_cast<const U1 *>(it.base[q] + (ptrdiff_t)wr * stride_ref + at)->v;   // (not touched until consumed)
		}
	};
	// H pass of a loaded tile: through the wave's LDS stage into operand layout (lane = row, 16 consecutive
	// pixels), pixel - 128 as int8, two k-steps; + 128 * sum(taps) undoes the offset
	auto h_mfma = [&](const Raw &raw, const v4i32 (&bh)[2]) -> v16i32
	{
		uint8_t *st = stage[wave_in_wg];
#pragma unroll
		for (int k = 0; k < NLD; ++k)
		{
			const int idx = l + kWave * k;
			const int row = idx / DPR, j = idx - row * DPR;
			if (idx < 32 * DPR)
			{
				const bool lastdw = j == DPR - 1;
				// the row's last dword was loaded 3 bytes early: keep its upper three bytes, at the next aligned slot
				*reinterpret_cast<uint32_t *>(st + row * kStagePitch + (lastdw ? LAST_AT + 1 : 4 * j)) = lastdw ? raw.d[k] >> 8 : raw.d[k];
			}
		}
		const u32x4 v0 = *reinterpret_cast<const u32x4 *>(st + r * kStagePitch + 16 * h);
		const u32x2 v1 = *reinterpret_cast<const u32x2 *>(st + r * kStagePitch + 32);
		const v4i32 a0 = { (int)(v0.x ^ 0x80808080u), (int)(v0.y ^ 0x80808080u), (int)(v0.z ^ 0x80808080u), (int)(v0.w ^ 0x80808080u) };
		const v4i32 a1 = { (int)(v1.x ^ 0x80808080u), (int)(v1.y ^ 0x80808080u), 0, 0 };   // columns >= 32 + TAPS - 1 meet zeros of the band
		v16i32 acc;
#pragma unroll
		for (int i = 0; i < 16; ++i) acc[i] = 128 * 64;
		acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, bh[0], acc, 0, 0, 0);
		acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, bh[1], acc, 0, 0, 0);
		return acc;
	};

	// Software pipeline: the loads of the window tile that will be needed NEXT (the following tile of this
	// strip, or the first tile of the wave's next strip) are in flight while the current tile is filtered.
	const uint32_t item_step = gridDim.x * WAVES;
	uint32_t item = blockIdx.x * WAVES + wave_in_wg;
	if (item >= n_items) return;
	Item cur_it = describe(item);
	Raw pend[NREF];
	issue(cur_it, 0, pend);

	for (; item < n_items; item += item_step)
	{
		const bool more_items = item + item_step < n_items;
		Item next_it = cur_it;
		if (more_items) next_it = describe(item + item_step);

		v4i32 bh[NREF][2], av[NREF][2];
#pragma unroll
		for (int q = 0; q < NREF; ++q)
		{
			const int xf = cur_it.frac[q] & (FRACS - 1), yf = (cur_it.frac[q] >> 4) & (FRACS - 1);
#pragma unroll
			for (int ks = 0; ks < 2; ++ks)
			{
				bh[q][ks] = *reinterpret_cast<const v4i32 *>(th[xf][ks][l]);
				av[q][ks] = *reinterpret_cast<const v4i32 *>(tv[yf][ks][l]);
			}
		}

		v16i32 cur[NREF];
#pragma unroll
		for (int q = 0; q < NREF; ++q) cur[q] = h_mfma(pend[q], bh[q]);
		if (mtiles > 1) issue(cur_it, 1, pend);
		else if (more_items) issue(next_it, 0, pend);

		for (int t = 0; t < ntiles; ++t)
		{
			const bool has_next = t + 1 < mtiles;   // wave-uniform
			v16i32 nxt[NREF];
#pragma unroll
			for (int q = 0; q < NREF; ++q) nxt[q] = cur[q];
			if (has_next)
			{
#pragma unroll
				for (int q = 0; q < NREF; ++q) nxt[q] = h_mfma(pend[q], bh[q]);
				if (t + 2 < mtiles) issue(cur_it, t + 2, pend);
				else if (more_items) issue(next_it, 0, pend);
			}

			int sum[NREF][16];
#pragma unroll
			for (int q = 0; q < NREF; ++q)
			{
				v4i32 lo0, hi0, lo1, hi1;
				tile_to_operands(cur[q], lo0, hi0);
				v16i32 o_lo, o_hi = { 0 };
#pragma unroll
				for (int i = 0; i < 16; ++i) o_lo[i] = 128 * 64 + (BI ? 0 : 2048);   // low-byte offset, and the uni rounding constant
				o_lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[q][0], lo0, o_lo, 0, 0, 0);
				o_hi = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[q][0], hi0, o_hi, 0, 0, 0);
				if (has_next)
				{
					// (the 128 * 64 in the accumulator covers both k-steps: every stored output row meets each tap
					// exactly once across them; without a next tile all of a stored row's taps lie in the first)
					tile_to_operands(nxt[q], lo1, hi1);
					o_lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[q][1], lo1, o_lo, 0, 0, 0);
					o_hi = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[q][1], hi1, o_hi, 0, 0, 0);
				}
#pragma unroll
				for (int i = 0; i < 16; ++i) sum[q][i] = (o_hi[i] << 8) + o_lo[i];
				cur[q] = nxt[q];
			}
			// exact 2-D sums -> pixels (uni: rounding already inside; bi: pred_inter.c:490-530)
			uint8_t (*tile)[32] = out_tile[wave_in_wg];
#pragma unroll
			for (int i = 0; i < 16; i += 2)
			{
				int v0, v1;
				if (BI)
				{
					v0 = (sum[0][i] >> 6) + (sum[NREF - 1][i] >> 6) + 64;
					v1 = (sum[0][i + 1] >> 6) + (sum[NREF - 1][i + 1] >> 6) + 64;
				}
				else
				{
					v0 = sum[0][i];
					v1 = sum[0][i + 1];
				}
				const uint32_t two = __builtin_amdgcn_ashr_pk_u8_i32(v0, v1, BI ? 7 : 12);
				tile[acc_row(i, h)][r] = (uint8_t)two;
				tile[acc_row(i + 1, h)][r] = (uint8_t)(two >> 8);
			}
			// the wave's own tile: LDS operations of one wave complete in order, no barrier needed
			const int row = l >> 1, y = 32 * t + row;
			Chunk<16> px;
			const u32x4 v = *reinterpret_cast<const u32x4 *>(&tile[row][16 * (l & 1)]);
			px.d[0] = v.x; px.d[1] = v.y; px.d[2] = v.z; px.d[3] = v.w;
			if (y < H) px.store(cur_it.dp + (ptrdiff_t)y * stride_dst + 16 * (l & 1));
		}
		cur_it = next_it;
	}
}

template <int TAPS, bool BI>
static int launch_pred_mfma_t(uint8_t *dst, ptrdiff_t sd, const uint8_t *ref0, const uint8_t *ref1, ptrdiff_t sr,
		const uint32_t *dof, const uint32_t *r0, const uint32_t *r1, const uint8_t *f0, const uint8_t *f1,
		int W, int H, size_t n, hipStream_t st)
{
	const size_t items = n * (size_t)(W / 32);
	if (items > 0xffffffffull) return set_error(HEVCASM_HIP_EINVAL, "n_blocks too large");
	// every workgroup builds the band matrices once, then its waves walk the items: exactly as many
	// workgroups as are resident at once, so that no second round of workgroups trails behind
	static const int resident = []
	{
		int per_cu = 0, dev = 0;
		hipDeviceProp_t prop;
		if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256 * 2;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pred_mfma_kernel<TAPS, BI>, kThreads, 0) != hipSuccess || per_cu < 1) per_cu = 1;
		return prop.multiProcessorCount * per_cu;
	}();
	const size_t wanted = (items + kThreads / kWave - 1) / (kThreads / kWave);
	const uint32_t grid = (uint32_t)(wanted < (size_t)resident ? wanted : (size_t)resident);
	hipLaunchKernelGGL((pred_mfma_kernel<TAPS, BI>), dim3(grid), dim3(kThreads), 0, st,
			dst, sd, ref0, ref1, sr, dof, r0, r1, f0, f1, W, H, (uint32_t)items);
	return launch_status("pred_mfma_kernel");
}

int launch_pred_mfma(int taps, bool bi, uint8_t *dst, ptrdiff_t sd, const uint8_t *ref0, const uint8_t *ref1, ptrdiff_t sr,
		const uint32_t *dof, const uint32_t *r0, const uint32_t *r1, const uint8_t *f0, const uint8_t *f1,
		int W, int H, size_t n, hipStream_t st)
{
	if (taps == 8) return bi ? launch_pred_mfma_t<8, true>(dst, sd, ref0, ref1, sr, dof, r0, r1, f0, f1, W, H, n, st)
	                         : launch_pred_mfma_t<8, false>(dst, sd, ref0, ref1, sr, dof, r0, r1, f0, f1, W, H, n, st);
	return bi ? launch_pred_mfma_t<4, true>(dst, sd, ref0, ref1, sr, dof, r0, r1, f0, f1, W, H, n, st)
	          : launch_pred_mfma_t<4, false>(dst, sd, ref0, ref1, sr, dof, r0, r1, f0, f1, W, H, n, st);
}

}  // namespace hevc
