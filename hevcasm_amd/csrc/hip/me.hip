// me.hip -- full-search motion estimation: hevcasm_sad evaluated at every displacement of a
// +-range window, arg-min per block.
//
// One workgroup per block.  The source block and the reference window are staged once in LDS
// (64x64, R = 64: 4 KiB + 39 KiB).  Work item = (dy, group of 16 consecutive dx): for each block
// row the thread reads W + 16 window bytes with aligned 16-byte LDS reads and feeds
// v_qsad_pk_u16_u8, which slides a 4-pixel source dword over an 8-byte reference window and
// accumulates FOUR neighbouring candidates per instruction (16 byte differences per lane-op, no
// byte realignment: the groups start on dword boundaries of the window).  The packed u16 sums are
// folded into 32-bit accumulators every 4 rows (4 rows x 16 dwords x 1020 < 65536).
// Consecutive threads take consecutive groups: consecutive 16-byte LDS addresses, conflict free.
// The arg-min key is (cost << 16) | candidate index, so ties go to the first candidate in raster
// order of (dy, dx), the order of the nested loops a CPU full search uses.
// VALU/LDS-bound (about 1.7 k byte differences per byte fetched), not HBM-bound.

#include "common.hpp"
#include <algorithm>
#include <cstdlib>
#include <vector>

namespace hevc {

// Where the blocks of a launch lie.  LIST: block i = base + off[i], all W x H (the batch entry point).  FRAME: the launch
// is rows of a CTU grid -- block i is CTU (i / cols, i % cols) of a slab whose first pixel both planes point at, CTUs
// cut off by the frame's right / bottom edge are narrower / shorter (a 2160-row frame ends in a row of 64x48 CTUs) --
// one launch for a whole frame or for one rank's rows of it, no descriptor arrays at all.
struct MeFrame
{
	int cols, ctu, frame_w, slab_h, first;      // first: the launch's first CTU, counted in raster order from column 0 of the slab's first CTU row
	// round 6: gridDim.y slabs of the same geometry (consecutive frames, one rank's rows of each) in ONE launch: slab f lies f * pitch bytes further on in both
	// planes and owns results [f * per_frame, (f + 1) * per_frame)
	ptrdiff_t pitch_src, pitch_ref;
	uint32_t per_frame;
};

// the 8-byte-aligned 8 bytes inside block k's 12-byte result slot: its first 8 bytes or its last 8, whichever the ADDRESS makes aligned
// (result itself need only be int-aligned)
__device__ __forceinline__ unsigned long long *me_key_slot(int *result, uint32_t k)
{
	int *slot = result + 3 * (size_t)k;
	return reinterpret_cast<unsigned long long *>(slot + ((reinterpret_cast<uintptr_t>(slot) >> 2) & 1u));
}

// W64: every block of the launch is 64 wide (the common case: CTUs) -- the row loops then have constant trip counts and unroll.
// parts (1 .. 8): workgroups per block.  A block's candidates cost 1.15 M SIMD-cycles of v_qsad, i.e. a whole CU for 119 us, however
// many threads work on it; a launch of fewer blocks than the chip has CUs x 4 (one rank's rows of a frame split over 8 GPUs: 240-300 CTUs)
// leaves CUs idle or gives some of them two blocks in a row.  Such launches cut every block's candidate ROWS into `parts` ranges, one
// workgroup each (its window is the H + rows - 1 window rows the range needs); the ranges' minima meet in an atomicMin on the 64-bit key
// (cost << 16 | index: ties still go to the raster-first candidate) held in the block's own 12-byte result slot, and me_decode_kernel
// rewrites the keys as (dx, dy, cost) afterwards.
// `wg`: the workgroup's index among the launch's (blocks x parts) of this geometry
template <bool FRAME, bool W64>
__device__ __forceinline__ void me_search_workgroup(
	const uint8_t *__restrict__ src, ptrdiff_t stride_src, const uint8_t *__restrict__ ref, ptrdiff_t stride_ref,
	const uint32_t *__restrict__ src_off, const uint32_t *__restrict__ ref_off, int *__restrict__ result,
	int W, int H, int R, int win_stride /* bytes, multiple of 16 */, const MeFrame &fr, int parts, uint32_t wg, uint8_t *lds)
{
	const uint32_t blk = wg / (uint32_t)parts;
	const int part = (int)(wg - blk * (uint32_t)parts);
	const uint8_t *sp, *rp0;
	if (FRAME)
	{
		src += (ptrdiff_t)blockIdx.y * fr.pitch_src;
		ref += (ptrdiff_t)blockIdx.y * fr.pitch_ref;
		result += 3 * (size_t)blockIdx.y * fr.per_frame;
		const int g = fr.first + (int)blk;
		const int row = g / fr.cols, col = g - row * fr.cols;
		W = W64 ? 64 : min(fr.ctu, fr.frame_w - col * fr.ctu);
		H = min(fr.ctu, fr.slab_h - row * fr.ctu);
		sp = src + (ptrdiff_t)row * fr.ctu * stride_src + col * fr.ctu;
		rp0 = ref + (ptrdiff_t)row * fr.ctu * stride_ref + col * fr.ctu;
	}
	else
	{
		sp = src + src_off[blk];
		rp0 = ref + ref_off[blk];
	}
	// Window first, source tile behind it.  Rows are exactly as wide as the window (rounded to 16): the last
	// dx group reads up to 16 * n_groups + W + 4 bytes of a row, i.e. into the next row -- and from the last
	// row into the source tile -- but only for candidates past the window, which are never scored.  For 64x64
	// blocks and +-64 that makes 36 864 + 4 096 = 40 960 bytes: FOUR workgroups per CU (160 KiB), and a 4K
	// frame's 2040 CTUs are then two full rounds of 1024 instead of three of 768.
	const int side = 2 * R + 1;
	const int per = (side + parts - 1) / parts;
	const int cy0 = part * per, cnt = min(per, side - cy0);              // this workgroup's candidate rows cy0 .. cy0 + cnt - 1 (the launcher keeps cnt >= 1)
	uint8_t *s_win = lds;                                                  // (H + cnt - 1) rows of win_stride bytes: window rows cy0 ..
	if (W64) W = 64;
	const int wd = W64 ? 16 : W / 4;
	const int win_w = W + 2 * R, win_h = H + cnt - 1;
	uint32_t *s_src = reinterpret_cast<uint32_t *>(lds + (size_t)win_stride * win_h);         // H rows of W / 4 dwords
	unsigned long long *best_of_wave = reinterpret_cast<unsigned long long *>(lds);          // reused after the search

	for (int i = threadIdx.x; i < wd * H; i += kThreads)
	{
		const int y = i / wd, x = i - y * wd;
		s_src[i] = reinterpret_cast<const U1 *>(sp + (ptrdiff_t)y * stride_src + 4 * x)->v;
	}
	// window rows, zero-padded to win_stride: whole 16-byte pieces first (one load and one ds_write_b128 per piece, the (row, piece) pair
	// stepped without a division), then what is left of each row dword by dword
	const uint8_t *rp = rp0 + (ptrdiff_t)(cy0 - R) * stride_ref - R;
	{
		const int cpr = win_w >> 4;                                          // whole pieces per row
		if (cpr > 0)
		{
			int y = (int)threadIdx.x / cpr, x = (int)threadIdx.x - y * cpr;
			const int dy = kThreads / cpr, dx = kThreads - dy * cpr;
			while (y < win_h)
			{
				*reinterpret_cast<u32x4 *>(s_win + y * win_stride + 16 * x) = reinterpret_cast<const U4 *>(rp + (ptrdiff_t)y * stride_ref + 16 * x)->v;
				y += dy; x += dx;
				if (x >= cpr) { x -= cpr; y += 1; }
			}
		}
		const int first = cpr * 4, row_dwords = win_stride / 4, rest = row_dwords - first;       // dwords [first, row_dwords) of every row
		for (int i = threadIdx.x; i < rest * win_h; i += kThreads)
		{
			const int y = i / rest, x = first + i - y * rest;
			const uint8_t *p = rp + (ptrdiff_t)y * stride_ref + 4 * x;
			uint32_t v = 0;
			if (4 * x + 4 <= win_w) v = reinterpret_cast<const U1 *>(p)->v;
			else
				for (int b = 0; 4 * x + b < win_w; ++b) v |= (uint32_t)p[b] << (8 * b);   // never read past the window the caller guarantees
			*reinterpret_cast<uint32_t *>(s_win + y * win_stride + 4 * x) = v;
		}
	}
	__syncthreads();

	unsigned long long best = ~0ull;
	// Work items.  A FULL item = (dy, 16 consecutive dx): 16 v_qsad per 16 source bytes, the window bytes shared by the item's
	// four quads of candidates.  A QUARTER item = (dy, 4 consecutive dx): one v_qsad per source dword.  side = 2R + 1 is odd,
	// so a row of candidates is nfg full groups and a remainder (+-64: 8 groups and ONE candidate, dx = +64); dealing
	// ceil(side / 16) full items per row -- as this kernel did until round 3 -- spends a ninth of the work on candidates
	// outside the window, and side * 9 = 1161 items over 256 threads is 5 rounds of which the last is 13 % full.  Instead:
	// the side * nfg full items are dealt in whole rounds of 256; what is left over (1032 - 4 * 256 = 8 items) is cut into
	// quarters and joins the remainder columns (129 quarter items) in ONE short round: 4.25 rounds instead of 5, and no lane
	// ever scores more than 3 candidates outside the window.
	const int nfg = side >> 4;                                 // full groups per candidate row
	const int rem_q = ((side & 15) + 3) >> 2;                  // quarter groups that hold a row's remaining candidates
	const int full_items = cnt * nfg;
	const int whole = full_items & ~(kThreads - 1);             // dealt as full items
	const int quarters = 4 * (full_items - whole) + cnt * rem_q;
	// (Round 4: dealing the lanes so that every 16-lane group of a ds_read_b128 covers all 64 LDS banks -- two candidate rows 384 bytes apart per group
	// instead of rows 192 bytes apart, which collide 2-way -- removes the bank conflicts of profiles/r03_me_counters.txt and changes nothing:
	// +0.9 % on the 4K frame, profiles/r04_me_lane_deal_ab.txt.  The LDS is 41 % busy; the kernel waits on the v_qsad issue rate.  Natural deal kept.)
	for (int item = threadIdx.x; item < whole; item += kThreads)
	{
		const int cy = item / nfg, g = item - cy * nfg;   // candidates (cy, 16 g .. 16 g + 15)
		uint32_t acc[16];
#pragma unroll
		for (int k = 0; k < 16; ++k) acc[k] = 0;
		for (int y0 = 0; y0 < H; y0 += 4)
		{
			unsigned long long pk[4] = { 0, 0, 0, 0 };   // 4 x (4 packed u16 sums): candidates 4q .. 4q+3
			for (int y = y0; y < y0 + 4; ++y)
			{
				const uint8_t *wrow = s_win + (y + cy) * win_stride + 16 * g;
				const uint32_t *srow = s_src + y * wd;
				int j = 0;
				for (; j + 4 <= wd; j += 4)
				{
					// 16 source bytes (same address in every lane: LDS broadcast) and the 36 window bytes
					// starting above them, as two aligned 16-byte reads and one dword
					const u32x4 s4 = *reinterpret_cast<const u32x4 *>(srow + j);
					const u32x4 wa = *reinterpret_cast<const u32x4 *>(wrow + 4 * j);
					const u32x4 wb = *reinterpret_cast<const u32x4 *>(wrow + 4 * j + 16);
					const uint32_t wc = *reinterpret_cast<const uint32_t *>(wrow + 4 * j + 32);
					const uint32_t w[9] = { wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w, wc };
					const uint32_t sv[4] = { s4.x, s4.y, s4.z, s4.w };
#pragma unroll
					for (int jj = 0; jj < 4; ++jj)
#pragma unroll
						for (int q = 0; q < 4; ++q)
						{
							// candidates 4q .. 4q+3 of this group see source dword j + jj against window dwords (j + jj + q, j + jj + q + 1)
							const unsigned long long window = ((unsigned long long)w[jj + q + 1] << 32) | w[jj + q];
							pk[q] = __builtin_amdgcn_qsad_pk_u16_u8(window, sv[jj], pk[q]);
						}
				}
				for (; j < wd; ++j)   // widths that are not a multiple of 16: one source dword at a time
				{
					const uint32_t sv = srow[j];
					uint32_t w[5];
#pragma unroll
					for (int q = 0; q < 5; ++q) w[q] = *reinterpret_cast<const uint32_t *>(wrow + 4 * (j + q));
#pragma unroll
					for (int q = 0; q < 4; ++q)
						pk[q] = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)w[q + 1] << 32) | w[q], sv, pk[q]);
				}
			}
#pragma unroll
			for (int q = 0; q < 4; ++q)
			{
				acc[4 * q + 0] += (uint32_t)(pk[q] & 0xffff);
				acc[4 * q + 1] += (uint32_t)((pk[q] >> 16) & 0xffff);
				acc[4 * q + 2] += (uint32_t)((pk[q] >> 32) & 0xffff);
				acc[4 * q + 3] += (uint32_t)(pk[q] >> 48);
			}
		}
#pragma unroll
		for (int k = 0; k < 16; ++k)
		{
			const unsigned long long key = ((unsigned long long)acc[k] << 16) | (unsigned)((cy0 + cy) * side + 16 * g + k);      // every candidate of a full group is inside the window
			best = key < best ? key : best;
		}
	}
	for (int qi = threadIdx.x; qi < quarters; qi += kThreads)
	{
		int cy, cx0;       // candidates (cy, cx0 .. cx0 + 3)
		if (qi < 4 * (full_items - whole))
		{
			const int item = whole + (qi >> 2);
			cy = item / nfg;
			cx0 = 16 * (item - cy * nfg) + 4 * (qi & 3);
		}
		else
		{
			const int k = qi - 4 * (full_items - whole);
			cy = k / rem_q;
			cx0 = 16 * nfg + 4 * (k - cy * rem_q);
		}
		uint32_t acc[4] = { 0, 0, 0, 0 };
		for (int y0 = 0; y0 < H; y0 += 4)
		{
			unsigned long long pk = 0;
			for (int y = y0; y < y0 + 4; ++y)
			{
				const uint32_t *wrow = reinterpret_cast<const uint32_t *>(s_win + (y + cy) * win_stride + cx0);
				const uint32_t *srow = s_src + y * wd;
				uint32_t lo = wrow[0];
				int j = 0;
				// four source dwords a trip: one 16-byte read of the source row and the next four window dwords (dword-aligned: two ds_read2_b32) for four v_qsad --
				// a dword at a time it was two LDS reads per v_qsad, and searches of a few hundred candidates (a partition's +-8, the frame chain's +-16) run
				// entirely on this path (one or two 16-candidate groups per row: no whole round of full items)
				for (; j + 4 <= wd; j += 4)
				{
					typedef u32x4 u32x4_dword_aligned __attribute__((aligned(4)));
					const u32x4 s4 = *reinterpret_cast<const u32x4 *>(srow + j);
					const u32x4 w4 = *reinterpret_cast<const u32x4_dword_aligned *>(wrow + j + 1);
					pk = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)w4.x << 32) | lo, s4.x, pk);
					pk = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)w4.y << 32) | w4.x, s4.y, pk);
					pk = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)w4.z << 32) | w4.y, s4.z, pk);
					pk = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)w4.w << 32) | w4.z, s4.w, pk);
					lo = w4.w;
				}
				for (; j < wd; ++j)
				{
					const uint32_t hi = wrow[j + 1];
					pk = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)hi << 32) | lo, srow[j], pk);
					lo = hi;
				}
			}
			acc[0] += (uint32_t)(pk & 0xffff);
			acc[1] += (uint32_t)((pk >> 16) & 0xffff);
			acc[2] += (uint32_t)((pk >> 32) & 0xffff);
			acc[3] += (uint32_t)(pk >> 48);
		}
#pragma unroll
		for (int k = 0; k < 4; ++k)
		{
			if (cx0 + k < side)
			{
				const unsigned long long key = ((unsigned long long)acc[k] << 16) | (unsigned)((cy0 + cy) * side + cx0 + k);
				best = key < best ? key : best;
			}
		}
	}

#pragma unroll
	for (int d = 32; d >= 1; d >>= 1)
	{
		const unsigned long long o = __shfl_xor(best, d, kWave);
		best = o < best ? o : best;
	}
	__syncthreads();   // every wave is done reading the window: its first bytes become the per-wave results
	if ((threadIdx.x & 63) == 0) best_of_wave[threadIdx.x >> 6] = best;
	__syncthreads();
	if (threadIdx.x == 0)
	{
		for (int w = 1; w < kThreads / kWave; ++w) best = best_of_wave[w] < best ? best_of_wave[w] : best;
		if (parts == 1)
		{
			const int c = (int)(best & 0xffff);
			result[3 * blk + 0] = c % side - R;
			result[3 * blk + 1] = c / side - R;
			result[3 * blk + 2] = (int)(best >> 16);
		}
		else
			atomicMin(me_key_slot(result, blk), best);
	}
}

template <bool FRAME, bool W64>
__global__ __launch_bounds__(kThreads) void me_full_search_kernel(
	const uint8_t *__restrict__ src, ptrdiff_t stride_src, const uint8_t *__restrict__ ref, ptrdiff_t stride_ref,
	const uint32_t *__restrict__ src_off, const uint32_t *__restrict__ ref_off, int *__restrict__ result,
	int W, int H, int R, int win_stride /* bytes, multiple of 16 */, MeFrame fr, int parts)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	me_search_workgroup<FRAME, W64>(src, stride_src, ref, stride_ref, src_off, ref_off, result, W, H, R, win_stride, fr, parts, blockIdx.x, lds);
}

// ---- one launch for a frame's partition: runs of equally sized blocks (round 6) -------------------------------------------------------------------
//
// The kernel above takes its geometry at run time, so a frame's PUs of many sizes (pred_inter.c:438-444) need only a table: run i = `count` consecutive
// descriptors of one size; a workgroup finds its run by comparing its index with the runs' first workgroups (one workgroup per block, the table in the
// kernel arguments, as pred_inter.hip's pred_runs_kernel) and searches its block.  The dynamic LDS is sized for the largest run.
constexpr int kMaxMeRuns = 96;
struct MeRunDev { uint32_t first, count; uint16_t W, H; };
struct MeRunsArgs
{
	const uint8_t *src, *ref;
	ptrdiff_t stride_src, stride_ref;
	const uint32_t *src_off, *ref_off;
	int *result;
	int R;
	uint32_t wg_start[kMaxMeRuns];      // first workgroup of run i; unused entries hold 0xffffffff
	MeRunDev runs[kMaxMeRuns];
};

__global__ __launch_bounds__(kThreads) void me_full_search_runs_kernel(const MeRunsArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	uint32_t at_or_before = 0, start = 0;
#pragma unroll
	for (int i = 0; i < kMaxMeRuns; ++i)
	{
		const bool in = a.wg_start[i] <= blockIdx.x;
		at_or_before += in ? 1u : 0u;
		start = in ? a.wg_start[i] : start;      // starts ascend: the last one taken is the run's
	}
	const MeRunDev r = a.runs[at_or_before - 1];
	const int W = r.W, win_stride = (W + 2 * a.R + 15) & ~15;
	me_search_workgroup<false, false>(a.src, a.stride_src, a.ref, a.stride_ref, a.src_off + r.first, a.ref_off + r.first, a.result + 3 * (size_t)r.first,
		W, r.H, a.R, win_stride, MeFrame{}, 1, blockIdx.x - start, lds);
}

// several workgroups per block: the 8-byte-aligned 8 bytes inside block k's 12-byte result slot (me_key_slot) hold the running minimum key,
// preset to all ones; this turns them into the result
__global__ __launch_bounds__(kThreads) void me_decode_kernel(int *result, uint32_t n_blocks, int R)
{
	const uint32_t k = blockIdx.x * kThreads + threadIdx.x;
	if (k >= n_blocks) return;
	const unsigned long long key = *me_key_slot(result, k);
	const int side = 2 * R + 1, c = (int)(key & 0xffff);
	result[3 * k + 0] = c % side - R;
	result[3 * k + 1] = c / side - R;
	result[3 * k + 2] = (int)(key >> 16);
}

}  // namespace hevc

using namespace hevc;

// LDS bytes of one workgroup: `win_rows` window rows of win_stride bytes (H + 2 range of them when one workgroup takes all of a block's
// candidate rows), then the source tile, which also absorbs what the last dx group reads past the last window row (see the kernel's layout note)
static size_t me_lds_bytes(int W, int H, int range, int win_rows, int *win_stride)
{
	const int n_groups = (2 * range + 1 + 15) / 16;                  // dx groups of 16 (the quarter items stay inside the same bound)
	*win_stride = (W + 2 * range + 15) & ~15;
	const int overread = 16 * n_groups + W + 4 - *win_stride;        // bytes the last window row is read past its end
	const int tail = W * H > overread ? W * H : overread;
	return (size_t)*win_stride * win_rows + (size_t)((tail + 15) & ~15);
}

// Workgroups per block.  One when the launch fills the device's workgroup slots by itself; a launch too small for that (one rank's share of a
// frame split over 8 GPUs) cuts every block's candidate ROWS into 2 or 4 parts.  Slots = compute units of the CURRENT device (a CPX / DPX
// partition has 32 / 128) x the workgroups a CU holds at the LDS footprint of the launch that would actually be issued -- the PART's window
// (H + rows-per-part - 1 rows), not the whole one: 160 KiB of LDS, at most 8 workgroups of 256 threads; at 64x64 / +-64 that is 6 per CU in 4 parts
// (23 KiB each), 5 in 2 parts (29 KiB), 3 whole (41 KiB).  Splitting costs a memset, a decode launch and re-staged window rows, so it is only done
// where a block is a large piece of work -- W H (2 range + 1)^2 of at least 2^23 byte differences (32x32 / +-64 upwards; smaller blocks were
// measured in round 5, profiles/r05_me_small_parts_ab.txt) -- and while the part-workgroups fit one round of those slots (4 parts) or 1.25 rounds
// (2 parts): on 256 CUs at 64x64 / +-64 up to 384 blocks in 4 parts, up to 800 in 2 (measured: profiles/r03_me_shard_launches.txt, 255 / 510 / 1020 CTUs).
static int me_parts(size_t n_blocks, int W, int H, int range)
{
	static const int forced = getenv("HEVCASM_ME_PARTS") ? atoi(getenv("HEVCASM_ME_PARTS")) : 0;    // measurement switch, read once per process
	const int side = 2 * range + 1;
	int parts = 1;
	if (forced >= 1 && forced <= 8) parts = forced;
	else if ((size_t)W * H * side * side >= ((size_t)1 << 23))
	{
		// four parts only for the largest blocks (64x64 / +-64 = 2^26 byte differences): at 32x32 / +-64 two parts beat four at every list length
		// (64 blocks 0.048 against 0.061 ms, 256 blocks 0.066 against 0.082: profiles/r05_me_small_parts_ab.txt)
		const int most = (size_t)W * H * side * side >= ((size_t)1 << 26) ? 4 : 2;
		for (int p = most; p >= 2 && parts == 1; p >>= 1)
		{
			int ws;
			const size_t lds = me_lds_bytes(W, H, range, H + (side + p - 1) / p - 1, &ws);
			const size_t per_cu = (160 * 1024) / ((lds + 1023) & ~(size_t)1023);
			const size_t slots = (size_t)device_cus() * (per_cu < 1 ? 1 : per_cu > 8 ? 8 : per_cu);
			if (p == 4 ? n_blocks * 4 <= slots : n_blocks * 2 * 4 <= slots * 5) parts = p;
		}
	}
	while (parts > 1 && side < 8 * parts) --parts;              // every part gets candidate rows
	return parts;
}

// launches the search (and, with several workgroups per block, the preset of the key slots before it and their decoding after it)
template <bool FRAME>
static int me_launch(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref, const uint32_t *src_off, const uint32_t *ref_off,
		int *result, int W, int H, int range, bool w64, size_t n_blocks, const MeFrame &fr, hipStream_t st, size_t n_frames = 1)
{
	int win_stride;
	// the domain does not depend on the batch: a window must fit LDS whole, whether or not this launch cuts it into parts
	const size_t lds_whole = me_lds_bytes(W, H, range, H + 2 * range, &win_stride);
	if (lds_whole > 64 * 1024) return set_error(HEVCASM_HIP_EINVAL, "search window does not fit in 64 KiB of LDS");
	const int parts = me_parts(n_blocks * n_frames, W, H, range);      // what counts is how many workgroups the LAUNCH has
	const int per = (2 * range + 1 + parts - 1) / parts;             // candidate rows per workgroup
	const size_t shmem = me_lds_bytes(W, H, range, H + per - 1, &win_stride);
	if (parts > 1) HEVC_HIP_TRY(hipMemsetAsync(result, 0xff, 12 * n_blocks * n_frames, st));
	const dim3 grid((uint32_t)(n_blocks * parts), (uint32_t)n_frames);
	if (w64) hipLaunchKernelGGL((me_full_search_kernel<FRAME, true>), grid, dim3(kThreads), shmem, st, src, stride_src, ref, stride_ref, src_off, ref_off, result, W, H, range, win_stride, fr, parts);
	else hipLaunchKernelGGL((me_full_search_kernel<FRAME, false>), grid, dim3(kThreads), shmem, st, src, stride_src, ref, stride_ref, src_off, ref_off, result, W, H, range, win_stride, fr, parts);
	if (int e = launch_status("me_full_search_kernel")) return e;
	if (parts > 1)
	{
		hipLaunchKernelGGL(me_decode_kernel, dim3(ceil_div(n_blocks * n_frames, (size_t)kThreads)), dim3(kThreads), 0, st, result, (uint32_t)(n_blocks * n_frames), range);
		return launch_status("me_decode_kernel");
	}
	return 0;
}

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
	if (n_blocks > 0x1fffffffull) return set_error(HEVCASM_HIP_EINVAL, "n_blocks too large");
	if (!src || !ref || !src_off || !ref_off || !result) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	return me_launch<false>(src, stride_src, ref, stride_ref, src_off, ref_off, result, W, H, range, W == 64, n_blocks, MeFrame{}, as_stream(stream));
}

extern "C" int hevcasm_hip_me_full_search_runs(
	const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref,
	const uint32_t *src_off, const uint32_t *ref_off, int *result,
	const hevcasm_hip_cost_run *runs, int n_runs, int range, void *stream)
{
	if (int e = check_device()) return e;
	if (n_runs < 0) return set_error(HEVCASM_HIP_EINVAL, "negative run count");
	if (range < 0 || range > 127) return set_error(HEVCASM_HIP_EINVAL, "range must be 0..127");
	if (n_runs == 0) return 0;
	if (!runs || !src || !ref || !src_off || !ref_off || !result) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	std::vector<MeRunDev> table;
	table.reserve((size_t)n_runs);
	size_t shmem = 0;
	for (int i = 0; i < n_runs; ++i)
	{
		const int W = (int)(runs[i].rect >> 8), H = (int)(runs[i].rect & 0xff);
		if (W < 4 || W > 64 || H < 4 || H > 64 || (W & 3) || (H & 3) || (runs[i].rect >> 16)) return set_error(HEVCASM_HIP_EINVAL, "block size must be a multiple of 4 in 4..64");
		if ((uint64_t)runs[i].first + runs[i].count > 0x1fffffffull) return set_error(HEVCASM_HIP_EINVAL, "n_blocks too large");
		int win_stride;
		const size_t lds = me_lds_bytes(W, H, range, H + 2 * range, &win_stride);
		if (lds > 64 * 1024) return set_error(HEVCASM_HIP_EINVAL, "search window does not fit in 64 KiB of LDS");
		if (runs[i].count == 0) continue;
		shmem = lds > shmem ? lds : shmem;
		table.push_back(MeRunDev{ runs[i].first, runs[i].count, (uint16_t)W, (uint16_t)H });
	}
	// the largest blocks first: the launch ends when its last workgroup does
	std::stable_sort(table.begin(), table.end(), [](const MeRunDev &x, const MeRunDev &y) { return (int)x.W * x.H > (int)y.W * y.H; });
	for (size_t base = 0; base < table.size(); base += kMaxMeRuns)      // more runs than one table holds: the next launch
	{
		MeRunsArgs a = {};
		a.src = src; a.ref = ref; a.stride_src = stride_src; a.stride_ref = stride_ref; a.src_off = src_off; a.ref_off = ref_off; a.result = result; a.R = range;
		uint32_t wgs = 0, used = 0;
		for (int i = 0; i < kMaxMeRuns; ++i) a.wg_start[i] = 0xffffffffu;
		for (size_t i = base; i < table.size() && i < base + kMaxMeRuns; ++i)
		{
			a.runs[used] = table[i];
			a.wg_start[used++] = wgs;
			wgs += table[i].count;
		}
		hipLaunchKernelGGL(me_full_search_runs_kernel, dim3(wgs), dim3(kThreads), shmem, as_stream(stream), a);
		if (int e = launch_status("me_full_search_runs_kernel")) return e;
	}
	return 0;
}

static int me_frames_ctus(const uint8_t *cur, ptrdiff_t stride_cur, ptrdiff_t pitch_cur, const uint8_t *ref, ptrdiff_t stride_ref, ptrdiff_t pitch_ref, size_t n_frames,
	int frame_w, int frame_h, int ctu, int range, int first_ctu, int n_ctus, int *result, void *stream)
{
	if (int e = check_device()) return e;
	if (ctu < 4 || ctu > 64 || (ctu & 3)) return set_error(HEVCASM_HIP_EINVAL, "CTU size must be a multiple of 4 in 4..64");
	if (frame_w < 4 || frame_h < 4 || (frame_w & 3) || (frame_h & 3)) return set_error(HEVCASM_HIP_EINVAL, "frame dimensions must be multiples of 4");
	if (range < 0 || range > 127) return set_error(HEVCASM_HIP_EINVAL, "range must be 0..127");
	const int rows = (frame_h + ctu - 1) / ctu, cols = (frame_w + ctu - 1) / ctu;
	if (first_ctu < 0 || n_ctus < 0 || (long long)first_ctu + n_ctus > (long long)rows * cols) return set_error(HEVCASM_HIP_EINVAL, "CTUs outside the frame");
	if (n_frames > 65535) return set_error(HEVCASM_HIP_EINVAL, "at most 65535 frames per launch");
	if ((unsigned long long)n_frames * (unsigned long long)n_ctus > 0x1fffffffull) return set_error(HEVCASM_HIP_EINVAL, "too many CTUs in one launch");
	if (n_ctus == 0 || n_frames == 0) return 0;
	if (!cur || !ref || !result) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	// the planes point at column 0 of the CTU row that holds first_ctu; the slab ends with the row that holds the last CTU
	const int row0 = first_ctu / cols, row1 = (first_ctu + n_ctus - 1) / cols + 1;
	const int y1 = row1 * ctu < frame_h ? row1 * ctu : frame_h;
	const MeFrame fr = { cols, ctu, frame_w, y1 - row0 * ctu, first_ctu - row0 * cols, pitch_cur, pitch_ref, (uint32_t)n_ctus };
	return me_launch<true>(cur, stride_cur, ref, stride_ref, nullptr, nullptr, result, ctu, ctu, range, ctu == 64 && frame_w % 64 == 0,
			(size_t)n_ctus, fr, as_stream(stream), n_frames);
}

extern "C" int hevcasm_hip_me_full_search_frame_ctus(
	const uint8_t *cur, ptrdiff_t stride_cur, const uint8_t *ref, ptrdiff_t stride_ref,
	int frame_w, int frame_h, int ctu, int range, int first_ctu, int n_ctus, int *result, void *stream)
{
	return me_frames_ctus(cur, stride_cur, 0, ref, stride_ref, 0, 1, frame_w, frame_h, ctu, range, first_ctu, n_ctus, result, stream);
}

extern "C" int hevcasm_hip_me_full_search_frames_ctus(
	const uint8_t *cur, ptrdiff_t stride_cur, ptrdiff_t frame_pitch_cur, const uint8_t *ref, ptrdiff_t stride_ref, ptrdiff_t frame_pitch_ref, size_t n_frames,
	int frame_w, int frame_h, int ctu, int range, int first_ctu, int n_ctus, int *result, void *stream)
{
	return me_frames_ctus(cur, stride_cur, frame_pitch_cur, ref, stride_ref, frame_pitch_ref, n_frames, frame_w, frame_h, ctu, range, first_ctu, n_ctus, result, stream);
}

extern "C" int hevcasm_hip_me_full_search_frame_rows(
	const uint8_t *cur, ptrdiff_t stride_cur, const uint8_t *ref, ptrdiff_t stride_ref,
	int frame_w, int frame_h, int ctu, int range, int row0, int row1, int *result, void *stream)
{
	if (int e = check_device()) return e;
	if (ctu < 4 || ctu > 64 || (ctu & 3)) return set_error(HEVCASM_HIP_EINVAL, "CTU size must be a multiple of 4 in 4..64");
	if (frame_w < 4 || frame_h < 4 || (frame_w & 3) || (frame_h & 3)) return set_error(HEVCASM_HIP_EINVAL, "frame dimensions must be multiples of 4");
	const int rows = (frame_h + ctu - 1) / ctu, cols = (frame_w + ctu - 1) / ctu;
	if (row0 < 0 || row1 > rows || row0 > row1) return set_error(HEVCASM_HIP_EINVAL, "CTU rows outside the frame");
	return hevcasm_hip_me_full_search_frame_ctus(cur, stride_cur, ref, stride_ref, frame_w, frame_h, ctu, range, row0 * cols, (row1 - row0) * cols, result, stream);
}
