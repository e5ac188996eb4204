// sad.hip -- batched SAD, 4-way multi-reference SAD and SSD for gfx950.
//
// Replaces, for a whole batch per launch: hevcasm_sad (reference sad.h:50, C body sad.c:47-60),
// hevcasm_sad_multiref 4-way (sad.h:95, sad.c:101-121) and hevcasm_ssd (ssd.h:53, ssd.c:43-55).
//
// Layout: frames stay as the codec has them (row-strided u8 planes in HBM); a block is a base
// pointer + u32 byte offset.  Mapping: each block row of W bytes is cut into chunks of C = 16 / 12 / 8 /
// 4 bytes, one lane per chunk, and R rows of the block are in flight per instruction, so
// (W / C) * R lanes own one block (see Tiling, Plan and launch_w below for the measurements behind the
// choices) and a wave covers 64 / LPB blocks.  Consecutive blocks are normally horizontal
// neighbours in the frame, so the lanes of a wave read a few long runs of a few frame rows: no
// LDS staging is needed (the four-reference kernel of 64-/48-wide blocks further down is the exception).
// Reference blocks sit at arbitrary byte alignment (motion vectors).  What round 2 measured about the vector L1 decides the
// rest: it works per (quad of lanes, row) access and per dword when a load is not dword-aligned, so narrow blocks get FEW lanes to a
// row, MANY rows to an instruction, loads of the aligned words that contain the lane's bytes (one dword wider than the chunk,
// realigned with v_alignbyte_b32) and ONE trip; and the lanes of the narrowest are dealt row-major over the wave's blocks (Deal).
// Arithmetic: v_sad_u8 accumulates 4 byte differences per instruction; SSD uses
// sum(a-b)^2 = sum a^2 + sum b^2 - 2 sum ab with three v_dot4_u32_u8 (exact: every partial sum
// is below 2^30 for blocks up to 64x64).  The LPB partial sums are combined with wave shuffles.
// HBM-bound for one reference; with four overlapping candidates the vector L1 / L2 carry the re-reads
// and HBM sees each plane byte about once (DESIGN.md section 4 has the byte accounting).

#include "common.hpp"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace hevc {

// Lane mapping of one W x H block: LPR = W / C lanes side by side cover a row, R such rows are
// in flight per instruction, so LPB = LPR * R lanes own a block and lane (r, c) walks rows
// r, r + R, r + 2R, ...  A wave holds 64 / LPB blocks.
//
// Why not simply one lane per block (R = 1)?  With per-block motion vectors the lanes of a wave
// then read 64 different frame rows per instruction and the vector L1 (TCP) becomes the limit:
// measured on MI355X, 1 Mi 16x16 blocks, displacements uniform in +-7: R = 1 -> 3.0 TB/s, 6.4x
// the TCP accesses of the aligned case (profiles/r01_sad16_counters_R1_mapping.txt).  Keeping a block's
// rows together (8-32 lanes per block) lets neighbouring lanes hit the same lines:
// C = 8, R = 4 -> 4.7 TB/s at 16x16; C = 16, R = 8 -> 5.7 TB/s at 32x32 and 6.2 TB/s at 64x64
// (on-GPU sweeps of an exploration build of this kernel: profiles/r01_sad_tiling_sweep_ways{1,4}.txt).
// Coherent motion is within 10 % of its best mapping.
template <int W, int C, int R> struct Tiling
{
	static constexpr int LPR = W / C;        // lanes per row
	static constexpr int LPB = LPR * R;      // lanes per block
	static constexpr int BPW = kWave / LPB;  // blocks per wave
	static_assert(LPB <= kWave, "a block must fit in one wave");
};

// Lane deal of a wave's BPW blocks.  Block-major: lane = (block, row, chunk) -- a block's lanes are neighbours.  ROW-major: lane = (row,
// block, chunk) -- the same row of the wave's blocks lies in neighbouring lanes, so that a quad of lanes (what the address unit takes per
// cycle) reads ONE frame row of two or four raster-neighbour blocks instead of two or four rows of one block: fewer vector-L1 accesses
// and address translations for the source rows (the reference rows are displaced block by block either way).  Same-box A/B, row-major
// against block-major (profiles/r02_sad_rowmajor_ab.txt): one reference 8x8 -13 %, 8x16 -6 %, 8x32 -8 %, 16x8 -3 %, 16x16 -1 %, 16x32 / 16x64
// -1.5 %; four references 32x8 ... 32x64 -2 ... -4 %; but one reference 32-wide +1 ... +2 % and four references 16-wide +-0: those stay block-major.
template <int W, int C, int R, int WAYS> struct Deal
{
	typedef Tiling<W, C, R> T;
	static constexpr int LPR = T::LPR, LPB = T::LPB, BPW = T::BPW;
	static constexpr bool ROWMAJOR = BPW > 1 && LPB * BPW == kWave && (R & (R - 1)) == 0 && (LPR & (LPR - 1)) == 0 &&
		((WAYS == 1 && (W == 8 || W == 12 || W == 16 || W == 24)) || (WAYS == 4 && W == 32));
	static __device__ __forceinline__ uint32_t sub_block(uint32_t lane) { return ROWMAJOR ? (lane / LPR) % BPW : lane / LPB; }
	static __device__ __forceinline__ uint32_t row(uint32_t lane) { return ROWMAJOR ? lane / (LPR * BPW) : (lane % LPB) / LPR; }
	static __device__ __forceinline__ uint32_t chunk(uint32_t lane) { return lane % LPR; }   // LPR divides LPB
	static __device__ __forceinline__ bool leader(uint32_t lane) { return ROWMAJOR ? lane < (uint32_t)(LPR * BPW) && lane % LPR == 0 : lane % LPB == 0; }
};

// Chunk bytes and rows in flight for a block width.  Among R = 8, 4, 2, 1 that fit a wave, take the
// one with the best (lane utilisation x row-locality weight); narrow blocks stop at R = 4.
constexpr int tiling_score(int lpr, int r)
{
	const int lpb = lpr * r;
	if (lpb > kWave) return 0;
	const int used = (kWave / lpb) * lpb;                       // busy lanes out of 64
	const int weight = r >= 4 ? 100 : (r == 2 ? 85 : 70);      // measured: R = 1 loses 30-40 % on divergent motion
	return used * weight;
}
constexpr int best_rows(int lpr, int max_r)
{
	int best = 1, best_score = 0;
	for (int r = max_r; r >= 1; r /= 2)
		if (tiling_score(lpr, r) > best_score) { best = r; best_score = tiling_score(lpr, r); }
	return best;
}
template <int W> struct Plan
{
	// 48 = 4 lanes x 12 bytes (global_load_dwordx3): a row is exactly one quad of lanes, as at W = 64.
	// With 3 x 16 bytes the quads of the address unit straddle rows: 1.64x the vector-L1 accesses per
	// pixel of the 64-wide mapping, and 4-way 48x64 was the slowest kernel of the whole mix.
	// 12 = ONE lane of 12 bytes per row, as 8-wide blocks are one lane of 8 (round 2: three lanes of 4 bytes cost the vector L1 three times the
	// accesses per row -- 12x16 one reference -26 %, four references -38 %, 12x8 -26 / -37 %, 12x32 -26 %: profiles/r02_sad_rowmajor_ab.txt)
	// 24 = two lanes of 12 bytes (aligned 16-byte loads that overlap by a dword, no tail load) instead of three of 8: one reference -5 ... -16 %,
	// four references -2 ... -15 % (round 2; in round 1, before the other changes to these kernels, the two were equal)
	static constexpr int C = (W == 48 || W == 12 || W == 24) ? 12 : (W >= 32 && W % 16 == 0) ? 16 : (W % 8 == 0 ? 8 : 4);
	static constexpr int LPR = W / C;
	static constexpr int R_PREF = best_rows(LPR, W >= 32 ? 8 : 4);   // used when it divides H
	static constexpr int R_SAFE = best_rows(LPR, 4);                  // H is always a multiple of 4
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

__device__ __forceinline__ int med3_i32(int x, int lo, int hi)
{
	int r;
	asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi));
	return r;
}

// Sum over the 64 lanes of a wave with DPP adds (no LDS crossbar): quad, row of 16, then the row broadcasts; the total
// is in lane 63.
__device__ __forceinline__ uint32_t wave_sum_to_lane63(uint32_t v)
{
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, false);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e /* quad_perm [2,3,0,1] */, 0xf, 0xf, false);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114 /* row_shr:4 */, 0xf, 0xe, false);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118 /* row_shr:8 */, 0xf, 0xc, false);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
	return v;
}

// Sum `v` over the LPR x R lanes of a block (lane j = r * LPR + c); valid in the block's lane 0.
template <int LPR, int R> __device__ __forceinline__ uint32_t reduce_block(uint32_t v)
{
	constexpr int LPB = LPR * R;
	if ((LPB & (LPB - 1)) == 0)
	{
#pragma unroll
		for (int d = LPB / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, kWave);
		return v;
	}
	// rows first (R is a power of two), then the LPR chunks of row 0
#pragma unroll
	for (int d = R / 2; d >= 1; d >>= 1) v += __shfl_down(v, d * LPR, kWave);
	uint32_t t = v;
#pragma unroll
	for (int c = 1; c < LPR; ++c) t += __shfl_down(v, c, kWave);
	return t;
}

// Reference rows of the 4-way kernel, loaded dword-ALIGNED.  A byte-misaligned 16-byte lane load
// costs the vector L1 about a quarter more than a dword-aligned one (measured: 4-way 64x64,
// 1.41 ms with arbitrary dx, 1.11 ms with dx a multiple of 4; tools/explore_sad4_align.py), so the
// lane loads the aligned dwords that contain its chunk and realigns in registers with
// v_alignbyte_b32.  The one extra dword comes from the next lane of the row (DPP) or, for the
// row's last lane, from a 4-byte load that is issued only when the row really is misaligned.
// Every address touched lies in an aligned dword that holds at least one byte of the block row,
// so no access can fault even at the very edges of a plane.
template <int C, int LPR> struct AlignedRefChunk
{
	uint32_t d[C / 4];
	uint32_t tail;
	// `base` is dword-aligned, so (off & 3) is the chunk's misalignment and (off & ~3) never underflows.
	// The C aligned bytes that contain the chunk's first byte:
	__device__ __forceinline__ void load_main(const uint8_t *base, uint32_t off)
	{
		Chunk<C> c;
		c.load(base + (size_t)(off & ~3u));
#pragma unroll
		for (int i = 0; i < C / 4; ++i) d[i] = c.d[i];
	}
	// the row's last lane only: the dword after them (when the row is aligned it is not needed and
	// must not be touched -- it may lie outside the plane -- so an in-row dword is loaded instead)
	__device__ __forceinline__ void load_tail(const uint8_t *base, uint32_t off)
	{
		tail = *reinterpret_cast<const uint32_t *>(base + (size_t)((off & ~3u) + ((off & 3u) ? C : 0)));
	}
	__device__ __forceinline__ void finish(uint32_t sh, bool last_lane_of_row, Chunk<C> &out) const
	{
		uint32_t next = tail;
		if (LPR > 1)
		{
			const uint32_t from_neighbour = (uint32_t)__builtin_amdgcn_mov_dpp((int)d[0], 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
			next = last_lane_of_row ? tail : from_neighbour;
		}
#pragma unroll
		for (int i = 0; i < C / 4; ++i)
			out.d[i] = __builtin_amdgcn_alignbyte(i + 1 < C / 4 ? d[i + 1] : next, d[i], sh);
	}
};

// Reference rows of 8- or 12-byte chunks, two or more lanes to a row, loaded dword-aligned with NO tail load: each
// lane loads the C + 4 aligned bytes that contain its C (global_load_dwordx3 / x4), so the loads of neighbouring lanes
// overlap by a dword.  The vector L1 counts one access per quad of lanes and row for dword-aligned loads whether or
// not they overlap (profiles/r01_tcp_access_model.txt: 12 bytes at 8-byte pitch, 2 lanes per row -> 32 accesses for
// 32 rows), where the same row costs two accesses when it is loaded at its byte address and is misaligned.
// A row's last lane whose chunk is aligned would read the dword past the row, which may lie outside the plane: it
// loads the dword BEFORE its chunk instead (inside the row, LPR >= 2) and uses the upper part.
// One lane to a row (LPR == 1, 8-wide blocks): there is no chunk before the lane's own, so the flag the caller passes is
// not "last lane of the row" but "this row has the highest address of the block": every other row may read the 4 bytes
// past its end (the next row up lies beyond them), and the highest one reads the 4 bytes before its start (the row below
// lies before them) -- both inside the plane as long as rows are at least 8 bytes apart, which the launcher checks
// (closer rows get the TIGHT kernels, which load at the byte address).
template <int C, int LPR> struct OverlapRefChunk
{
	static_assert(C == 4 || C == 8 || C == 12, "chunk + one dword must be a load width");
	uint32_t d[C / 4 + 1];
	__device__ __forceinline__ void load(const uint8_t *base, uint32_t off, bool last_lane_of_row)
	{
		const uint32_t back = (last_lane_of_row && !(off & 3u)) ? 4u : 0u;
		Chunk<C + 4> c;
		c.load(base + (size_t)((off & ~3u) - back));
#pragma unroll
		for (int i = 0; i <= C / 4; ++i) d[i] = c.d[i];
	}
	__device__ __forceinline__ void finish(uint32_t sh, bool last_lane_of_row, Chunk<C> &out) const
	{
		const bool back = last_lane_of_row && !sh;
		uint32_t w[C / 4 + 1];
#pragma unroll
		for (int i = 0; i < C / 4; ++i) w[i] = back ? d[i + 1] : d[i];
		w[C / 4] = d[C / 4];
#pragma unroll
		for (int i = 0; i < C / 4; ++i) out.d[i] = __builtin_amdgcn_alignbyte(w[i + 1], w[i], sh);
	}
};

// the same for the row-major lane deal: the block's lanes are LPR neighbours in each of R groups LPR * BPW lanes apart; valid in every lane
template <int LPR, int R, int BPW> __device__ __forceinline__ uint32_t reduce_block_rowmajor(uint32_t v)
{
#pragma unroll
	for (int d = LPR * BPW * R / 2; d >= LPR * BPW; d >>= 1) v += __shfl_xor(v, d, kWave);
#pragma unroll
	for (int d = LPR / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, kWave);
	return v;
}

// U = rows per lane per trip (independent loads in flight: U * (1 + WAYS) per lane)
// LOOP = false: the block is exactly one trip (H == R * U): no reload loop, and only one register set.
// TIGHT: the reference rows of an 8-wide block are less than 8 bytes apart (see OverlapRefChunk, one lane to a row)
// One wave's sums: lane `lane` belongs to sub-block Deal::sub_block(lane) of the wave and works on block `blk` (every lane of a
// sub-block is given the same, valid, block index); rows [row_first, row_first + H) of it; the sums arrive in the sub-block's lane 0.
template <int W, int C, int R, int U, int WAYS, int OP, bool LOOP, bool TIGHT>
__device__ __forceinline__ void block_cost_sums(
	const uint8_t *__restrict__ src, ptrdiff_t stride_src,
	const uint8_t *__restrict__ ref, ptrdiff_t stride_ref,
	const uint32_t *__restrict__ src_off, const uint32_t *__restrict__ ref_off,
	int H /* rows of this part */, uint32_t lane, uint32_t blk, uint32_t row_first /* the part starts at this row of the block */, uint32_t (&total)[WAYS],
	int scalar_first = -1 /* >= 0: wave-uniform index of the wave's first block, all BPW of which exist: descriptors by scalar loads */)
{
	typedef Tiling<W, C, R> T;
	constexpr int LPR = T::LPR, BPW = T::BPW;
	// scalar descriptors where a wave holds 2 .. 16 blocks (one block: the caller's index is uniform already; 32 blocks: the picking costs more than it saves)
#ifndef HEVCASM_SAD_SCALAR_DESC_MAX
#define HEVCASM_SAD_SCALAR_DESC_MAX 16
#endif
	constexpr bool SCALAR_DESC = BPW >= 2 && BPW <= HEVCASM_SAD_SCALAR_DESC_MAX && BPW * T::LPB == kWave;

	typedef Deal<W, C, R, WAYS> D;
	const uint32_t row0 = D::row(lane), chunk = D::chunk(lane);

	// Descriptors.  Left as per-lane vector loads they are a wave's first trip through the CU's vector-memory pipeline, which returns in order: a new
	// wave's offsets queue behind the row loads of the waves already resident -- 5 k of a 16x16 wave's 11 k cycles pass before they arrive
	// (profiles/r01_wave_timestamps.txt).  A wave's BPW blocks are consecutive in the arrays, so (round 4) a FULL wave fetches them with SCALAR loads
	// (s_load_dwordx8 / x16 through the scalar cache: a separate, unordered path) and each lane then picks its block's words with v_cndmask on the
	// bits of its block index.  scalar_first = the wave's first block (wave-uniform), or -1: the last, partial wave and the one-block-per-wave callers.
	uint32_t so;
	uint32_t ro[WAYS];
	if (SCALAR_DESC && scalar_first >= 0)
	{
		const uint32_t sub = D::sub_block(lane);
		uint32_t sv[BPW];
#pragma unroll
		for (int j = 0; j < BPW; ++j) sv[j] = src_off[(uint32_t)scalar_first + (uint32_t)j];
		so = pick_by_index<BPW>(sv, sub);
#pragma unroll
		for (int k = 0; k < WAYS; ++k)
		{
			uint32_t rv[BPW];
#pragma unroll
			for (int j = 0; j < BPW; ++j) rv[j] = ref_off[(size_t)((uint32_t)scalar_first + (uint32_t)j) * WAYS + k];
			ro[k] = pick_by_index<BPW>(rv, sub);
		}
	}
	else
	{
		so = src_off[blk];
#pragma unroll
		for (int k = 0; k < WAYS; ++k) ro[k] = ref_off[(size_t)blk * WAYS + k];
	}

	// Addresses are a uniform base (SGPR pair) + a 32-bit per-lane offset (one VGPR per plane and
	// candidate, one integer add per row step): every byte of every block lies within 4 GiB of its
	// plane's base pointer (hevcasm_hip.h), so the offsets cannot wrap.  The reference base is
	// rounded down to a dword so that (offset & 3) is the byte misalignment of the address itself.
	const uint32_t ref_mis = (uint32_t)(reinterpret_cast<uintptr_t>(ref) & 3u);
	const uint8_t *const ref_al = ref - ref_mis;
	const uint32_t ss32 = (uint32_t)stride_src, sr32 = (uint32_t)stride_ref;
	const uint32_t step_src = R * ss32, step_ref = R * sr32;
	uint32_t os = so + (row_first + row0) * ss32 + chunk * C;
	uint32_t orf[WAYS];
#pragma unroll
	for (int k = 0; k < WAYS; ++k) orf[k] = ro[k] + ref_mis + (row_first + row0) * sr32 + chunk * C;

	Acc<C, OP> acc[WAYS];

	// Row loop.  A trip is U row slots per lane, U * (1 + WAYS) loads.  Written slot by slot -- consume
	// slot u, reload slot u for the next trip -- with only a scheduling barrier in between, which lets
	// the compiler hoist every reload above the consumption of the slot it replaces (into fresh
	// registers, copied back once per trip): the next trip's loads go out interleaved with this trip's
	// arithmetic and two trips are in flight per wave.  The alternatives were built and timed on one
	// box (profiles/r01_sad_pipeline_ab.txt; 4-way 64x64, ms per 1 Mi blocks): consume a trip then
	// load the next 2.09; reload each slot strictly in place (half the registers) 2.17; two explicit
	// buffers trip by trip 1.94, slot by slot 2.11; this form 1.81.  Memory-level parallelism per
	// wave outweighs the occupancy lost to the second register set.
	const int trips = LOOP ? H / (R * U) : 1;
	// (the vector L1 splits a load that is not dword-aligned into dwords -- profiles/r01_tcp_access_model.txt -- so the
	// aligned form quarters the accesses of a 64-byte row; it pays for the 4-way kernel, which is bound by that rate.
	// For one reference it was measured slower: 64x64 1.48 against 1.38 ms, 32x32 0.42 against 0.37 -- HBM-bound, and the
	// extra tail loads and funnel shifts are pure cost.  8-byte chunks: slower too, half the lanes issue the tail load.)
	// 8-byte chunks (16-, 24-, 40-, 56-wide): overlapping aligned loads, see OverlapRefChunk.  Same-box A/B, ms per 1 Mi
	// blocks (profiles/r01_sad_overlap_ab.txt): 4-way 16x64 0.75 -> 0.59, 16x32 0.38 -> 0.29, 16x16 0.192 -> 0.170,
	// 24x32 0.59 -> 0.46; one reference 16x64 0.43 -> 0.41, shorter blocks unchanged.
	// Round 1 measured three more re-cuts as no better on the kernels of the time (48-wide as 4 x 16 bytes at 12-byte pitch, 24-wide as
	// 2 x 12-byte chunks with 16-byte loads, 4-way 32-wide as 4 x 8-byte chunks); round 2 re-measured them on the kernels as they then
	// stood and all three pay -- they are what Plan<24> / Plan<32> / Plan<48> below use (profiles/r02_sad_rowmajor_ab.txt: 24-wide as
	// 2 x 12 bytes -6 ... -16 %, 4-way 32-wide as 4 x 8 bytes with 4 rows in flight -5 ... -9 %, one-reference 48-wide with aligned 16-byte loads -4 %).
	// 4-byte chunks (12-, 20-, 28-wide ...): the same with 8-byte aligned loads -- a misaligned dword load costs the L1 four
	// times an aligned one: 12x16 four references 0.27 -> 0.20 ms, one reference 0.125 -> 0.116
	// (With the loads still alternating between source and reference rows, the short single-trip blocks lost 4-10 % to the
	// realignment; since they issue source rows first they gain too: 4-way 16x12 0.140 -> 0.105, 16x8 0.100 -> 0.079.)
	// (12-byte chunks, 24-wide, and 48-wide with one reference -- 48x64 -3.9 %, 48x32 -4.2 %, round 2 -- load 16 aligned bytes the same way)
	constexpr bool OVERLAP = ((C == 8 || C == 4) && LPR >= 2) || (W == 24 && C == 12) || (W == 48 && WAYS == 1);
	// 8-wide blocks (one lane to a row), single trip: the same aligned 12-byte loads, see OverlapRefChunk -- 4-way 8x8
	// 0.078 -> 0.059 ms, 8x16 0.145 -> 0.101, 8x32 0.279 -> 0.193, one reference -2 ... -5 %.  (Deciding about close rows with a
	// branch inside the kernel instead of a template flag made the 4-way kernels three times slower.)
	constexpr bool OVERLAP1 = (C == 8 || C == 12) && LPR == 1 && !LOOP && !TIGHT;
	constexpr bool ALIGNED = WAYS == 4 && C >= 12 && LPR >= 2 && W != 24;      // (one reference, 48-wide, through this path: +1 % -- round 2, profiles/r02_sad_rowmajor_ab.txt)
	const bool last = chunk == LPR - 1;

	// Source rows are read exactly once.  Where the blocks of a wave cover whole 128-byte lines of a frame row between
	// them (8-, 16-, 32-, 64-wide: BPW * W = 128 when the blocks are raster neighbours) they are loaded as streaming
	// (non-temporal) loads, which keeps them from evicting the reference lines the neighbouring blocks are about to hit
	// in the vector L1: same-box A/B (profiles/r01_sad_stream_src_ab.txt) 64x64 -5 %, 64x32 -5 %, 32x32 -5 %, 16x16 -4 %,
	// 8x8 -6 %, 4-way 64x64 -3 %.  For the other widths a wave uses part of a line and the neighbour wants the rest
	// from the L1: 48x64 +11 %, 24x32 +6 %, 12x16 +8 % with streaming loads, so they keep ordinary ones.
	// (Reference rows as streaming loads, one reference: 16-25 % slower at every size -- their lines ARE reused.)
	constexpr bool STREAM_SRC = (W * BPW) % 128 == 0;
	Chunk<C> s[U];
	AlignedRefChunk<C, LPR> ra[ALIGNED ? U : 1][ALIGNED ? WAYS : 1];   // wide 4-way blocks: dword-aligned loads + funnel shift
	OverlapRefChunk<OVERLAP || OVERLAP1 ? C : 8, OVERLAP || OVERLAP1 ? LPR : 2> rv[OVERLAP || OVERLAP1 ? U : 1][OVERLAP || OVERLAP1 ? WAYS : 1];   // 4-/8-byte chunks: overlapping aligned loads
	Chunk<C> r[ALIGNED || OVERLAP || OVERLAP1 ? 1 : U][ALIGNED || OVERLAP || OVERLAP1 ? 1 : WAYS];   // everything else: loads at the byte address
	// OVERLAP1: does slot u hold the block's highest-address row?  (single trip: this lane's row in slot u is row0 + R * u)
	auto top_row = [&](int u) { return stride_ref > 0 ? row0 + R * u == (uint32_t)H - 1 : row0 + R * u == 0; };

	// slot u <- row u of the trip `ahead` trips below the current one
	auto fill_src = [&](int u, int ahead)
	{
		if constexpr (STREAM_SRC) s[u].load_stream(src + (size_t)(os + (ahead * U + u) * step_src));
		else s[u].load(src + (size_t)(os + (ahead * U + u) * step_src));
	};
	auto fill_ref = [&](int u, int ahead)
	{
#pragma unroll
		for (int k = 0; k < WAYS; ++k)
		{
			const uint32_t o = orf[k] + (ahead * U + u) * step_ref;
			if constexpr (ALIGNED) ra[u][k].load_main(ref_al, o);
			else if constexpr (OVERLAP) rv[u][k].load(ref_al, o, last);
			else if constexpr (OVERLAP1) rv[u][k].load(ref_al, o, top_row(u));
			else r[u][k].load(ref_al + (size_t)o);
		}
		if constexpr (ALIGNED)
		{
			if (last)
			{
#pragma unroll
				for (int k = 0; k < WAYS; ++k) ra[u][k].load_tail(ref_al, orf[k] + (ahead * U + u) * step_ref);
			}
		}
	};
	auto fill = [&](int u, int ahead)
	{
		fill_src(u, ahead);
		fill_ref(u, ahead);
	};
	auto drain = [&](int u)
	{
#pragma unroll
		for (int k = 0; k < WAYS; ++k)
		{
			if constexpr (ALIGNED)
			{
				Chunk<C> x;
				ra[u][k].finish((orf[k] + u * step_ref) & 3u, last, x);
				acc[k].add(s[u], x);
			}
			else if constexpr (OVERLAP)
			{
				Chunk<C> x;
				rv[u][k].finish((orf[k] + u * step_ref) & 3u, last, x);
				acc[k].add(s[u], x);
			}
			else if constexpr (OVERLAP1)
			{
				Chunk<C> x;
				rv[u][k].finish((orf[k] + u * step_ref) & 3u, top_row(u), x);
				acc[k].add(s[u], x);
			}
			else
			{
				acc[k].add(s[u], r[u][k]);
			}
		}
	};

	// Single-trip blocks issue all their source rows first and then all their reference rows (the compiler is kept from
	// interleaving them): same-box A/B, ms per 1 Mi blocks (profiles/r01_sad_prologue_order_ab.txt): 16x16 0.116 -> 0.107,
	// 4-way 16x16 0.169 -> 0.148, 32x32 0.358 -> 0.346, 8x16 0.073 -> 0.069; reference rows first does about as well.
	// Blocks of several trips lose with either order (4-way 64x64 +4 %, 16x64 +6 %, 48x64 +9 %) and 4-way 8-wide blocks
	// lose 1.5-3 %: they keep the slot-by-slot order.
	constexpr bool SPLIT_PROLOGUE = !LOOP && !(WAYS == 4 && C == 8 && LPR == 1);
	if constexpr (SPLIT_PROLOGUE)
	{
#pragma unroll
		for (int u = 0; u < U; ++u) fill_src(u, 0);
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int u = 0; u < U; ++u) fill_ref(u, 0);
		__builtin_amdgcn_sched_barrier(0);
	}
	else
	{
#pragma unroll
		for (int u = 0; u < U; ++u) fill(u, 0);
	}
	if constexpr (LOOP)
	{
		for (int t = 1; t < trips; ++t)
		{
#pragma unroll
			for (int u = 0; u < U; ++u)
			{
				drain(u);
				__builtin_amdgcn_sched_barrier(0);
				fill(u, 1);
				__builtin_amdgcn_sched_barrier(0);
			}
			os += U * step_src;
#pragma unroll
			for (int k = 0; k < WAYS; ++k) orf[k] += U * step_ref;
		}
	}
#pragma unroll
	for (int u = 0; u < U; ++u) drain(u);

#pragma unroll
	for (int k = 0; k < WAYS; ++k) total[k] = D::ROWMAJOR ? reduce_block_rowmajor<LPR, R, BPW>(acc[k].value()) : reduce_block<LPR, R>(acc[k].value());
}

// One wave's work: lane `lane` belongs to sub-block Deal::sub_block(lane) of the wave and handles block `blk` (every lane of a
// sub-block is given the same, valid, block index); only the leaders of `live` sub-blocks store.
template <int W, int C, int R, int U, int WAYS, int OP, bool LOOP, bool TIGHT>
__device__ __forceinline__ void block_cost_body(
	const uint8_t *__restrict__ src, ptrdiff_t stride_src,
	const uint8_t *__restrict__ ref, ptrdiff_t stride_ref,
	const uint32_t *__restrict__ src_off, const uint32_t *__restrict__ ref_off,
	int *__restrict__ out, int H, uint32_t lane, uint32_t blk, bool live, int scalar_first = -1)
{
	uint32_t total[WAYS];
	block_cost_sums<W, C, R, U, WAYS, OP, LOOP, TIGHT>(src, stride_src, ref, stride_ref, src_off, ref_off, H, lane, blk, 0u, total, scalar_first);
	if (live && Deal<W, C, R, WAYS>::leader(lane))
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

// One workgroup's waves of a batch of `n_blocks` equally sized blocks; `wg` = the workgroup's index within that batch's grid.
template <int W, int C, int R, int U, int WAYS, int OP, bool LOOP, bool TIGHT>
__device__ __forceinline__ void block_cost_workgroup(
	const uint8_t *__restrict__ src, ptrdiff_t stride_src,
	const uint8_t *__restrict__ ref, ptrdiff_t stride_ref,
	const uint32_t *__restrict__ src_off, const uint32_t *__restrict__ ref_off,
	int *__restrict__ out, int H, uint32_t n_blocks, uint32_t wg)
{
	typedef Tiling<W, C, R> T;
	constexpr int BPW = T::BPW;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(wg * (kThreads / kWave) + (threadIdx.x >> 6));
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t sub = Deal<W, C, R, WAYS>::sub_block(lane);
	// Lanes past the batch (or the spare lanes of a wave when 64 % LPB != 0) redo the last valid
	// block so that every lane executes the same loads and shuffles; only real leaders store.
	uint32_t blk = wave * BPW + (sub < (uint32_t)BPW ? sub : BPW - 1);
	const bool live = sub < (uint32_t)BPW && blk < n_blocks;
	if (blk >= n_blocks) blk = n_blocks - 1;
	// a wave all of whose blocks exist reads its descriptors by scalar loads (block_cost_sums); HEVCASM_SAD_VECTOR_DESC = the A/B switch
#ifndef HEVCASM_SAD_VECTOR_DESC
	const int scalar_first = (wave + 1u) * (uint32_t)BPW <= n_blocks ? (int)(wave * (uint32_t)BPW) : -1;
#else
	const int scalar_first = -1;
#endif
	block_cost_body<W, C, R, U, WAYS, OP, LOOP, TIGHT>(src, stride_src, ref, stride_ref, src_off, ref_off, out, H, lane, blk, live, scalar_first);
}

template <int W, int C, int R, int U, int WAYS, int OP, bool LOOP, bool TIGHT>
__global__ __launch_bounds__(kThreads) void block_cost_kernel(
	const uint8_t *__restrict__ src, ptrdiff_t stride_src,
	const uint8_t *__restrict__ ref, ptrdiff_t stride_ref,
	const uint32_t *__restrict__ src_off, const uint32_t *__restrict__ ref_off,
	int *__restrict__ out, int H, uint32_t n_blocks)
{
	block_cost_workgroup<W, C, R, U, WAYS, OP, LOOP, TIGHT>(src, stride_src, ref, stride_ref, src_off, ref_off, out, H, n_blocks, xcd_contiguous(blockIdx.x, gridDim.x));
}

// ---- one launch for a frame's partition: runs of equally sized blocks, any of HEVC's sizes (round 6) ---------------------------------------------
//
// The motion search of a frame scores PUs of many sizes (HEVC: 24, pred_inter.c:438-444); a launch per size is 11 - 21 us each for data that moves in a few
// microseconds.  As for the prediction kernels (pred_inter.hip, pred_runs_kernel) the host lists the frame as RUNS -- `count` consecutive descriptors of one
// size -- and one launch serves them all: a workgroup finds its run by comparing its index with the runs' first workgroups (the table travels in the kernel
// arguments) and runs the general (looping) form of the block kernel for that width: 4 rows in flight per instruction, 1 / 2 / 4 rows per lane and trip by
// what divides the height.  The size-specialised single-trip and LDS-window variants of the one-size launches are a throughput matter; a frame is latency.
constexpr int kMaxCostRuns = 96;
struct CostRunDev { uint32_t wg_count, first, count; uint8_t variant, H, pad[2]; };
struct CostRunsArgs
{
	const uint8_t *src, *ref;
	ptrdiff_t stride_src, stride_ref;
	const uint32_t *src_off, *ref_off;
	int *out;
	uint32_t wg_start[kMaxCostRuns];      // first workgroup of run i; unused entries hold 0xffffffff
	CostRunDev runs[kMaxCostRuns];
};

// widths of HEVC's partitions, in the order of the variant index
__host__ __device__ constexpr int cost_run_width(int i) { return i == 0 ? 4 : i == 1 ? 8 : i == 2 ? 12 : i == 3 ? 16 : i == 4 ? 24 : i == 5 ? 32 : i == 6 ? 48 : 64; }
constexpr int kCostRunWidths = 8;

template <int WI> constexpr int cost_run_bpw() { return Tiling<cost_run_width(WI), Plan<cost_run_width(WI)>::C, 4>::BPW; }
constexpr int kCostRunBlocksPerWave[kCostRunWidths] = { cost_run_bpw<0>(), cost_run_bpw<1>(), cost_run_bpw<2>(), cost_run_bpw<3>(), cost_run_bpw<4>(), cost_run_bpw<5>(), cost_run_bpw<6>(), cost_run_bpw<7>() };

// rows per lane and trip: 4 with one reference; 2 with four (4 would be 160 row registers in the two sets the row loop keeps: 251 VGPRs, measured)
template <int WAYS> constexpr int kCostRunMaxRowsPerTrip = WAYS == 1 ? 4 : 2;

template <int WAYS, int OP>
__global__ __launch_bounds__(kThreads) void block_cost_runs_kernel(const CostRunsArgs a)
{
	uint32_t at_or_before = 0, start = 0;
#pragma unroll
	for (int i = 0; i < kMaxCostRuns; ++i)
	{
		const bool in = a.wg_start[i] <= blockIdx.x;
		at_or_before += in ? 1u : 0u;
		start = in ? a.wg_start[i] : start;      // starts ascend: the last one taken is the run's
	}
	const CostRunDev r = a.runs[at_or_before - 1];
	const uint32_t wg = xcd_contiguous(blockIdx.x - start, r.wg_count);
	// variant = 3 * (index of the width) + log2(rows per lane and trip)
#define HEVC_RUN(WI, U) case 3 * WI + (U == 4 ? 2 : U == 2 ? 1 : 0): \
		if constexpr (U <= kCostRunMaxRowsPerTrip<WAYS>) \
			block_cost_workgroup<cost_run_width(WI), Plan<cost_run_width(WI)>::C, 4, U, WAYS, OP, true, false>(a.src, a.stride_src, a.ref, a.stride_ref, \
				a.src_off + r.first, a.ref_off + (size_t)r.first * WAYS, a.out + (size_t)r.first * WAYS, r.H, r.count, wg); \
		break;
#define HEVC_RUN_W(WI) HEVC_RUN(WI, 1) HEVC_RUN(WI, 2) HEVC_RUN(WI, 4)
	switch (r.variant)
	{
		HEVC_RUN_W(0) HEVC_RUN_W(1) HEVC_RUN_W(2) HEVC_RUN_W(3) HEVC_RUN_W(4) HEVC_RUN_W(5) HEVC_RUN_W(6) HEVC_RUN_W(7)
	default: break;
	}
#undef HEVC_RUN_W
#undef HEVC_RUN
}

// ---- 4-way SAD of wide blocks whose candidates lie close together: the union window goes through LDS ----------------
//
// The four candidates of a motion-search step usually overlap: in config #2 of the benchmark they all lie within +-7
// pixels of one position.  The generic kernel above loads each candidate's rows separately -- 1 + 4 x 2.5 vector-L1
// accesses per 64-byte block row -- and runs at the L1's access rate, not at HBM's.  This kernel
// loads the candidates' UNION window once (at most (H + 16) rows of (W + 16 + 3) bytes, dword-aligned 16-byte lane
// loads: about 2.3 accesses per window row), parks it in LDS, and lets each lane read its 16 (12) bytes of every
// candidate row from there (dword reads + v_alignbyte_b32, as the generic kernel realigns its global loads).
//   * one block per wave, 4 lanes to a block row, 16 rows per trip; the descriptors are wave-uniform, so they arrive
//     by scalar loads (not queued behind other waves' row loads) and the window geometry is scalar arithmetic;
//   * a block qualifies when all four candidates lie within 16 rows and 16 columns of each other (decided from the
//     offsets alone: offset differences are split into rows and columns with the row stride); any other block is computed
//     by the same wave with the generic code (block_cost_sums at this kernel's one-block-per-wave mapping);
//   * memory touched: 4-byte-aligned words between the lowest and the highest word that hold a byte of any of the four
//     candidates, and nothing outside that range (hevcasm_hip.h): a 16-byte piece of the window that would start below the
//     first or end above the last of them is moved inside (and its LDS position with it: its neighbour's bytes, stored twice).
template <int W, int H> struct WindowPlan
{
	static constexpr int C = W / 4;                          // bytes per lane and row in the compute phase: 4 lanes to a row (16, 12, 8)
	static constexpr int CD = C / 4;
	static constexpr int SPAN = 16;                          // candidates at most this far apart, in rows and in columns
	static constexpr int WIN_DW = (W + SPAN + 3 + 3) / 4;    // dwords of the widest window row (3 more bytes of misalignment): 21, 17, 13
	static constexpr int LROW = W == 64 ? 21 : W == 48 ? 20 : 13;   // LDS row pitch in dwords (the compute phase's reads: 2-way / none / 2-way bank conflicts)
	// (64-wide at a pitch of 24 with three aligned ds_read_b64 per lane, row and candidate -- conflict-free: 12 y + 2 j mod 32 -- measured
	// 3 - 6 % SLOWER than the 2-way conflicts of dword reads at 21: the bank conflicts are not what holds this kernel; profiles/r02_sad4_window_ab.txt)
	static constexpr int LPRW = (WIN_DW + 3) / 4;            // staging: lanes (16 bytes each) per window row: 6, 5, 4
	static constexpr int ROWS = H + SPAN;
	static constexpr int NI = (ROWS * LPRW + kWave - 1) / kWave;   // staging instructions: lane l of instruction i holds piece 64 i + l of the window
	static constexpr int TRIPS = H / 16;
	static_assert(H % 16 == 0 && WIN_DW <= LROW && LPRW * 4 >= WIN_DW, "window row must fit its LDS row and its staging lanes");
};

// Where one block's four candidates lie: wave-uniform, so all of this is scalar loads and scalar arithmetic.
struct WindowGeometry
{
	bool close;            // all four candidates within SPAN rows and SPAN columns of each other
	uint32_t so;           // source offset
	uint32_t a_min, span;  // the lowest aligned word that holds a byte of a candidate (offset from the word-aligned plane base) and the
	                       // distance from it to the last 16-byte piece that ends inside the highest such word
	int rel0;              // window origin (top-left corner, word-aligned) relative to a_min: -(SPAN + 3) .. 0
	uint32_t last_row;     // last window row = H - 1 + the largest row displacement
	uint32_t X[4], Y[4];   // candidate k starts at byte X[k] (0 .. SPAN + 3) of window row Y[k] (0 .. SPAN)
};

template <int W, int H>
__device__ __forceinline__ WindowGeometry window_geometry(const uint32_t *__restrict__ src_off, const uint32_t *__restrict__ ref_off,
		uint32_t blk, uint32_t ref_mis, uint32_t sr32, uint32_t inv_stride)
{
	typedef WindowPlan<W, H> P;
	WindowGeometry g;
	g.so = src_off[blk];
	uint32_t o[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) o[k] = ref_off[(size_t)blk * 4 + k] + ref_mis;
	const uint32_t m = min(min(o[0], o[1]), min(o[2], o[3]));
	bool close = true;
	int dx[4];
#pragma unroll
	for (int k = 0; k < 4; ++k)
	{
		const uint32_t d = o[k] - m;                                 // candidate k lies d bytes above the lowest one:
		close = close && d <= (uint32_t)P::SPAN * sr32 + 32u;       // dy rows and dx columns, dx in [-32, stride - 32)
		const uint32_t nn = d + 32u;
		uint32_t q = __umulhi(nn, inv_stride);                      // nn / stride, or one more
		int r = (int)(nn - q * sr32);
		if (r < 0) { q -= 1; r += (int)sr32; }
		dx[k] = r - 32;
		g.Y[k] = q;
	}
	const int xmin = min(min(dx[0], dx[1]), min(dx[2], dx[3])), xmax = max(max(dx[0], dx[1]), max(dx[2], dx[3]));
	const uint32_t ymax = max(max(g.Y[0], g.Y[1]), max(g.Y[2], g.Y[3]));
	g.close = close && (xmax - xmin) <= P::SPAN && ymax <= (uint32_t)P::SPAN && inv_stride != 0u;
	const uint32_t wo = m + (uint32_t)xmin;          // byte offset of the window's top-left corner; modulo 2^32 like everything derived from it
	const uint32_t wbase = wo & ~3u;
#pragma unroll
	for (int k = 0; k < 4; ++k) g.X[k] = (uint32_t)(dx[k] - xmin) + (wo & 3u);
	g.last_row = (uint32_t)H + ymax - 1u;
	g.a_min = m & ~3u;
	const uint32_t a_max = (max(max(o[0], o[1]), max(o[2], o[3])) + (uint32_t)(H - 1) * sr32 + (uint32_t)(W - 1)) & ~3u;
	g.span = a_max - 12u - g.a_min;
	g.rel0 = (int)(wbase - g.a_min);
	return g;
}

// Issue every load of one block: the source rows and the window pieces (global -> registers; lane l of instruction i holds piece
// 64 i + l of the window, LPRW pieces of 16 bytes per row, the last piece of a row pulled in to end with the LDS row).
// No lane is ever switched off: a piece past the window's last row repeats that row, a piece below a_min / above a_max
// repeats its neighbour (same bytes to the same LDS words) -- nothing outside the candidates' own address range is read.
template <int W, int H>
__device__ __forceinline__ void window_issue(const WindowGeometry &g, const uint8_t *ref_al, const uint8_t *src, uint32_t sr32, uint32_t ss32, uint32_t lane,
		u32x4 (&st)[WindowPlan<W, H>::NI], uint32_t (&spos)[WindowPlan<W, H>::NI], Chunk<WindowPlan<W, H>::C> (&s)[WindowPlan<W, H>::TRIPS])
{
	typedef WindowPlan<W, H> P;
	// the source rows go first: loads return in issue order, and the first trip of the arithmetic starts as soon as ITS window rows have
	// landed (sad4_window_kernel), which must not have to wait for loads issued behind them
	const uint32_t y0 = lane >> 2, j = lane & 3u;
	auto load_src = [&]()
	{
#pragma unroll
		for (int t = 0; t < P::TRIPS; ++t)
		{
			const uint8_t *p = src + (size_t)(g.so + ((uint32_t)(t * 16) + y0) * ss32 + j * (uint32_t)P::C);
			if (W != 48) s[t].load_stream(p); else s[t].load(p);       // as the generic kernel: streaming where waves cover whole lines
		}
	};
	constexpr bool SRC_FIRST = true;
	if (SRC_FIRST) load_src();
#pragma unroll
	for (int i = 0; i < P::NI; ++i)
	{
		const uint32_t piece = (uint32_t)(i * kWave) + lane, srow = piece / P::LPRW;
		const uint32_t scol = min((piece - srow * P::LPRW) * 4u, (uint32_t)(P::WIN_DW - 4));
		const uint32_t row = min(srow, g.last_row);
		// the piece's address relative to a_min (a small number of either sign on row 0, positive below it: stride <= 2^20),
		// moved inside [0, span]; rows are word-aligned with each other (stride % 4 == 0: window_stride_ok)
		const int rel = (int)(__umul24(row, sr32) + 4u * scol) + g.rel0;
		const int rel_in = med3_i32(rel, 0, (int)g.span);
		st[i] = reinterpret_cast<const U4 *>(ref_al + (size_t)(g.a_min + (uint32_t)rel_in))->v;
		spos[i] = (__umul24(row, (uint32_t)P::LROW) + scol + ((uint32_t)(rel_in - rel) >> 2)) & 0x3fffffffu;     // (rel_in - rel): a small multiple of 4 of either sign
	}
	if (!SRC_FIRST) load_src();
}

// One trip (16 block rows) of the four SADs: candidates read from LDS (the trip's row offset is an immediate of the ds_read).
template <int W, int H>
__device__ __forceinline__ void window_trip(const WindowGeometry &g, const uint32_t *const (&rk)[4], int t,
		const Chunk<WindowPlan<W, H>::C> &s, uint32_t (&acc)[4])
{
	typedef WindowPlan<W, H> P;
#pragma unroll
	for (int k = 0; k < 4; ++k)
	{
		const uint32_t *r = rk[k] + t * 16 * P::LROW;
		uint32_t w[P::CD + 1];
#pragma unroll
		for (int q = 0; q <= P::CD; ++q) w[q] = r[q];
		const uint32_t sh = g.X[k] & 3u;
#pragma unroll
		for (int q = 0; q < P::CD; ++q)
			acc[k] = __builtin_amdgcn_sad_u8(s.d[q], __builtin_amdgcn_alignbyte(w[q + 1], w[q], sh), acc[k]);
	}
}

// One block per wave.  (More than one -- the next block's loads in flight while this one is computed -- measured slower: 4 consecutive
// blocks per wave +34 %, 4 blocks dealt round-robin over the workgroup's waves +11 %: many short waves on NEIGHBOURING blocks, which
// share the 128-byte lines of their source rows and of the overlap of their windows, beat fewer, deeper ones.  And every branch
// between a load and its use makes the compiler's wait counts conservative: a conditionally issued prefetch in a loop made every
// compute phase wait for ALL loads in flight, 1.3 - 2.2x slower.  profiles/r02_sad4_window_ab.txt)
template <int W, int H>
__global__ __launch_bounds__(kThreads) void sad4_window_kernel(
	const uint8_t *__restrict__ src, ptrdiff_t stride_src,
	const uint8_t *__restrict__ ref, ptrdiff_t stride_ref,
	const uint32_t *__restrict__ src_off, const uint32_t *__restrict__ ref_off,
	int *__restrict__ out, uint32_t n_blocks, uint32_t inv_stride /* floor(2^32 / stride_ref) + 1; 0: every block takes the generic code */)
{
	typedef WindowPlan<W, H> P;
	// 64 rows: the LDS allocation is rounded up to 40 KiB, i.e. four workgroups per CU instead of five (same-box sweep at 64x64 / 48x64:
	// five +2 %, three +13 %, two +34 % against four; 48 rows prefer their natural six; profiles/r02_sad4_window_ab.txt)
	// (32 / 48 rows rounded up to 32 KiB -- five workgroups per CU instead of seven: close candidates +7 ... +8 %, far ones -3 ... -10 %: left alone)
	__shared__ uint32_t lds_all[H == 64 ? 10240 : (kThreads / kWave) * P::ROWS * P::LROW];

	// the four waves of a workgroup are on four neighbouring blocks
	const uint32_t wg = xcd_contiguous(blockIdx.x, gridDim.x);
	const uint32_t wave_in_wg = threadIdx.x >> 6;
	const uint32_t blk = __builtin_amdgcn_readfirstlane(wg * (kThreads / kWave) + wave_in_wg);
	if (blk >= n_blocks) return;
	const uint32_t lane = threadIdx.x & 63;
	uint32_t *lds = lds_all + wave_in_wg * (P::ROWS * P::LROW);

	const uint32_t ref_mis = (uint32_t)(reinterpret_cast<uintptr_t>(ref) & 3u);
	const uint8_t *const ref_al = ref - ref_mis;
	const uint32_t sr32 = (uint32_t)stride_ref, ss32 = (uint32_t)stride_src;

	const WindowGeometry g = window_geometry<W, H>(src_off, ref_off, blk, ref_mis, sr32, inv_stride);
	if (g.close)
	{
		u32x4 st[P::NI];
		uint32_t spos[P::NI];
		Chunk<P::C> s[P::TRIPS];
		window_issue<W, H>(g, ref_al, src, sr32, ss32, lane, st, spos, s);

		const uint32_t y0 = lane >> 2, j = lane & 3u;
		const uint32_t *rk[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) rk[k] = lds + (__umul24(y0 + g.Y[k], (uint32_t)P::LROW) + (g.X[k] >> 2) + j * (uint32_t)P::CD);
		uint32_t acc[4] = { 0, 0, 0, 0 };
		// Trip t reads window rows up to 16 t + 15 + SPAN: as soon as the pieces that hold them have landed they are parked in LDS and the
		// trip runs, while the rest of the window is still on its way (straight-line code: the wait counts are exact).
		int parked = 0;
#pragma unroll
		for (int t = 0; t < P::TRIPS; ++t)
		{
			// (16 / 32 rows: -9 % / -4 % against parking the whole window first; 48 / 64 rows: +1 ... +5 %, so they park everything at once)
			constexpr bool PIPELINED = H <= 32;
			const int need = PIPELINED ? min(P::NI, ((t * 16 + 16 + P::SPAN) * P::LPRW + kWave - 1) / kWave) : P::NI;   // staging instructions that cover those rows
#pragma unroll
			for (int i = 0; i < P::NI; ++i)
			{
				if (i >= parked && i < need)
				{
					uint32_t *d = lds + spos[i];
					d[0] = st[i].x; d[1] = st[i].y; d[2] = st[i].z; d[3] = st[i].w;
				}
			}
			parked = need;
			// the wave's own LDS accesses are in order (one wave, one LDS queue); keep the compiler from reordering them
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			window_trip<W, H>(g, rk, t, s[t], acc);
			__builtin_amdgcn_sched_barrier(0);    // one trip's LDS reads live at a time
		}
		uint32_t total[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) total[k] = wave_sum_to_lane63(acc[k]);
		if (lane == 63)
		{
			u32x4 v; v.x = total[0]; v.y = total[1]; v.z = total[2]; v.w = total[3];
			reinterpret_cast<U4 *>(out + (size_t)blk * 4)->v = v;
		}
		return;
	}
	// Candidates far apart: the generic code at this kernel's mapping (one block per wave: 16 rows in flight, a single trip).
	// (64 rows in two parts of 2 rows per lane: as one part of 4 this path needs 161 registers and takes the window path's occupancy
	// from 5 waves per SIMD to 3: +6 % on close candidates.  48 rows as one part of 3: 124 registers, 4 waves instead of 5 -- close
	// candidates +4 %, far ones -13 %; holding that to 96 registers with launch bounds spills: close +15 %, far +46 %.)
	uint32_t sum[4] = { 0, 0, 0, 0 }, part[4];
	if constexpr (H == 64)
	{
		block_cost_sums<W, P::C, 16, 2, 4, OP_SAD, false, false>(src, stride_src, ref, stride_ref, src_off, ref_off, 32, lane, blk, 0u, sum);
		block_cost_sums<W, P::C, 16, 2, 4, OP_SAD, false, false>(src, stride_src, ref, stride_ref, src_off, ref_off, 32, lane, blk, 32u, part);
#pragma unroll
		for (int k = 0; k < 4; ++k) sum[k] += part[k];
	}
	else if constexpr (H == 48)
	{
		block_cost_sums<W, P::C, 16, 2, 4, OP_SAD, false, false>(src, stride_src, ref, stride_ref, src_off, ref_off, 32, lane, blk, 0u, sum);
		block_cost_sums<W, P::C, 16, 1, 4, OP_SAD, false, false>(src, stride_src, ref, stride_ref, src_off, ref_off, 16, lane, blk, 32u, part);
#pragma unroll
		for (int k = 0; k < 4; ++k) sum[k] += part[k];
	}
	else
	{
		block_cost_sums<W, P::C, 16, H / 16, 4, OP_SAD, false, false>(src, stride_src, ref, stride_ref, src_off, ref_off, H, lane, blk, 0u, sum);
	}
	if (lane == 0)
	{
		u32x4 v; v.x = sum[0]; v.y = sum[1]; v.z = sum[2]; v.w = sum[3];
		reinterpret_cast<U4 *>(out + (size_t)blk * 4)->v = v;
	}
}

// (Round 4: narrow 4-way blocks through ONE union window per wave of 8 blocks, staged by LDS-DMA -- sad4_strip_kernel -- was built in three forms,
// bit-exact, and is 2 ... 35 % slower than the generic kernel at every size but 16x32 (-2 %): profiles/r04_sad4_strip_window_ab.txt.  A window wave's life
// is serial -- descriptors, geometry, DMA, barrier, LDS reads -- where the generic wave's loads all go out at once.  Source: commit e9488f1.)

struct Args
{
	const uint8_t *src; ptrdiff_t ss; const uint8_t *ref; ptrdiff_t sr;
	const uint32_t *so; const uint32_t *ro; int *out; int H; size_t n; hipStream_t st;
};

// strides for which sad4_window_kernel applies: its row / column split of an offset difference is exact and unambiguous, and
// the rows of a window are word-aligned with each other (so that "the aligned words that hold a candidate's bytes" means the
// same in every row)
static bool window_stride_ok(ptrdiff_t sr, int W) { return sr >= W + 64 && sr <= (1 << 20) && (sr & 3) == 0; }

template <int W> static int launch_window(const Args &a)
{
	static const bool no_window = getenv("HEVCASM_SAD4_NOWINDOW") != nullptr;     // measurement switch, read once per process
	const uint32_t inv = no_window ? 0u : (uint32_t)((1ull << 32) / (uint64_t)a.sr) + 1u;    // (0: every block takes the generic code)
	const uint32_t grid = ceil_div(a.n, kThreads / kWave);
#define WINDOW(HM) hipLaunchKernelGGL((sad4_window_kernel<W, HM>), dim3(grid), dim3(kThreads), 0, a.st, a.src, a.ss, a.ref, a.sr, a.so, a.ro, a.out, (uint32_t)a.n, inv)
	if (a.H == 16) WINDOW(16); else if (a.H == 32) WINDOW(32); else if (a.H == 48) WINDOW(48); else WINDOW(64);
#undef WINDOW
	return launch_status("sad4_window_kernel");
}

template <int W, int C, int R, int U, int WAYS, int OP, bool TIGHT> static int launch_cfg(const Args &a)
{
	constexpr int blocks_per_wg = Tiling<W, C, R>::BPW * (kThreads / kWave);
	const uint32_t grid = ceil_div(a.n, blocks_per_wg);
	// 64- and 48-wide blocks of 16 / 32 / 48 / 64 rows go through sad4_window_kernel, which decides per block: candidates close together ->
	// union window through LDS (same-box A/B of the shipped form against the round-1 generic kernel, config #2's +-7 pixels: 64x64 -6 %,
	// 64x48 -3 %, 64x32 -2 %, 64x16 -8 %, 48x64 -10 %: DESIGN.md section 0); far apart -> generic code, one block per wave (-10 ... +14 %).
	// 32-wide blocks stay here: their window path measured +2 ... +21 % on close candidates (the window is 1.6x the block's width) although
	// the far path gained 15 % (profiles/r02_sad4_window_ab.txt).
	static const bool generic_only = getenv("HEVCASM_SAD4_GENERIC") != nullptr;    // measurement switch, read once per process
	if constexpr (WAYS == 4 && OP == OP_SAD && (W == 64 || W == 48) && !TIGHT && U <= 6)
	{
		if (a.H % 16 == 0 && a.H <= 64 && window_stride_ok(a.sr, W) && !generic_only)
		{
			return launch_window<W>(a);
		}
	}
	if (a.H == R * U)
		hipLaunchKernelGGL((block_cost_kernel<W, C, R, U, WAYS, OP, false, TIGHT>), dim3(grid), dim3(kThreads), 0, a.st,
				a.src, a.ss, a.ref, a.sr, a.so, a.ro, a.out, a.H, (uint32_t)a.n);
	else if constexpr (U <= 4)      // the deep unrolls are only ever picked for a single trip: no looping variant of them is built
		hipLaunchKernelGGL((block_cost_kernel<W, C, R, U, WAYS, OP, true, TIGHT>), dim3(grid), dim3(kThreads), 0, a.st,
				a.src, a.ss, a.ref, a.sr, a.so, a.ro, a.out, a.H, (uint32_t)a.n);
	else
		return set_error(HEVCASM_HIP_EINVAL, "internal: single-trip configuration chosen for a multi-trip height");
	return launch_status("block_cost_kernel");
}

// rows per lane = H / R; take the deepest unroll that divides it
template <int W, int C, int R, int WAYS, int OP, bool TIGHT> static int launch_rows(const Args &a)
{
	const int rows_per_lane = a.H / R;
	// one reference, 8 rows per lane: one trip of 8 (source rows first, then reference rows) instead of two trips of 4 --
	// 16x32 -8 %, 8x32 -9 %, 24x32 -7 %, 32x64 -4 %, 64x64 -1 %
	// (four references: 8-byte chunks only -- 16x32 -6 %, 24x32 -6 %; 64x64 and 32x64 the same either way, 48x64 +6 %)
	if constexpr (WAYS == 1 || C <= 8 || W == 12) { if (rows_per_lane == 8) return launch_cfg<W, C, R, 8, WAYS, OP, TIGHT>(a); }    // (12x32 four references as one trip: -39 %)
	if (rows_per_lane == 6) return launch_cfg<W, C, R, 6, WAYS, OP, TIGHT>(a);                                       // 64x48: -2 % (one and four references)
	if constexpr (WAYS == 1 && (C <= 8 || W == 12)) { if (rows_per_lane == 16) return launch_cfg<W, C, R, 16, WAYS, OP, TIGHT>(a); }    // 16x64: -10 %, 12x64: -18 % -> -32 %
	if (rows_per_lane % 4 == 0) return launch_cfg<W, C, R, 4, WAYS, OP, TIGHT>(a);
	if (rows_per_lane % 3 == 0) return launch_cfg<W, C, R, 3, WAYS, OP, TIGHT>(a);   // heights 12, 24, 48
	if (rows_per_lane % 2 == 0) return launch_cfg<W, C, R, 2, WAYS, OP, TIGHT>(a);
	return launch_cfg<W, C, R, 1, WAYS, OP, TIGHT>(a);
}

template <int W, int WAYS, int OP> static int launch_w(const Args &a)
{
	typedef Plan<W> P;
	if constexpr (W == 8)
	{
		if (a.sr < 8 && a.sr > -8) return launch_rows<W, P::C, P::R_SAFE, WAYS, OP, true>(a);
	}
	if constexpr (W == 12)
	{
		if (a.sr < 12 && a.sr > -12) return launch_rows<W, 4, 4, WAYS, OP, false>(a);     // overlapping rows: three lanes of 4 bytes per row (dword-aligned 8-byte loads that overlap between the lanes of a row, never between rows)
		// four references, heights that are not one trip of 1 / 2 / 3 / 4 / 6 / 8 rows per lane: the aligned loads of the one-lane-per-row mapping need a
		// single trip (they must know which row is the block's highest), and at the byte address it is 3 - 6 % slower than three lanes of 4 bytes
		const int rpl = a.H / 4;
		if (WAYS == 4 && !(rpl <= 4 || rpl == 6 || rpl == 8) && a.H != 48 && a.H != 64) return launch_rows<W, 4, 4, WAYS, OP, false>(a);
	}
	// (one reference, 64-/48-/32-wide, 16 rows in flight and one block per wave -- the mapping that wins for four far-apart candidates:
	// +18 ... +27 % slower on all seven sizes tried, round 2: profiles/r02_sad4_window_ab.txt)
	// four references, 16x64: 8 rows in flight make it a single trip of 8 (-3 %)
	// one reference, row-major 8- and 16-wide (see Deal), 8 rows: two rows in flight per instruction and twice the blocks per wave
	// (a 256-byte run per frame row): 8x8 -6.4 %, 16x8 -6.1 %, 12x8 -20 %, 24x8 -4 %, 32x8 -7 %; taller blocks +-2 % (16x64 +32 %), eight rows in flight
	// +14 ... +29 %; with four references it loses 2 ... 9 % (profiles/r02_sad_rowmajor_ab.txt)
	if constexpr (WAYS == 1 && (W == 8 || W == 12 || W == 16 || W == 24 || W == 32)) { if (a.H == 8) return launch_rows<W, P::C == 16 ? 8 : P::C, 2, WAYS, OP, false>(a); }
	// 4 rows: two in flight as well (one reference 16x4 -18 %, 8x4 -12 %, 4x4 -10 %; four references -3 ... -7 %; 12x4 +8 % with it, so not there)
	if constexpr (W == 4 || W == 8 || W == 16) { if (a.H == 4) return launch_rows<W, P::C, 2, WAYS, OP, false>(a); }
	// four references, 32-wide, up to 32 rows: four lanes of 8 bytes (aligned 12-byte loads, no tail load), 4 rows in flight: 32x8 -6 %, 32x16 -9 %, 32x24 -5.5 %,
	// 32x32 -5 % against two lanes of 16 bytes + tail load (which 32x64 keeps: +9 % with this; 8 rows in flight +1 ... +14 %) -- profiles/r02_sad_rowmajor_ab.txt
	if constexpr (WAYS == 4 && W == 32) { if (a.H <= 32) return launch_rows<W, 8, 4, WAYS, OP, false>(a); }
	// (one reference, 32-wide, the same mapping: -0.7 ... -2.4 % on all five heights, 8 rows in flight +14 ... +36 %)
	if constexpr (WAYS == 1 && W == 32) return launch_rows<W, 8, 4, WAYS, OP, false>(a);
	// four references, one lane to a row (8- and 12-wide), 48 or 64 rows: 8 rows in flight make the block ONE trip of 6 or 8 rows per lane, which the aligned loads of
	// that mapping need (see OVERLAP1): 8x64 -29 % against four trips at the byte address (one reference and shorter blocks lose 4 ... 27 % with 8 rows in flight)
	if constexpr (WAYS == 4 && (W == 8 || W == 12)) { if (a.H == 48 || a.H == 64) return launch_rows<W, P::C, 8, WAYS, OP, false>(a); }
	if constexpr (WAYS == 4 && W == 16) { if (a.H == 64) return launch_rows<W, P::C, 8, WAYS, OP, false>(a); }
	// four references, 24-wide, 32 rows or more: 8 rows in flight (24x32 -4 % -> -12 %, 24x64 -3 % -> -14 % against round 1; shorter blocks and one reference lose with it)
	if constexpr (WAYS == 4 && W == 24) { if (a.H >= 32 && a.H % 8 == 0) return launch_rows<W, P::C, 8, WAYS, OP, false>(a); }
	if (P::R_PREF != P::R_SAFE && a.H % P::R_PREF == 0) return launch_rows<W, P::C, P::R_PREF, WAYS, OP, false>(a);
	return launch_rows<W, P::C, P::R_SAFE, WAYS, OP, false>(a);
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
	const Args a = { src, ss, ref, sr, so, ro, out, H, n, as_stream(stream) };
	switch (W)
	{
#define CASE(w) case w: return launch_w<w, WAYS, OP>(a);
		CASE(4) CASE(8) CASE(12) CASE(16) CASE(20) CASE(24) CASE(28) CASE(32)
		CASE(36) CASE(40) CASE(44) CASE(48) CASE(52) CASE(56) CASE(60) CASE(64)
#undef CASE
	}
	return set_error(HEVCASM_HIP_EINVAL, "unsupported width");
}

// host side of the runs entry points: the run table, each run validated like a one-size call
template <int WAYS, int OP>
static int launch_runs(const uint8_t *src, ptrdiff_t ss, const uint8_t *ref, ptrdiff_t sr,
		const uint32_t *so, const uint32_t *ro, int *out, const hevcasm_hip_cost_run *runs, int n_runs, void *stream)
{
	if (int e = check_device()) return e;
	if (n_runs < 0) return set_error(HEVCASM_HIP_EINVAL, "negative run count");
	if (n_runs == 0) return 0;
	if (!runs || !src || !ref || !so || !ro || !out) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	std::vector<CostRunDev> table;
	table.reserve((size_t)n_runs);
	for (int i = 0; i < n_runs; ++i)
	{
		const int W = (int)(runs[i].rect >> 8), H = (int)(runs[i].rect & 0xff);
		int wi = -1;
		for (int j = 0; j < kCostRunWidths; ++j) if (cost_run_width(j) == W) wi = j;
		if (wi < 0 || (runs[i].rect >> 16)) return set_error(HEVCASM_HIP_EINVAL, "run width must be one of HEVC's: 4, 8, 12, 16, 24, 32, 48, 64 (other widths: the one-size batch calls)");
		if (H < 4 || H > 64 || (H & 3)) return set_error(HEVCASM_HIP_EINVAL, "block size must be a multiple of 4 in 4..64");
		if ((uint64_t)runs[i].first + runs[i].count > 0xffffffffull) return set_error(HEVCASM_HIP_EINVAL, "run exceeds 2^32-1 descriptors");
		if (runs[i].count == 0) continue;
		CostRunDev d = {};
		d.first = runs[i].first; d.count = runs[i].count; d.H = (uint8_t)H;
		d.variant = (uint8_t)(3 * wi + (H % 16 == 0 && kCostRunMaxRowsPerTrip<WAYS> >= 4 ? 2 : H % 8 == 0 ? 1 : 0));
		const uint32_t blocks_per_wg = (uint32_t)kCostRunBlocksPerWave[wi] * (kThreads / kWave);
		d.wg_count = ceil_div((size_t)d.count, (size_t)blocks_per_wg);
		table.push_back(d);
	}
	// the runs are dealt to the grid TALLEST FIRST: the launch ends when its last wave does, and a 64-row block that starts in the last generation of
	// workgroups would be the whole tail
	std::stable_sort(table.begin(), table.end(), [](const CostRunDev &x, const CostRunDev &y) { return x.H > y.H; });
	for (size_t base = 0; base < table.size(); base += kMaxCostRuns)      // more runs than one table holds: the next launch
	{
		CostRunsArgs a = {};
		a.src = src; a.ref = ref; a.stride_src = ss; a.stride_ref = sr; a.src_off = so; a.ref_off = ro; a.out = out;
		uint32_t wgs = 0, used = 0;
		for (int i = 0; i < kMaxCostRuns; ++i) a.wg_start[i] = 0xffffffffu;
		for (size_t i = base; i < table.size() && i < base + kMaxCostRuns; ++i)
		{
			a.runs[used] = table[i];
			a.wg_start[used++] = wgs;
			wgs += table[i].wg_count;
		}
		hipLaunchKernelGGL((block_cost_runs_kernel<WAYS, OP>), dim3(wgs), dim3(kThreads), 0, as_stream(stream), a);
		if (int e = launch_status("block_cost_runs_kernel")) return e;
	}
	return 0;
}

}  // namespace hevc

extern "C" int hevcasm_hip_sad_runs(
	const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref,
	const uint32_t *src_off, const uint32_t *ref_off, int *sad, const hevcasm_hip_cost_run *runs, int n_runs, void *stream)
{
	return hevc::launch_runs<1, hevc::OP_SAD>(src, stride_src, ref, stride_ref, src_off, ref_off, sad, runs, n_runs, stream);
}

extern "C" int hevcasm_hip_sad_multiref_runs(
	const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref,
	const uint32_t *src_off, const uint32_t *ref_off, int *sad, const hevcasm_hip_cost_run *runs, int n_runs, void *stream)
{
	return hevc::launch_runs<4, hevc::OP_SAD>(src, stride_src, ref, stride_ref, src_off, ref_off, sad, runs, n_runs, stream);
}

extern "C" int hevcasm_hip_ssd_runs(
	const uint8_t *srcA, ptrdiff_t stride_srcA, const uint8_t *srcB, ptrdiff_t stride_srcB,
	const uint32_t *offA, const uint32_t *offB, int *ssd, const hevcasm_hip_cost_run *runs, int n_runs, void *stream)
{
	return hevc::launch_runs<1, hevc::OP_SSD>(srcA, stride_srcA, srcB, stride_srcB, offA, offB, ssd, runs, n_runs, stream);
}

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
