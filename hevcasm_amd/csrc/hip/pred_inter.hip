// pred_inter.hip -- HEVC uni- and bi-directional inter prediction (motion compensation), 8-bit.
//
// Replaces hevcasm_pred_uni_8to8 (reference pred_inter.h:56; C bodies pred_inter.c:141-228) and
// hevcasm_pred_bi_8to8 (pred_inter.h:76; C body pred_inter.c:490-530) for whole batches of
// equally sized prediction units (PUs) with per-PU motion-vector fractions.
//
// Arithmetic.  The 2-D filter is separable and the reference keeps the first pass at full
// precision (shift 0, pred_inter.c:175), so the pass order can be swapped without changing a bit:
// the sum S = sum_j sum_i cy[j] cx[i] s[y+j][x+i] is exact either way.  One path serves every
// fraction pair (an unfiltered direction is the tap {64}):
//   uni: (S + 2048) >> 12, clip to 8 bits  == the reference's copy / H-only / V-only / HV cases,
//        because 64 * (s + 32) == 64 s + 2048 makes the single-direction rounding identical;
//   bi:  p = S >> 6 per reference (pred_inter.c:515: shift 6, no rounding), then
//        (p0 + p1 + 64) >> 7, clip (pred_inter.c:490-501).
//
// Mapping: everything stays in registers, no LDS.  Lane j of a PU owns the 4 window columns
// u = 4j .. 4j+3 (window = the W + T - 1 columns the filter reads, starting BEFORE = T/2 - 1
// left of the block) and produces output columns 4j .. 4j+3.  It streams down the rows with one
// dword load per row, runs the VERTICAL filter first on the bytes unpacked to packed int16
// (v_pk_mad_u16, partial sums stay below 2^15: |sum| <= 88 * 255), then takes its two right-hand
// neighbours' vertical results with DPP wave shifts and runs the HORIZONTAL filter with
// v_dot2_i32_i16 on (even, even) / (odd, odd) column pairs.  A sliding window of T unpacked rows
// lives in registers (loop unrolled by T so the slot indices are static).  ceil((W + T - 1) / 4)
// lanes per PU, as many PUs per wave as fit.
//
// Footprint.  The reference reads the filter margins only in a direction whose fraction is non-zero:
// W x H for an integer motion vector (memcpy, pred_inter.c:141-151), (W + T - 1) x H for a horizontal-only,
// W x (H + T - 1) for a vertical-only and (W + T - 1) x (H + T - 1) for a two-dimensional fraction
// (pred_inter.c:154-228); its x86 bi-prediction kernels do the same per reference.  This kernel reads exactly that
// rectangle per reference and nothing outside it: for a zero fraction the window of that direction starts at
// the block (the tap {64} sits at filter position 0) and ends with it -- a lane whose load would pass the window's
// last column is moved left to end there (and its bytes shifted back in registers), a row stream that
// reaches the block's last row stays on it (the rows "below" only ever meet zero coefficients).  Blocks narrower
// than a lane load (W < 4) with xFrac = 0 are loaded byte by byte.  Writes exactly W x H.
// Algorithmic bytes per PU: refs * (W + T - 1)(H + T - 1) + W * H (+ 4 (1 + refs) + refs descriptor bytes).

#include "common.hpp"
#include <algorithm>
#include <cstdlib>
#include <vector>

namespace hevc {

// Diagnostic build only (-DPRED_STAMPS, tools/pred_timeline.py): per-wave s_memtime stamps into a buffer that nothing else reads
#ifdef PRED_STAMPS
__device__ unsigned long long *g_pred_stamps = nullptr;
#define STAMP(i) do { if (stamps && lane == 0) stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

typedef short s16x2 __attribute__((ext_vector_type(2)));

// -DHEVCASM_PRED_PATTERN_ONLY (make pattern -> lib/pattern/libhevcasm_hip.so; bench.py's <key>_pattern_ms): the SAME kernels with the arithmetic taken out --
// every load (address, width, order, rows in flight), every point where a loaded row is first touched, every store (address, width) stays; the filters, the
// cross-lane exchanges and the packing are replaced by an XOR that keeps the loaded registers alive.  What such a build takes is a floor of the shipped
// kernel BY CONSTRUCTION (it does strictly less per wave), provided it runs at the shipped kernel's occupancy: without the filters it needs fewer registers,
// so the four kernels the bench rows use are held to their real waves per SIMD by a dummy LDS allocation (occupancy_pad: 4 workgroups of 4 waves per CU, or 2).
#ifdef HEVCASM_PRED_PATTERN_ONLY
template <int WAVES_PER_SIMD> __device__ __forceinline__ void occupancy_pad(uint32_t never_true)
{
	if constexpr (WAVES_PER_SIMD < 8)
	{
		constexpr int BYTES = 163840 / WAVES_PER_SIMD < 65536 ? 163840 / WAVES_PER_SIMD : 65536;      // floor(160 KiB / BYTES) == WAVES_PER_SIMD workgroups per CU
		__shared__ uint8_t pad[BYTES];
		if (never_true == 0xffffffffu) pad[threadIdx.x] = 1;      // (n_blocks is never 2^32 - 1: the allocation must stay referenced)
	}
}
#endif

// Filter taps (HEVC 8.5.3.3.3: luma quarter-pel, chroma eighth-pel).  Every tap fits a signed byte, so tap k of all phases is one
// packed literal -- luma: 4 phases in a dword; chroma: 8 phases in two -- and a lane picks its phase with one v_bfe_i32: no table
// in memory, no load (the coefficient loads used to be a second dependent memory round trip in every wave's prologue, behind the
// descriptors and in front of the first window rows).  Phase 0 is the single tap {64} at filter position 0 (see Footprint).
struct PredTaps
{
	uint32_t luma[8];          // luma[k]: byte f = tap k of phase f
	uint32_t chroma[4][2];     // chroma[k][h]: byte f & 3 = tap k of phase 4 h + (f & 3)
};

constexpr PredTaps make_pred_taps()
{
	constexpr int luma[4][8] = {
		{ 64, 0, 0, 0, 0, 0, 0, 0 } /* window starts at the block */, { -1, 4, -10, 58, 17, -5, 1, 0 },
		{ -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
	constexpr int chroma[8][4] = {
		{ 64, 0, 0, 0 } /* window starts at the block */, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
		{ -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };
	PredTaps t = {};
	for (int k = 0; k < 8; ++k)
		for (int f = 0; f < 4; ++f) t.luma[k] |= ((uint32_t)luma[f][k] & 0xffu) << (8 * f);
	for (int k = 0; k < 4; ++k)
		for (int f = 0; f < 8; ++f) t.chroma[k][f >> 2] |= ((uint32_t)chroma[f][k] & 0xffu) << (8 * (f & 3));
	return t;
}

static constexpr PredTaps kTaps = make_pred_taps();

__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c)
{
	const s16x2 r = __builtin_bit_cast(s16x2, a) * __builtin_bit_cast(s16x2, b) + __builtin_bit_cast(s16x2, c);
	return __builtin_bit_cast(uint32_t, r);
}
// the same as single instructions the compiler may not take apart: pk_mad with a register addend (left to itself it moves the seed of a
// sum to the end as a separate v_pk_add), and the non-accumulating dot product that starts a chain (v_dot2c needs its start value moved
// into the destination first)
__device__ __forceinline__ uint32_t pk_mad_fused(uint32_t a, uint32_t b, uint32_t c)
{
	uint32_t r;
	asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
}
__device__ __forceinline__ int dot2_first(uint32_t a, uint32_t b)
{
	int r;
	asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int c)
{
	return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), c, false);
}
// (bi-prediction keeps the compiler's own choice: with the single instructions its two streams need 174 registers, 2 waves per SIMD instead of 3)
template <bool LEAN> __device__ __forceinline__ int first_dot(uint32_t a, uint32_t b) { return LEAN ? dot2_first(a, b) : dot2(a, b, 0); }
// The 14-bit intermediate of bi-prediction: (int16_t)(S >> 6).  The reference's second pass assigns its int sum to an int16_t (pred_inter.c:125
// through the 16-bit dst of :519-527), so a sum beyond int16 WRAPS -- reachable: 255 under every positive tap and 0 under every negative one in both
// directions gives 33 150 (tests/cases.py: bi8_wrap).  One v_bfe_i32 (bits 6..21, sign-extended) where a plain shift is one v_ashrrev_i32.
__device__ __forceinline__ int trunc16_shr6(int s) { return __builtin_amdgcn_sbfe(s, 6, 16); }
__device__ __forceinline__ uint32_t med3_u32(uint32_t x, uint32_t lo, uint32_t hi)
{
	uint32_t r;
	asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi));
	return r;
}
// value held by lane + 1 (lane 63 receives 0)
__device__ __forceinline__ uint32_t from_next_lane(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, true /* bound_ctrl: out-of-range source reads 0 */);
}

template <int TAPS> struct Coefs
{
	uint32_t v[TAPS];        // (cy[k], cy[k])
	uint32_t h[TAPS / 2];    // (cx[0], cx[2]) (cx[1], cx[3]) [(cx[4], cx[6]) (cx[5], cx[7])]
	// tap k of phase f, sign-extended
	static __device__ __forceinline__ int tap(int k, int f)
	{
		if (TAPS == 8) return __builtin_amdgcn_sbfe((int)kTaps.luma[k], (uint32_t)(8 * (f & 3)), 8u);
		// the word of phases 0..3 or of phases 4..7, picked by ARITHMETIC on two literals.  Written as `(f & 4) ? chroma[k][1] : chroma[k][0]` the compiler turns the
		// select of two constant-array elements back into a LOAD from the table in constant memory indexed by f & 4 -- eight global_load_dword per reference behind the
		// fraction byte, the very second dependent round trip of the prologue this packing was meant to remove (found in round 6: the pattern-only build issues 8 fewer
		// loads per wave and reference, profiles/r06_pred_floor.txt).
		const uint32_t lo = kTaps.chroma[k & 3][0], hi = kTaps.chroma[k & 3][1];
		const uint32_t packed = lo ^ ((lo ^ hi) & (0u - (((uint32_t)f >> 2) & 1u)));
		return __builtin_amdgcn_sbfe((int)packed, (uint32_t)(8 * (f & 3)), 8u);
	}
	__device__ __forceinline__ void load(int frac)
	{
		const int xf = frac & 15, yf = frac >> 4;
		int cx[TAPS];
#pragma unroll
		for (int k = 0; k < TAPS; ++k)
		{
			const uint32_t c = (uint32_t)tap(k, yf);
			v[k] = __builtin_amdgcn_perm(c, c, 0x01000100);                 // (c, c) as two int16
			cx[k] = tap(k, xf);
		}
#pragma unroll
		for (int k = 0; k < TAPS / 2; ++k)
		{
			const int lo = TAPS == 8 ? (k < 2 ? k : k + 2) : k;              // pairs (0, 2) (1, 3) [(4, 6) (5, 7)]
			h[k] = __builtin_amdgcn_perm((uint32_t)cx[lo + 2], (uint32_t)cx[lo], 0x05040100);
		}
	}
};

// One reference's state: the sliding window of unpacked rows, and the row stream.
// NDW = dwords (4 window columns each) per lane per row.
template <int TAPS, int NDW, int AHEAD> struct RefStream
{
	static constexpr int NX = NDW + (TAPS == 8 ? 2 : 1);   // column groups the horizontal pass reads
	uint32_t e[TAPS][NDW], o[TAPS][NDW];   // window slots: even / odd columns of each dword as int16 pairs
	uint32_t pending[AHEAD][NDW];   // the AHEAD window rows whose loads are in flight (ring, slot = row % AHEAD)
	const uint8_t *plane;        // the reference plane (wave-uniform: the loads are global_load v, v_offset, s[plane])
	uint32_t off;                // this lane's bytes of the next window row to load, as an offset from `plane`
	uint32_t lo, hi;             // the stream's row offsets stay inside [lo, hi]: the block's own rows when yFrac == 0
	                             // (nothing above or below the block may be read), everything otherwise
	uint32_t shift;              // 8 * bytes this lane's load was moved left to stay inside the window
	bool bytewise;               // NARROW only: the window is narrower than a lane load (W < 4, xFrac == 0)
	Coefs<TAPS> c;

	// Issue the load of the next window row of the stream (rows are consumed strictly in order) and step to the one after:
	// one add and one v_med3_u32 -- a stream with yFrac == 0 that has reached the block's last row stays on it.  The
	// registers are left exactly as loaded: touching them here -- even the realigning shift -- would make the wave wait for
	// the load here and throw the look-ahead away (realign() is applied on use).
	template <bool NARROW> __device__ __forceinline__ void next_row(uint32_t (&d)[NDW], uint32_t step, int W)
	{
		const uint8_t *p = plane + (size_t)off;
		if (NDW == 1)
		{
			if (NARROW && bytewise)
			{
				uint32_t v = p[0];
				if (W > 1) v |= (uint32_t)p[1] << 8;
				if (W > 2) v |= (uint32_t)p[2] << 16;
				d[0] = v;
			}
			else
			{
				d[0] = reinterpret_cast<const U1 *>(p)->v;
			}
		}
		else
		{
			const u32x2 v = reinterpret_cast<const U2 *>(p)->v;
			d[0] = v.x;
			d[NDW - 1] = v.y;
		}
		off = med3_u32(off + step, lo, hi);
	}
	// the last lane's load was moved left to stay inside the window: move its bytes back
	__device__ __forceinline__ void realign(uint32_t (&d)[NDW]) const
	{
#ifdef HEVCASM_PRED_PATTERN_ONLY
		(void)d; return;
#endif
		if (NDW == 1)
		{
			d[0] >>= shift;
		}
		else
		{
			const unsigned long long q = (((unsigned long long)d[NDW - 1] << 32) | d[0]) >> shift;
			d[0] = (uint32_t)q;
			d[NDW - 1] = (uint32_t)(q >> 32);
		}
	}
	__device__ __forceinline__ void zero_row(uint32_t (&d)[NDW])
	{
#pragma unroll
		for (int m = 0; m < NDW; ++m) d[m] = 0;
	}
	__device__ __forceinline__ void unpack(int slot, const uint32_t (&d)[NDW])
	{
#pragma unroll
		for (int m = 0; m < NDW; ++m)
		{
#ifdef HEVCASM_PRED_PATTERN_ONLY
			e[slot][m] = d[m]; o[slot][m] = d[m];
#else
			e[slot][m] = __builtin_amdgcn_perm(0, d[m], 0x0c020c00);   // (byte 0, byte 2) zero-extended
			o[slot][m] = __builtin_amdgcn_perm(0, d[m], 0x0c030c01);   // (byte 1, byte 3)
#endif
		}
	}
	// vertical filter for the output row whose oldest window row sits in slot `first`
	// `seed` (the same 16-bit value in both halves) starts every sum: the horizontal taps add up to 64 whatever the phase, so a
	// seed of 32 comes out of the horizontal pass as the uni-prediction rounding constant 32 * 64 = 2048, for free -- seeding the
	// horizontal sums themselves costs a register move per output column (v_dot2c accumulates in place).  |sum| <= 88 * 255 + 32 < 2^15.
	template <bool SEEDED> __device__ __forceinline__ void vertical(int first, uint32_t seed, uint32_t (&ve)[NX], uint32_t (&vo)[NX]) const
	{
#ifdef HEVCASM_PRED_PATTERN_ONLY
#pragma unroll
		for (int m = 0; m < NX; ++m) { ve[m] = m < NDW ? e[first % TAPS][m] ^ e[(first + TAPS - 1) % TAPS][m] : 0u; vo[m] = ve[m]; }      // oldest and newest window row: every loaded row is used
		(void)seed; return;
#endif
#pragma unroll
		for (int m = 0; m < NDW; ++m)
		{
			if (SEEDED)
			{
				ve[m] = pk_mad_fused(e[first % TAPS][m], c.v[0], seed);
				vo[m] = pk_mad_fused(o[first % TAPS][m], c.v[0], seed);
			}
			else
			{
				ve[m] = pk_mad(e[first % TAPS][m], c.v[0], 0);
				vo[m] = pk_mad(o[first % TAPS][m], c.v[0], 0);
			}
#pragma unroll
			for (int k = 1; k < TAPS; ++k)
			{
				ve[m] = pk_mad(e[(first + k) % TAPS][m], c.v[k], ve[m]);
				vo[m] = pk_mad(o[(first + k) % TAPS][m], c.v[k], vo[m]);
			}
		}
	}
	// horizontal filter: exact 2-D sums S for this lane's 4 * NDW output columns.  ve/vo[0..NDW) are
	// this lane's vertical results; the rest is fetched from the lanes to the right.  (The uni-prediction rounding constant is
	// already inside: vertical()'s seed.)
	template <bool LEAN> __device__ __forceinline__ void horizontal(uint32_t (&ve)[NX], uint32_t (&vo)[NX], int (&s)[4 * NDW]) const
	{
#ifdef HEVCASM_PRED_PATTERN_ONLY
#pragma unroll
		for (int m = 0; m < 4 * NDW; ++m) s[m] = (int)ve[m / 4];
		(void)vo; return;
#endif
#pragma unroll
		for (int m = NDW; m < NX; ++m)
		{
			// column group m lives in lane + 1 as its group m - NDW (NDW == 1, 8 taps: group 2 = lane + 2)
			ve[m] = from_next_lane(ve[m - NDW]);
			vo[m] = from_next_lane(vo[m - NDW]);
		}
		uint32_t a[NX - 1], b[NX - 1];
#pragma unroll
		for (int m = 0; m < NX - 1; ++m)
		{
			a[m] = __builtin_amdgcn_alignbit(ve[m + 1], ve[m], 16);   // (col 4m+2, col 4m+4)
			b[m] = __builtin_amdgcn_alignbit(vo[m + 1], vo[m], 16);   // (col 4m+3, col 4m+5)
		}
#pragma unroll
		for (int m = 0; m < NDW; ++m)
		{
			if (TAPS == 8)
			{
				s[4 * m + 0] = dot2(vo[m + 1], c.h[3], dot2(ve[m + 1], c.h[2], dot2(vo[m], c.h[1], first_dot<LEAN>(ve[m], c.h[0]))));
				s[4 * m + 1] = dot2(a[m + 1], c.h[3], dot2(vo[m + 1], c.h[2], dot2(a[m], c.h[1], first_dot<LEAN>(vo[m], c.h[0]))));
				s[4 * m + 2] = dot2(b[m + 1], c.h[3], dot2(a[m + 1], c.h[2], dot2(b[m], c.h[1], first_dot<LEAN>(a[m], c.h[0]))));
				s[4 * m + 3] = dot2(ve[m + 2], c.h[3], dot2(b[m + 1], c.h[2], dot2(ve[m + 1], c.h[1], first_dot<LEAN>(b[m], c.h[0]))));
			}
			else
			{
				s[4 * m + 0] = dot2(vo[m], c.h[1], first_dot<LEAN>(ve[m], c.h[0]));
				s[4 * m + 1] = dot2(a[m], c.h[1], first_dot<LEAN>(vo[m], c.h[0]));
				s[4 * m + 2] = dot2(b[m], c.h[1], first_dot<LEAN>(a[m], c.h[0]));
				s[4 * m + 3] = dot2(ve[m + 1], c.h[1], first_dot<LEAN>(b[m], c.h[0]));
			}
		}
	}
};

// luma bi-prediction on the branch-free loop starts its horizontal sums on the VOP3P dot product too (round 4: 156 -> 164 registers, still three
// waves per SIMD; -2.2 % over the 22 PU sizes, -4.6 ... +0.9 %: profiles/r04_pred_bi_lean_ab.txt); the other bi variants would lose a wave
#ifndef HEVCASM_BI_LEAN
#define HEVCASM_BI_LEAN 1
#endif
template <bool B> struct Flag { static constexpr bool value = B; };

// a second set of planes for the same PU list (null dst = none), as byte distances from the first set
struct Planes2 { int count; ptrdiff_t dst, ref0, ref1; };
static ptrdiff_t plane_distance(const void *b, const void *a) { return (ptrdiff_t)(reinterpret_cast<uintptr_t>(b) - reinterpret_cast<uintptr_t>(a)); }   // (separate allocations: integer arithmetic on the flat addresses)


// FAST: W % 4 == 0 and (rows per wave) % TAPS == 0 -- the row loop then has no branch around any load or store (see the loop).
// (Round 4: holding the uni 16-rows-in-flight variant to 96 registers for a fifth wave per SIMD (5 dwords of scratch) costs 4-9 %, and a third wave for
// the bi strip kernel (168 registers, 10 dwords of scratch) changes nothing: profiles/r04_pred_occupancy_ab.txt.  The compiler's own allocation stays.)
// (the body of pred_stream_kernel as a device function: the same code serves the one-size launches below and, per run of equally sized PUs, the one-launch
// frame entry points hevcasm_hip_pred_*_runs; `wg` = the workgroup's index within its launch or run)
template <int TAPS, int NDW, bool BI, int AHEAD, bool NARROW, bool FAST>
__device__ __forceinline__ void pred_stream_body(
	uint8_t *__restrict__ dst, ptrdiff_t stride_dst,
	const uint8_t *__restrict__ ref0, const uint8_t *__restrict__ ref1, ptrdiff_t stride_ref,
	const uint32_t *__restrict__ dst_off, const uint32_t *__restrict__ ref0_off, const uint32_t *__restrict__ ref1_off,
	const uint8_t *__restrict__ frac0, const uint8_t *__restrict__ frac1,
	int W, int H, int LW /* lanes per PU */, int PPW /* PUs per wave */, uint32_t n_blocks, int log2_segments, const uint32_t wg)
{
	constexpr int NREF = BI ? 2 : 1;
	constexpr int BEFORE = TAPS / 2 - 1;
	constexpr int CPL = 4 * NDW;            // window / output columns per lane
#ifdef HEVCASM_PRED_PATTERN_ONLY
	occupancy_pad<(TAPS == 8 && NDW == 2 && !BI && AHEAD == 16 && FAST) ? 4 : 8>(n_blocks);      // luma uni 16x16's kernel runs 4 waves per SIMD (99 registers)
#endif
	typedef RefStream<TAPS, NDW, AHEAD> Stream;
	static_assert(TAPS % AHEAD == 0 || AHEAD % TAPS == 0, "the look-ahead ring and the unroll must divide one another");
	constexpr int UNROLL = AHEAD > TAPS ? AHEAD : TAPS;     // slot indices (i % TAPS, i % AHEAD) are static after unrolling by this

	const uint32_t wave = __builtin_amdgcn_readfirstlane(wg * (kThreads / kWave) + (threadIdx.x >> 6));
	const int lane = threadIdx.x & 63;
	int sub = lane / LW;
	int j = lane - sub * LW;
	const bool spare = sub >= PPW;          // lanes left over when 64 % LW != 0: park them on lane 0's data
	if (spare) { sub = PPW - 1; j = 0; }
	// Small launches of tall PUs are cut into 2 or 4 row segments, one wave each (neighbouring waves = the segments of
	// one group of PUs, so the T-1 window rows two segments share come from L2): see launch_pred_ndw().
	const int segment = (int)(wave & ((1u << log2_segments) - 1u));
	const int H_pu = H;
	const int y_first = segment * (H >> log2_segments);
	H >>= log2_segments;                       // the launcher only splits heights that divide
	// stream index of the PU's last row for a stream that starts AT row y_first (yFrac == 0).  Segments are at least 2 T rows,
	// so only the PU's last segment ever gets there (the others' T - 1 look-ahead rows are rows of the next segment).
	const int last_row = H_pu - 1 - y_first;
	uint32_t pu = (wave >> log2_segments) * PPW + sub;
	const bool in_batch = pu < n_blocks;
	if (!in_batch) pu = n_blocks - 1;

	const int cols = min(CPL, W - CPL * j);           // output columns this lane stores (<= 0: none)
	const bool stores = in_batch && !spare && cols > 0;
#ifdef PRED_STAMPS
	unsigned long long *stamps = g_pred_stamps ? g_pred_stamps + (size_t)wave * 32 : nullptr;
#endif
	STAMP(0);

	// Every byte a block touches lies within 4 GiB above its plane's base pointer (hevcasm_hip.h), so rows are addressed as
	// uniform base + 32-bit lane offset: one VGPR and one add per row and plane instead of a 64-bit pointer.
	const uint32_t step_ref = (uint32_t)stride_ref, step_dst = (uint32_t)stride_dst;
	Stream rs[NREF];
	// (Round 4: the wave's reference offsets and fraction bytes by scalar loads + v_cndmask, as the SAD kernels fetch their descriptors: bit-exact and
	// 6-9 % slower on every luma size -- these kernels are VALU-bound and the selects cost more than the trip they shorten:
	// profiles/r04_pred_scalar_descriptors_ab.txt.  Per-lane loads stay.)
#pragma unroll
	for (int r = 0; r < NREF; ++r)
	{
		const int frac = (r ? frac1 : frac0)[pu];
		const uint32_t ref_off_pu = (r ? ref1_off : ref0_off)[pu];
		const bool fx = (frac & 15) != 0, fy = (frac >> 4) != 0;
		// the columns / rows the reference reads for this fraction: the filter margins only where it filters
		const int window = fx ? W + TAPS - 1 : W;
		const int before_x = fx ? BEFORE : 0, before_y = fy ? BEFORE : 0;
		// A lane whose natural load [CPL j, CPL j + CPL) would pass the window's end is moved left to end there; lanes
		// wholly beyond it (the window of a zero fraction is T - 1 columns shorter than the lane count allows for) hold
		// nothing any stored pixel depends on with a non-zero coefficient: they load the same last CPL columns.
		const int natural_end = CPL * j + CPL;
		int over = natural_end > window ? natural_end - window : 0;
		rs[r].bytewise = NARROW && window < CPL;
		if (NARROW && window < CPL) over = CPL * j;                       // byte loads start at the block's first column
		rs[r].plane = r ? ref1 : ref0;
		const uint32_t first = ref_off_pu + (uint32_t)(y_first - before_y) * step_ref + (uint32_t)(CPL * j - over - before_x);
		const uint32_t last = first + (uint32_t)last_row * step_ref;
		rs[r].off = first;
		rs[r].lo = fy ? 0u : min(first, last);
		rs[r].hi = fy ? 0xffffffffu : max(first, last);
		rs[r].shift = (NARROW && window < CPL) ? 0u : (uint32_t)min(8 * over, 32 * NDW - 8);
		rs[r].c.load(frac);
	}
	uint32_t out_off = dst_off[pu] + (uint32_t)y_first * step_dst + (uint32_t)(CPL * j);
	const int rows = H + TAPS - 1;                    // window rows
#ifdef PRED_STAMPS
	if (stamps && lane == 0 && (rs[0].off + out_off) == 0xdeadbeefu) stamps[31] = 1;     // forces the descriptors to have arrived
	STAMP(1);
#endif

	// prologue: window rows 0 .. T-2 into slots 0 .. T-2, and the load of row T-1 in flight
#pragma unroll
	for (int r = 0; r < NREF; ++r)
	{
		uint32_t first[TAPS - 1][NDW];
#pragma unroll
		for (int q = 0; q < TAPS - 1; ++q) rs[r].template next_row<NARROW>(first[q], step_ref, W);   // rows = H + T - 1 >= T: the first T - 1 window rows always exist (a yFrac == 0 stream repeats the PU's last row)
#pragma unroll
		for (int q = 0; q < TAPS - 1; ++q)
		{
			rs[r].realign(first[q]);
			rs[r].unpack(q, first[q]);
#ifdef PRED_STAMPS
			if (r == 0 && q == 0) { if (stamps && lane == 0 && rs[0].e[0][0] == 0xdeadbeefu) stamps[31] = 2; STAMP(2); }
#endif
		}
#ifdef PRED_STAMPS
		if (r == 0) { if (stamps && lane == 0 && rs[0].e[TAPS - 2][0] == 0xdeadbeefu) stamps[31] = 3; STAMP(3); }
#endif
		// window rows T-1 .. T-2+AHEAD; row T-1+a has ring slot a (output row a consumes it).  One branch around the whole
		// group when the PU has that many rows (a branch around every single load keeps them from being issued back to back)
		if (FAST || H >= AHEAD)
		{
#pragma unroll
			for (int a = 0; a < AHEAD; ++a) rs[r].template next_row<NARROW>(rs[r].pending[a], step_ref, W);
		}
		else
		{
#pragma unroll
			for (int a = 0; a < AHEAD; ++a)
			{
				if (a < H) rs[r].template next_row<NARROW>(rs[r].pending[a], step_ref, W); else rs[r].zero_row(rs[r].pending[a]);
			}
		}
	}

	// One row of output per trip; the loop is unrolled by T so that window slots are static.  The loads
	// of the next AHEAD window rows are in flight while a row is computed (a ring of AHEAD * NDW registers
	// per reference; nothing may touch them before their turn, see next_row()).  Measured on one box, mean
	// algorithmic GB/s over all PU sizes, AHEAD = 1 / 2 / 4 / 8: luma uni 3075 / 3155 / 3260 / 3365, luma bi
	// 2985 / 3085 / 3105 / 3155, chroma uni 2460 / 2560 / 2600, chroma bi 2120 / 2195 / 2285 -- the kernel
	// was latency-bound, and before the loads were left untouched until use (when the realigning shift
	// made every load wait at issue) it ran at 2890 / 2430 / 2375 / 1905.
	const bool store_full = stores && cols >= CPL;
	const bool store_half = stores && NDW == 2 && cols == 4;         // FAST only: W % 8 == 4, the PU's last storing lane
	// one output row: window row y + T - 1 leaves the ring into slot (i + T - 1) % T, its ring slot is refilled with window row
	// y + T - 1 + AHEAD when LOADS, the two passes run and the lane stores its pixels.  i = y % TAPS, static after unrolling.
	auto output_row = [&](auto loads_tag, auto fast_tag, const int i, const int y)
	{
		constexpr bool LOADS = decltype(loads_tag)::value, BRANCH_FREE = decltype(fast_tag)::value;
		int s[NREF][CPL];
#pragma unroll
		for (int r = 0; r < NREF; ++r)
		{
			if (!BI)
			{
				// consume the ring slot in place, then refill it: the other order needs a copy of the slot per row
				rs[r].realign(rs[r].pending[i % AHEAD]);
				rs[r].unpack((i + TAPS - 1) % TAPS, rs[r].pending[i % AHEAD]);           // window row y + T - 1
				if (LOADS && (BRANCH_FREE || y + TAPS - 1 + AHEAD < rows)) rs[r].template next_row<NARROW>(rs[r].pending[i % AHEAD], step_ref, W);     // window row (y + AHEAD) + T - 1
			}
			else
			{
				uint32_t newest[NDW];
#pragma unroll
				for (int m = 0; m < NDW; ++m) newest[m] = rs[r].pending[i % AHEAD][m];
				if (LOADS && (BRANCH_FREE || y + TAPS - 1 + AHEAD < rows)) rs[r].template next_row<NARROW>(rs[r].pending[i % AHEAD], step_ref, W);
				rs[r].realign(newest);
				rs[r].unpack((i + TAPS - 1) % TAPS, newest);
			}
			uint32_t ve[Stream::NX], vo[Stream::NX];
			rs[r].template vertical<!BI>(i, 0x00200020u, ve, vo);
			rs[r].template horizontal<(!BI || (HEVCASM_BI_LEAN && FAST && TAPS == 8))>(ve, vo, s[r]);
		}
		// shift, saturate to [0, 255] and pack: v_ashr_pk_u8_i32 does two pixels per instruction
		uint32_t px[NDW];
#pragma unroll
		for (int m = 0; m < NDW; ++m)
		{
#ifdef HEVCASM_PRED_PATTERN_ONLY
			px[m] = (uint32_t)s[0][4 * m] ^ (BI ? (uint32_t)s[NREF - 1][4 * m] : 0u);
#else
			int t[4];
#pragma unroll
			for (int b = 0; b < 4; ++b)
				t[b] = BI ? trunc16_shr6(s[0][4 * m + b]) + trunc16_shr6(s[NREF - 1][4 * m + b]) + 64 : s[0][4 * m + b];   // uni: 2048 already inside (vertical()'s seed)
			constexpr int SHIFT = BI ? 7 : 12;
			const uint32_t lo = __builtin_amdgcn_ashr_pk_u8_i32(t[0], t[1], SHIFT);
			const uint32_t hi = __builtin_amdgcn_ashr_pk_u8_i32(t[2], t[3], SHIFT);
			px[m] = __builtin_amdgcn_perm(hi, lo, 0x05040100);   // low halves of (hi, lo)
#endif
		}
		uint8_t *out = dst + (size_t)out_off;
		if (BRANCH_FREE)
		{
			// two predicated single stores, no loop
			if (NDW == 1) { if (store_full) reinterpret_cast<U1 *>(out)->v = px[0]; }
			else
			{
				u32x2 v; v.x = px[0]; v.y = px[NDW - 1];
				if (store_full) __builtin_nontemporal_store(v, reinterpret_cast<u32x2_u *>(out));
				if (store_half) reinterpret_cast<U1 *>(out)->v = px[0];
			}
		}
		else if (stores)
		{
			if (cols >= CPL)
			{
				if (NDW == 1) reinterpret_cast<U1 *>(out)->v = px[0];
				else
				{
					u32x2 v; v.x = px[0]; v.y = px[NDW - 1];
					// written once and never read back: streaming store (uni 8-tap -2 % over the 24 PU sizes, 16x16 -7 %;
					// chroma bi -7 %; nothing slower: profiles/r01_pred_ablation.txt)
					__builtin_nontemporal_store(v, reinterpret_cast<u32x2_u *>(out));
				}
			}
			else
			{
				for (int b = 0; b < cols; ++b) out[b] = (uint8_t)(px[b >> 2] >> (8 * (b & 3)));
			}
		}
		out_off += step_dst;
#ifdef PRED_STAMPS
		if (y < 24) { if (stamps && lane == 0 && px[0] == 0xdeadbeefu && y == 99) stamps[31] = 4; STAMP(4 + y); }
#endif
	};

	// One row of output per trip; the loop is unrolled by T so that window slots are static.  The loads
	// of the next AHEAD window rows are in flight while a row is computed (a ring of AHEAD * NDW registers
	// per reference; nothing may touch them before their turn, see next_row()).  Measured on one box, mean
	// algorithmic GB/s over all PU sizes, AHEAD = 1 / 2 / 4 / 8: luma uni 3075 / 3155 / 3260 / 3365, luma bi
	// 2985 / 3085 / 3105 / 3155, chroma uni 2460 / 2560 / 2600, chroma bi 2120 / 2195 / 2285 -- the kernel
	// was latency-bound, and before the loads were left untouched until use (when the realigning shift
	// made every load wait at issue) it ran at 2890 / 2430 / 2375 / 1905.
	if (FAST)
	{
		// H = n T rows with AHEAD = T: every row of the first n - 1 groups refills its ring slot, no row of the last group does.
		// No branch stands between the loads, so the compiler knows how many are in flight at every use and waits for exactly the
		// one it needs (s_waitcnt vmcnt(N)); with a branch around a load it waits for ALL of them at the top of every group of T
		// rows -- the whole memory latency once per T rows (that is what the general loop below still does).
		static_assert(AHEAD >= TAPS, "the load pattern of the branch-free loop");
		// (AHEAD = 2 T: a 16-row luma PU has its whole 23-row window in flight before its first row is computed)
		const int groups = H / UNROLL;
		int y = 0;
		for (int g = 0; g < groups - 1; ++g)
		{
#pragma unroll
			for (int i = 0; i < UNROLL; ++i) output_row(Flag<true>(), Flag<true>(), i, y + i);
			y += UNROLL;
		}
#pragma unroll
		for (int i = 0; i < UNROLL; ++i) output_row(Flag<false>(), Flag<true>(), i, y + i);
	}
	else
	{
		for (int y0 = 0; y0 < H; y0 += UNROLL)
		{
#pragma unroll
			for (int i = 0; i < UNROLL; ++i)
			{
				const int y = y0 + i;
				if (y < H)   // wave-uniform
					output_row(Flag<true>(), Flag<false>(), i, y);
			}
		}
	}
}


template <int TAPS, int NDW, bool BI, int AHEAD, bool NARROW, bool FAST>
__global__ __launch_bounds__(kThreads) void pred_stream_kernel(
	uint8_t *__restrict__ dst, ptrdiff_t stride_dst,
	const uint8_t *__restrict__ ref0, const uint8_t *__restrict__ ref1, ptrdiff_t stride_ref,
	const uint32_t *__restrict__ dst_off, const uint32_t *__restrict__ ref0_off, const uint32_t *__restrict__ ref1_off,
	const uint8_t *__restrict__ frac0, const uint8_t *__restrict__ frac1,
	int W, int H, int LW /* lanes per PU */, int PPW /* PUs per wave */, uint32_t n_blocks, int log2_segments,
	ptrdiff_t plane2_dst, ptrdiff_t plane2_ref0, ptrdiff_t plane2_ref1)
{
	// gridDim.y = 2: the same PU list on a second set of planes (Cb and Cr share motion vectors and geometry); the second
	// set is given as byte distances from the first (blockIdx.y is wave-uniform: scalar adds)
	if (blockIdx.y)
	{
		dst += plane2_dst;
		ref0 += plane2_ref0;
		ref1 += plane2_ref1;
	}
	pred_stream_body<TAPS, NDW, BI, AHEAD, NARROW, FAST>(dst, stride_dst, ref0, ref1, stride_ref, dst_off, ref0_off, ref1_off, frac0, frac1,
		W, H, LW, PPW, n_blocks, log2_segments, xcd_contiguous(blockIdx.x, gridDim.x));
}

// ---- one launch for a frame's partition: runs of equally sized PUs, any sizes, several planes (round 6) -----------------------------------------
//
// A codec's frame is never one PU size: HEVC has 24 (pred_inter.c:438-444) on three planes, and one launch per (size, plane) is up to 72 launches of
// 11 - 21 us for a 1080p frame whose whole data moves in ~3 us (profiles/r05 bench rows inter_pred_latency).  Here the host lists the frame as RUNS --
// `count` consecutive descriptors of one size on one plane -- and ONE launch serves them all: a workgroup finds its run by binary search over the runs'
// first workgroups (the table travels in the kernel arguments) and then runs the general form of the stream kernel for that run's geometry (4 lane
// widths x tap counts; the size-specialised variants of the one-size launches are a throughput matter, a single frame is latency).
constexpr int kMaxPredRuns = 96;
struct PredRunDev { uint32_t wg_count, first, count; uint8_t variant, W, H, lw, ppw, log2_segments, plane, pad; };
struct PredPlaneDev { uint8_t *dst; const uint8_t *ref0, *ref1; ptrdiff_t stride_dst, stride_ref; };
struct PredRunsArgs
{
	PredPlaneDev planes[4];
	const uint32_t *dst_off, *ref0_off, *ref1_off;
	const uint8_t *frac0, *frac1;
	uint32_t n_runs, total_wgs;
	uint32_t wg_start[kMaxPredRuns];      // first workgroup of run i; unused entries hold 0xffffffff.  Dense, so that the lookup is a handful of wide scalar loads
	PredRunDev runs[kMaxPredRuns];
};

template <bool BI>
__global__ __launch_bounds__(kThreads) void pred_runs_kernel(const PredRunsArgs a)
{
	// the run this workgroup belongs to = the number of runs that start at or before it, minus one: all 96 starts are compared (independent scalar loads and
	// compares; a binary search would be seven DEPENDENT loads from the kernel-argument segment before the first descriptor load can be issued)
	uint32_t at_or_before = 0, start = 0;
#pragma unroll
	for (int i = 0; i < kMaxPredRuns; ++i)
	{
		const bool in = a.wg_start[i] <= blockIdx.x;
		at_or_before += in ? 1u : 0u;
		start = in ? a.wg_start[i] : start;      // starts ascend: the last one taken is the run's
	}
	const PredRunDev r = a.runs[at_or_before - 1];
	const PredPlaneDev p = a.planes[r.plane];
	const uint32_t wg = xcd_contiguous(blockIdx.x - start, r.wg_count);
#define HEVC_RUN(T, NDW) pred_stream_body<T, NDW, BI, T, NDW == 1, false>(p.dst, p.stride_dst, p.ref0, p.ref1, p.stride_ref, a.dst_off + r.first, a.ref0_off + r.first, \
		a.ref1_off + r.first, a.frac0 + r.first, a.frac1 + r.first, r.W, r.H, r.lw, r.ppw, r.count, r.log2_segments, wg)
	switch (r.variant)
	{
	case 0: HEVC_RUN(8, 2); break;
	case 1: HEVC_RUN(8, 1); break;
	case 2: HEVC_RUN(4, 2); break;
	default: HEVC_RUN(4, 1); break;
	}
#undef HEVC_RUN
}

template <int TAPS, int NDW, bool BI>
static int launch_pred_ndw(uint8_t *dst, ptrdiff_t sd, const uint8_t *ref0, const uint8_t *ref1, ptrdiff_t sr,
		const uint32_t *dof, const uint32_t *r0, const uint32_t *r1, const uint8_t *f0, const uint8_t *f1,
		int W, int H, size_t n, hipStream_t st, const Planes2 &p2)
{
	const int lw = (W + TAPS - 1 + 4 * NDW - 1) / (4 * NDW);   // lanes per PU: one per 4*NDW window columns
	const int ppw = kWave / lw;
	// A wave lives for H rows and starts with two dependent round trips (descriptors, then window rows); a launch of
	// fewer than ~16 waves per SIMD is one or two generations deep and ends with most of the chip idle.  Such launches
	// of tall PUs are split into row segments of >= 2 T rows (the T-1 extra window rows per segment are loads only).
	// 64 frames of 1080p per launch, same box, ms: uni 8-tap 64x64 0.0955 -> 0.0847, bi 64x64 0.162 -> 0.1445, bi 32x32
	// 0.188 -> 0.170, bi 16x64 0.209 -> 0.192; nothing slower (profiles/r01_pred_segments_ab.txt).
	size_t waves = ceil_div(n, (size_t)ppw) * (size_t)p2.count;     // (the launch's waves: what the segment heuristic looks at)
	int log2_segments = 0;
	while (log2_segments < 2 && waves < 16384 && H % (2 << log2_segments) == 0 && (H >> (log2_segments + 1)) >= 2 * TAPS)
	{
		++log2_segments;
		waves *= 2;
	}
	const uint32_t grid = ceil_div(waves / (size_t)p2.count, (size_t)(kThreads / kWave));
	// TAPS window rows in flight per reference (the ring then turns once per unrolled group)
	static const bool no_fast = getenv("HEVCASM_PRED_NOFAST") != nullptr;   // A/B switch for the measurements in profiles/
	static const bool no_deep = getenv("HEVCASM_PRED_NODEEP") != nullptr;   // A/B switch: T rows in flight everywhere
	if (NDW == 1 && W < 4)   // a zero-xFrac window is narrower than the 4-byte lane load: those lanes load byte by byte
		hipLaunchKernelGGL((pred_stream_kernel<TAPS, NDW, BI, TAPS, NDW == 1, false>), dim3(grid, p2.count), dim3(kThreads), 0, st,
				dst, sd, ref0, ref1, sr, dof, r0, r1, f0, f1, W, H, lw, ppw, (uint32_t)n, log2_segments, p2.dst, p2.ref0, p2.ref1);
	// 2 T window rows in flight instead of T where the rows per wave are a multiple of 2 T: a 16-row luma PU then has its whole 23-row window
	// requested before its first row is computed.  Same box, 64 frames per launch (profiles/r03_pred_deep_ab.txt): luma uni 16-wide -10 ... -13 %,
	// 8- / 24-wide -9 ... -10 %, 32- / 48-wide -4 ... -7 %, 64-wide -0 ... -2 %; chroma -2 ... -16 % (mean -3 %); luma bi +6 ... +12 % (205 registers, two waves
	// per SIMD instead of three): not there.
	// (Also measured: 4 T = 32 rows in flight for 32- and 64-row luma PUs, 163 registers: +-1 %; luma bi with 4-byte lanes and 2 T rows, 128 registers: +8 %.)
	else if ((TAPS == 4 || !BI) && !no_deep && W % 4 == 0 && (H >> log2_segments) % (2 * TAPS) == 0 && !no_fast)
		hipLaunchKernelGGL((pred_stream_kernel<TAPS, NDW, BI, 2 * TAPS, false, true>), dim3(grid, p2.count), dim3(kThreads), 0, st,
				dst, sd, ref0, ref1, sr, dof, r0, r1, f0, f1, W, H, lw, ppw, (uint32_t)n, log2_segments, p2.dst, p2.ref0, p2.ref1);
	else if (W % 4 == 0 && (H >> log2_segments) % TAPS == 0 && !no_fast)
		hipLaunchKernelGGL((pred_stream_kernel<TAPS, NDW, BI, TAPS, false, true>), dim3(grid, p2.count), dim3(kThreads), 0, st,
				dst, sd, ref0, ref1, sr, dof, r0, r1, f0, f1, W, H, lw, ppw, (uint32_t)n, log2_segments, p2.dst, p2.ref0, p2.ref1);
	else
		hipLaunchKernelGGL((pred_stream_kernel<TAPS, NDW, BI, TAPS, false, false>), dim3(grid, p2.count), dim3(kThreads), 0, st,
				dst, sd, ref0, ref1, sr, dof, r0, r1, f0, f1, W, H, lw, ppw, (uint32_t)n, log2_segments, p2.dst, p2.ref0, p2.ref1);
	return launch_status("pred_stream_kernel");
}

// ---- 4-tap (chroma) prediction, one lane per 8-column strip ------------------------------------------------------------------
//
// A chroma PU row is narrow: an 8-wide PU's window is 11 bytes.  pred_stream_kernel gives it two lanes of 8 bytes (the second moved
// left to end with the window), byte-misaligned, and exchanges vertical results between the lanes by DPP; half of its lanes' horizontal
// work produces no pixel.  Here a lane owns a STRIP of up to 8 output columns and loads the whole 11-byte window of the strip itself --
// ONE dword-aligned 16-byte load per row (a quad of lanes = four strips: one vector-L1 access per strip row instead of two or three),
// realigned in registers with v_alignbit -- runs the vertical pass on 12 columns and the horizontal pass on its own results: no
// cross-lane traffic at all, every lane stores pixels, twice the PUs per wave.  The vertical pass is repeated on the 3 columns two
// strips share, which 4 taps can afford (24 v_pk_mad_u16 per row against 16) and 8 taps cannot (round 2 measured that form of the luma
// kernel 6-31 % slower).  Any W (strips of 8, the last one narrower) and any H.
// Footprint: whole 4-byte-aligned words, never below the lowest nor above the highest word that holds a byte of the windows of the PUs
// of ONE wave (at most 64 consecutive PUs of the batch): a load that would pass that word is moved left and the bytes are picked from
// further up the 16 (the wave's span is at least 16 bytes: the launcher sends H < 2 or |stride| < 16 to pred_stream_kernel).
// v_cndmask_b32 on two register values
__device__ __forceinline__ uint32_t pick(bool c, uint32_t a, uint32_t b)
{
	uint32_t r;
	asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(__builtin_amdgcn_ballot_w64(c)));
	return r;
}

// one reference's state of a strip lane (pred_strip4_kernel): window slots, the rows in flight, the row stream
template <int AHEAD> struct StripStream4
{
	static constexpr int TAPS = 4;
	uint32_t e[TAPS][3], o[TAPS][3];     // window slots: even / odd columns of the strip's 12 window columns as int16 pairs
	uint32_t pending[AHEAD][4];          // rows in flight: the 16 aligned bytes
	uint32_t where[AHEAD];               // ... and where in them the strip's first byte is (0 .. 15)
	const uint8_t *plane;
	uint32_t off, lo, hi;                // byte offset of the strip's window in the next row to load; row clamp of a yFrac == 0 stream
	uint32_t last_load;                  // the highest offset a 16-byte load may start at
	Coefs<4> c;

	// load the next window row into ring slot `slot` (registers untouched until take(): the wave must not wait here)
	__device__ __forceinline__ void issue(int slot, uint32_t step)
	{
		const uint32_t a = min(off & ~3u, last_load);
		where[slot] = off - a;
		const u32x4 v = reinterpret_cast<const U4 *>(plane + (size_t)a)->v;      // dword-aligned when the plane base is (any base works)
		pending[slot][0] = v.x; pending[slot][1] = v.y; pending[slot][2] = v.z; pending[slot][3] = v.w;
		off = med3_u32(off + step, lo, hi);
	}
	// ring slot `slot` -> window slot `into`: pick the strip's 12 bytes out of the 16, unpack to int16 pairs
	__device__ __forceinline__ void take(int slot, int into)
	{
		// the strip's bytes start `at` (0 .. 15) bytes into the 16: 0 .. 3 unless the load was moved left, further up the narrower the strip.
		// (Selects on VALUES held in registers: written as selects between array elements the compiler turns them into one load with a
		// per-lane index -- the ring in scratch memory, 3-5 x slower.)
		const uint32_t at = where[slot], sh = 8u * (at & 3u);
		const uint32_t d0 = pending[slot][0], d1 = pending[slot][1], d2 = pending[slot][2], d3 = pending[slot][3];
		const uint32_t x0 = pick(at >= 12u, d3, pick(at >= 8u, d2, pick(at >= 4u, d1, d0)));
		const uint32_t x1 = pick(at >= 8u, d3, pick(at >= 4u, d2, d1));
		const uint32_t x2 = pick(at >= 4u, d3, d2);
#ifdef HEVCASM_PRED_PATTERN_ONLY
		e[into][0] = d0 ^ d3 ^ at; e[into][1] = d1; e[into][2] = d2;
#pragma unroll
		for (int m = 0; m < 3; ++m) o[into][m] = e[into][m];
		(void)sh; return;
#endif
		const uint32_t w[3] = { __builtin_amdgcn_alignbit(x1, x0, sh), __builtin_amdgcn_alignbit(x2, x1, sh), __builtin_amdgcn_alignbit(d3, x2, sh) };
#pragma unroll
		for (int m = 0; m < 3; ++m)
		{
			e[into][m] = __builtin_amdgcn_perm(0, w[m], 0x0c020c00);
			o[into][m] = __builtin_amdgcn_perm(0, w[m], 0x0c030c01);
		}
	}
};

template <bool BI, bool FAST, int AHEAD>
__global__ __launch_bounds__(kThreads) void pred_strip4_kernel(
	uint8_t *__restrict__ dst, ptrdiff_t stride_dst,
	const uint8_t *__restrict__ ref0, const uint8_t *__restrict__ ref1, ptrdiff_t stride_ref,
	const uint32_t *__restrict__ dst_off, const uint32_t *__restrict__ ref0_off, const uint32_t *__restrict__ ref1_off,
	const uint8_t *__restrict__ frac0, const uint8_t *__restrict__ frac1,
	int W, int H, int S /* strips per PU */, int PPW /* PUs per wave */, uint32_t n_blocks,
	ptrdiff_t plane2_dst, ptrdiff_t plane2_ref0, ptrdiff_t plane2_ref1, int log2_segments)
{
	if (blockIdx.y)
	{
		dst += plane2_dst;
		ref0 += plane2_ref0;
		ref1 += plane2_ref1;
	}
	constexpr int NREF = BI ? 2 : 1;
	constexpr int TAPS = 4, UNROLL = AHEAD > TAPS ? AHEAD : TAPS;
#ifdef HEVCASM_PRED_PATTERN_ONLY
	occupancy_pad<(FAST && AHEAD == 8) ? (BI ? 2 : 4) : 8>(n_blocks);      // chroma 8x8's kernels: uni 106 registers = 4 waves per SIMD, bi 177 = 2
#endif
	const uint32_t wg = xcd_contiguous(blockIdx.x, gridDim.x);
	uint32_t wave = __builtin_amdgcn_readfirstlane(wg * (kThreads / kWave) + (threadIdx.x >> 6));      // (wave-uniform: the address arithmetic that follows stays in SGPRs)
	// row segments (round 4): a small launch of tall PUs is few, long waves -- 64 frames of 8x32 chroma PUs are 1920 waves of 32 rows, not two
	// per SIMD -- so such launches cut every PU into 2 or 4 segments of rows, one wave each (neighbouring waves: the 3 window rows two segments
	// share come from the L2), exactly as pred_stream_kernel does; the segment is the PU as far as the rest of the kernel is concerned
	const uint32_t seg = wave & ((1u << log2_segments) - 1u);
	wave >>= log2_segments;
	H >>= log2_segments;
	const uint32_t seg_row0 = seg * (uint32_t)H;
	const int lane = threadIdx.x & 63;
	int sub = lane / S;
	int strip = lane - sub * S;
	const bool spare = sub >= PPW;
	if (spare) { sub = PPW - 1; strip = 0; }
	uint32_t pu = wave * PPW + sub;
	const bool in_batch = pu < n_blocks;
	if (!in_batch) pu = n_blocks - 1;
	const int cols = min(8, W - 8 * strip);
	const bool stores = in_batch && !spare;
	const uint32_t step_ref = (uint32_t)stride_ref, step_dst = (uint32_t)stride_dst;

	typedef StripStream4<AHEAD> Stream;
	Stream rs[NREF];

#pragma unroll
	for (int r = 0; r < NREF; ++r)
	{
		const int frac = (r ? frac1 : frac0)[pu];
		const bool fx = (frac & 15) != 0, fy = (frac >> 4) != 0;
		const int need = cols + (fx ? TAPS - 1 : 0);
		const uint32_t first = (r ? ref1_off : ref0_off)[pu] + seg_row0 * step_ref - (fy ? step_ref : 0u) + (uint32_t)(8 * strip - (fx ? 1 : 0));
		const uint32_t last = first + (uint32_t)(H - 1 + (fy ? TAPS - 1 : 0)) * step_ref;
		rs[r].plane = r ? ref1 : ref0;
		rs[r].off = first;
		rs[r].lo = fy ? 0u : min(first, first + (uint32_t)(H - 1) * step_ref);
		rs[r].hi = fy ? 0xffffffffu : max(first, first + (uint32_t)(H - 1) * step_ref);
		rs[r].c.load(frac);
		// the span of this reference's windows over the wave (per reference: two planes)
		const uint32_t lo_l = min(first, last) & ~3u, hi_l = (max(first, last) + (uint32_t)need + 3u) & ~3u;
		const uint32_t lo_w = wave_min_u32(cols > 0 ? lo_l : 0xffffffffu), hi_w = wave_max_u32(cols > 0 ? hi_l : 0u);
		rs[r].last_load = hi_w - 16u;          // >= lo_w: the launcher guarantees a span of at least 16 bytes
		(void)lo_w;
	}
	uint32_t out_off = dst_off[pu] + seg_row0 * step_dst + (uint32_t)(8 * strip);

	// prologue: window rows 0 .. 2 into slots 0 .. 2, rows 3 .. 6 in flight
#pragma unroll
	for (int r = 0; r < NREF; ++r)
	{
#pragma unroll
		for (int q = 0; q < TAPS - 1; ++q) rs[r].issue(q, step_ref);
#pragma unroll
		for (int q = 0; q < TAPS - 1; ++q) rs[r].take(q, q);
		if (FAST || H >= AHEAD)
		{
#pragma unroll
			for (int a = 0; a < AHEAD; ++a) rs[r].issue(a, step_ref);
		}
		else
		{
#pragma unroll
			for (int a = 0; a < AHEAD; ++a) { if (a < H) rs[r].issue(a, step_ref); }
		}
	}

	const int rows = H + TAPS - 1;
	auto output_row = [&](auto loads_tag, auto fast_tag, const int i, const int y)
	{
		constexpr bool LOADS = decltype(loads_tag)::value, BRANCH_FREE = decltype(fast_tag)::value;
		int s[NREF][8];
#pragma unroll
		for (int r = 0; r < NREF; ++r)
		{
			Stream &st = rs[r];
			st.take(i % AHEAD, (i + TAPS - 1) % TAPS);                                              // window row y + 3
			if (LOADS && (BRANCH_FREE || y + TAPS - 1 + AHEAD < rows)) st.issue(i % AHEAD, step_ref);     // window row y + 3 + AHEAD
			uint32_t ve[3], vo[3];
#ifdef HEVCASM_PRED_PATTERN_ONLY
#pragma unroll
			for (int m = 0; m < 8; ++m) s[r][m] = (int)(st.e[i % TAPS][m / 4] ^ st.e[(i + TAPS - 1) % TAPS][m / 4] ^ (m == 0 ? st.e[i % TAPS][2] ^ st.e[(i + TAPS - 1) % TAPS][2] : 0u));
			(void)ve; (void)vo;
			continue;
#endif
#pragma unroll
			for (int m = 0; m < 3; ++m)
			{
				if (!BI)
				{
					ve[m] = pk_mad_fused(st.e[i % TAPS][m], st.c.v[0], 0x00200020u);      // the seed: 32 * 64 = the rounding constant after the horizontal pass
					vo[m] = pk_mad_fused(st.o[i % TAPS][m], st.c.v[0], 0x00200020u);
				}
				else
				{
					ve[m] = pk_mad(st.e[i % TAPS][m], st.c.v[0], 0);
					vo[m] = pk_mad(st.o[i % TAPS][m], st.c.v[0], 0);
				}
#pragma unroll
				for (int k = 1; k < TAPS; ++k)
				{
					ve[m] = pk_mad(st.e[(i + k) % TAPS][m], st.c.v[k], ve[m]);
					vo[m] = pk_mad(st.o[(i + k) % TAPS][m], st.c.v[k], vo[m]);
				}
			}
			uint32_t a[2], b[2];
#pragma unroll
			for (int m = 0; m < 2; ++m)
			{
				a[m] = __builtin_amdgcn_alignbit(ve[m + 1], ve[m], 16);
				b[m] = __builtin_amdgcn_alignbit(vo[m + 1], vo[m], 16);
			}
#pragma unroll
			for (int m = 0; m < 2; ++m)
			{
				s[r][4 * m + 0] = dot2(vo[m], st.c.h[1], first_dot<!BI>(ve[m], st.c.h[0]));
				s[r][4 * m + 1] = dot2(a[m], st.c.h[1], first_dot<!BI>(vo[m], st.c.h[0]));
				s[r][4 * m + 2] = dot2(b[m], st.c.h[1], first_dot<!BI>(a[m], st.c.h[0]));
				s[r][4 * m + 3] = dot2(ve[m + 1], st.c.h[1], first_dot<!BI>(b[m], st.c.h[0]));
			}
		}
		uint32_t px[2];
#pragma unroll
		for (int m = 0; m < 2; ++m)
		{
#ifdef HEVCASM_PRED_PATTERN_ONLY
			px[m] = (uint32_t)s[0][4 * m] ^ (BI ? (uint32_t)s[NREF - 1][4 * m] : 0u);
#else
			int t[4];
#pragma unroll
			for (int q = 0; q < 4; ++q)
				t[q] = BI ? trunc16_shr6(s[0][4 * m + q]) + trunc16_shr6(s[NREF - 1][4 * m + q]) + 64 : s[0][4 * m + q];
			constexpr int SHIFT = BI ? 7 : 12;
			const uint32_t lo = __builtin_amdgcn_ashr_pk_u8_i32(t[0], t[1], SHIFT);
			const uint32_t hi = __builtin_amdgcn_ashr_pk_u8_i32(t[2], t[3], SHIFT);
			px[m] = __builtin_amdgcn_perm(hi, lo, 0x05040100);
#endif
		}
		uint8_t *out = dst + (size_t)out_off;
		if (stores)
		{
			if (cols == 8)
			{
				u32x2 v; v.x = px[0]; v.y = px[1];
				__builtin_nontemporal_store(v, reinterpret_cast<u32x2_u *>(out));
			}
			else if (cols == 4) reinterpret_cast<U1 *>(out)->v = px[0];
			else if (cols == 6) { reinterpret_cast<U1 *>(out)->v = px[0]; reinterpret_cast<H1 *>(out + 4)->v = (uint16_t)px[1]; }
			else if (cols == 2) reinterpret_cast<H1 *>(out)->v = (uint16_t)px[0];
			else
				for (int q = 0; q < cols; ++q) out[q] = (uint8_t)(px[q >> 2] >> (8 * (q & 3)));
		}
		out_off += step_dst;
	};

	if (FAST)
	{
		// H = n * UNROLL rows: every row of the first n - 1 groups refills its ring slot, no row of the last group does (no branch between
		// the loads); with AHEAD = 8 and H = 8 the whole window is in flight before the first row is computed
		const int groups = H / UNROLL;
		int y = 0;
		for (int g = 0; g < groups - 1; ++g)
		{
#pragma unroll
			for (int i = 0; i < UNROLL; ++i) output_row(Flag<true>(), Flag<true>(), i, y + i);
			y += UNROLL;
		}
		if (AHEAD == UNROLL)
		{
#pragma unroll
			for (int i = 0; i < UNROLL; ++i) output_row(Flag<false>(), Flag<true>(), i, y + i);
		}
	}
	else
	{
		for (int y0 = 0; y0 < H; y0 += UNROLL)
		{
#pragma unroll
			for (int i = 0; i < UNROLL; ++i)
			{
				const int y = y0 + i;
				if (y < H) output_row(Flag<true>(), Flag<false>(), i, y);
			}
		}
	}
}

template <bool BI>
static int launch_pred_strip4(uint8_t *dst, ptrdiff_t sd, const uint8_t *ref0, const uint8_t *ref1, ptrdiff_t sr,
		const uint32_t *dof, const uint32_t *r0, const uint32_t *r1, const uint8_t *f0, const uint8_t *f1,
		int W, int H, size_t n, hipStream_t st, const Planes2 &p2)
{
	const int strips = (W + 7) / 8, ppw = kWave / strips;
	// Small launches of tall PUs: segments of 8 or 16 rows, one wave each (see the kernel).  Only where the segment keeps the 8-rows-in-flight
	// branch-free loop (a multiple of 8 rows).  Same box, 64 frames per launch: profiles/r04_pred_strip_segments_ab.txt
	static const bool no_segments = getenv("HEVCASM_PRED_NOSEGMENTS") != nullptr;     // A/B switch, read once per process
	size_t waves = ceil_div(n, (size_t)ppw) * (size_t)p2.count;
	int log2_segments = 0;
	while (!no_segments && log2_segments < 2 && waves < 16384 && H % (2 << log2_segments) == 0 && (H >> (log2_segments + 1)) >= 8 && (H >> (log2_segments + 1)) % 8 == 0)
	{
		++log2_segments;
		waves *= 2;
	}
	const uint32_t grid = ceil_div(waves / (size_t)p2.count, (size_t)(kThreads / kWave));
	const int Hs = H >> log2_segments;
	// Rows in flight: 4 (the ring turns once per unrolled group of TAPS rows) or 8 -- with 8, an 8-row PU has its whole 11-row window in
	// flight before the first row is computed.  Same box (profiles/r03_pred_strip_ab.txt): 8x8 uni -18 %, 8x16 -8 % / -11 % (uni / bi), 16x16 -5 %,
	// 6x8 uni -10 %; heights that are not multiples of 8 take the general row loop with it and lose 20-30 %, 2-wide and bi 6-wide lose 8-16 %.
	const bool ahead8 = Hs % 8 == 0 && W >= 6 && !(BI && W < 8);
#define STRIP(FASTV, AH) hipLaunchKernelGGL((pred_strip4_kernel<BI, FASTV, AH>), dim3(grid, p2.count), dim3(kThreads), 0, st, \
		dst, sd, ref0, ref1, sr, dof, r0, r1, f0, f1, W, H, strips, ppw, (uint32_t)n, p2.dst, p2.ref0, p2.ref1, log2_segments)
	if (ahead8) STRIP(true, 8);            // Hs % 8 == 0 here: always the branch-free row loop
	else if (Hs % 4 == 0) STRIP(true, 4);
	else STRIP(false, 4);
#undef STRIP
	return launch_status("pred_strip4_kernel");
}

// ---- measured and not adopted (round 4): 8-tap uni-prediction with the HORIZONTAL pass first ----------------------------------------
// One lane per 8-column strip, the first pass in the byte domain (v_dot4_i32_i8 on pixel - 128), the 14-bit sums kept as int16 row pairs, the vertical
// pass on v_dot2_i32_i16 over the pairs; no cross-lane traffic, no idle lane.  Bit-exact (every GPU prediction test, footprints included), 22 % fewer vector
// instructions at 64x64 (11.2 lane-operations per pixel against 14.4) -- and 4 % SLOWER (geometric mean over the PU sizes it serves; 14 % with two 8-byte
// loads per lane instead of one of 16): the same vector-L1 accesses (an unaligned lane load is charged by the bytes asked for), a third more HBM traffic, more
// time waiting.  The stream kernel is not bound by its instruction count but by its access pattern (profiles/r04_pred_access_mix.txt: its loads and stores
// alone take the kernel's time).  profiles/r04_pred_hfirst_ab.txt; source: commit cf83a19.

// ---- 8-tap bi-prediction, horizontal pass first, one lane per (8-column strip, reference) (round 4: 16- / 32-wide PUs, 64x64) --------------
//
// Luma bi-prediction is the one prediction kernel that the VALU bounds (74-83 % busy, 1.15-1.35 x the time of its access pattern alone,
// profiles/r04_pred_access_mix.txt).  This form spends fewer vector instructions: a lane owns a strip of 8 output columns OF ONE REFERENCE, loads the strip's
// 16 window bytes itself (one load per row), filters horizontally in the byte domain (v_dot4_i32_i8 on pixel - 128), keeps the 14-bit sums as int16 row
// pairs and filters vertically with v_dot2_i32_i16 over the pairs; the two lanes of a strip then swap halves of their 14-bit intermediates (quad_perm DPP or
// ds_swizzle: a PU's lanes are its strips of reference 0, then those of reference 1 -- neighbouring lanes read neighbouring bytes; 32-wide -3 % against the
// interleaved deal) and each rounds, clips and stores 4 pixels.  Arithmetic = the reference's (pred_inter.c:490-530): per reference (sum >> 6) stored as
// int16 (it wraps: trunc16_shr6), then (a + b + 64) >> 7 clipped.  Footprints = the reference's, as in pred_stream_kernel (window rows and bytes clamped).
__device__ __forceinline__ int sdot4(uint32_t a, uint32_t b, int c) { return __builtin_amdgcn_sdot4((int)a, (int)b, c, false); }
__device__ __forceinline__ int sdot4_first(uint32_t a, uint32_t b, int k)
{
	int r;
	asm("v_dot4_i32_i8 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
	return r;
}
__device__ __forceinline__ int swap_pair_lane(int v) { return __builtin_amdgcn_mov_dpp(v, 0xb1 /* quad_perm [1, 0, 3, 2] */, 0xf, 0xf, false); }
// the value of the lane that filters the same strip of the other reference: lane ^ 1 (interleaved deal) or lane ^ S (blocked deal)
template <int BLOCKED> __device__ __forceinline__ int partner_lane(int v)
{
	if constexpr (BLOCKED == 0) return swap_pair_lane(v);
	else if constexpr (BLOCKED == 2) return __builtin_amdgcn_mov_dpp(v, 0x4e /* quad_perm [2, 3, 0, 1] */, 0xf, 0xf, false);
	else return __builtin_amdgcn_ds_swizzle(v, 0x1f | (BLOCKED << 10) /* bit-mask mode: lane ^ BLOCKED */);
}

template <int AHEAD, int BLOCKED /* 0: the two references of a strip in neighbouring lanes; else S: a PU's strips of reference 0, then those of reference 1 */>
__global__ __launch_bounds__(kThreads) void pred_hfirst_bi8_kernel(
	uint8_t *__restrict__ dst, ptrdiff_t stride_dst, const uint8_t *__restrict__ ref0, const uint8_t *__restrict__ ref1, ptrdiff_t stride_ref,
	const uint32_t *__restrict__ dst_off, const uint32_t *__restrict__ ref0_off, const uint32_t *__restrict__ ref1_off,
	const uint8_t *__restrict__ frac0, const uint8_t *__restrict__ frac1,
	int W, int H, int S /* strips per PU = W / 8 */, int PPW /* PUs per wave */, uint32_t n_blocks, ptrdiff_t plane2_dst, ptrdiff_t plane2_ref0, ptrdiff_t plane2_ref1)
{
	static_assert(AHEAD == 8 || AHEAD == 4, "the ring turns a whole number of times per unrolled group of 8 rows");
#ifdef HEVCASM_PRED_PATTERN_ONLY
	occupancy_pad<4>(n_blocks);      // 112 registers: 4 waves per SIMD
#endif
	if (blockIdx.y)
	{
		dst += plane2_dst;
		ref0 += plane2_ref0;
		ref1 += plane2_ref1;
	}
	const uint32_t wg = xcd_contiguous(blockIdx.x, gridDim.x);
	const uint32_t wave = __builtin_amdgcn_readfirstlane(wg * (kThreads / kWave) + (threadIdx.x >> 6));
	const int lane = threadIdx.x & 63;
	// which: the reference this lane filters; it stores columns 4 which .. 4 which + 3 of the strip
	int which, sub, j;
	if constexpr (BLOCKED)
	{
		sub = lane / (2 * BLOCKED);
		const int q = lane - sub * (2 * BLOCKED);
		which = q / BLOCKED;
		j = q - which * BLOCKED;
	}
	else
	{
		which = lane & 1;
		sub = (lane >> 1) / S;
		j = (lane >> 1) - sub * S;
	}
	const bool spare = sub >= PPW;
	if (spare) { sub = PPW - 1; j = 0; }
	uint32_t pu = wave * (uint32_t)PPW + (uint32_t)sub;
	const bool in_batch = pu < n_blocks;
	if (!in_batch) pu = n_blocks - 1;
	const bool stores = in_batch && !spare;
	const uint8_t *__restrict__ ref = which ? ref1 : ref0;

	const int fr = which ? frac1[pu] : frac0[pu];
	const int xf = fr & 15, yf = fr >> 4;
	const int bx = xf ? 3 : 0, ax = xf ? 4 : 0, by = yf ? 3 : 0, ay = yf ? 4 : 0;
	const uint32_t step_ref = (uint32_t)stride_ref, step_dst = (uint32_t)stride_dst;

	// taps.  Horizontal: signed bytes, taps 0..3 and 4..7 of phase xf.  Vertical: int16 pairs for the two row parities.
	uint32_t cA = 0, cB = 0;
	int v[8];
#pragma unroll
	for (int k = 0; k < 8; ++k)
	{
		const uint32_t tb = (kTaps.luma[k] >> (8 * (xf & 3))) & 0xffu;
		if (k < 4) cA |= tb << (8 * k); else cB |= tb << (8 * (k - 4));
		v[k] = __builtin_amdgcn_sbfe((int)kTaps.luma[k], (uint32_t)(8 * (yf & 3)), 8u);
	}
	auto pair16 = [](int lo, int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); };
	const uint32_t E0 = pair16(v[0], v[1]), E1 = pair16(v[2], v[3]), E2 = pair16(v[4], v[5]), E3 = pair16(v[6], v[7]);
	const uint32_t O0 = pair16(0, v[0]), O1 = pair16(v[1], v[2]), O2 = pair16(v[3], v[4]), O3 = pair16(v[5], v[6]), O4 = pair16(v[7], 0);

	// the strip's 16 window bytes as ONE load, moved left where it would pass the window's end (the PU's last strip: by one byte where xFrac filters,
	// by eight where it does not) and shifted back in registers (W >= 16: the window is at least 16 bytes wide)
	const int col_nat = 8 * j - bx;
	const int col0 = min(col_nat, W + ax - 16);
	const uint32_t sh_bytes = (uint32_t)(col_nat - col0);                       // 0, 1 or 8
	const bool sh8 = sh_bytes >= 8u;
	const uint32_t hsh = sh_bytes & 3u;
	const int last_row = H + ay + by - 1;                                       // last stream row the reference reads (stream row 0 = PU row -by)
	const uint32_t first_lo = (which ? ref1_off[pu] : ref0_off[pu]) + (uint32_t)(-by) * step_ref + (uint32_t)col0;
	const uint32_t end_lo = first_lo + (uint32_t)last_row * step_ref;
	const uint32_t lo_b = min(first_lo, end_lo), hi_b = max(first_lo, end_lo);
	uint32_t off = first_lo;
	uint32_t out_off = dst_off[pu] + (uint32_t)(8 * j + 4 * which);

	auto issue = [&](uint32_t (&d)[4])
	{
		const u32x4 a = reinterpret_cast<const U4 *>(ref + (size_t)off)->v;
		d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w;
		off = med3_u32(off + step_ref, lo_b, hi_b);
	};
	// 8 horizontal sums of one window row
	auto hrow = [&](const uint32_t (&d)[4], int (&h)[8])
	{
#ifdef HEVCASM_PRED_PATTERN_ONLY
#pragma unroll
		for (int x = 0; x < 8; ++x) h[x] = (int)(d[x & 3] ^ (x < 4 ? 0u : hsh + (sh8 ? 1u : 0u)));
		return;
#endif
		// shift back: by 8 bytes = take the upper half (what lands above it only meets zero taps), by one byte = v_alignbyte across the four words
		const uint32_t e0 = (sh8 ? d[2] : d[0]) ^ 0x80808080u, e1 = (sh8 ? d[3] : d[1]) ^ 0x80808080u;
		const uint32_t e2 = d[2] ^ 0x80808080u, e3 = d[3] ^ 0x80808080u;
		const uint32_t x0 = __builtin_amdgcn_alignbyte(e1, e0, hsh), x1 = __builtin_amdgcn_alignbyte(e2, e1, hsh);
		const uint32_t x2 = __builtin_amdgcn_alignbyte(e3, e2, hsh), x3 = __builtin_amdgcn_alignbyte(0x80808080u, e3, hsh);
		uint32_t g[12];
		g[0] = x0; g[4] = x1; g[8] = x2;
#pragma unroll
		for (int q = 1; q < 4; ++q)
		{
			g[q] = __builtin_amdgcn_alignbyte(x1, x0, q);
			g[4 + q] = __builtin_amdgcn_alignbyte(x2, x1, q);
			g[8 + q] = __builtin_amdgcn_alignbyte(x3, x2, q);
		}
#pragma unroll
		for (int x = 0; x < 8; ++x) h[x] = sdot4(g[x + 4], cB, sdot4_first(g[x], cA, 8192));     // sum c (p - 128) + 128 * 64
	};

	uint32_t P[4][8];             // P[m][c]: sums of (row 2 (i + m'), row 2 (i + m') + 1), column c
	uint32_t ring[AHEAD][4];      // window rows 8 + ... in flight
	{
		uint32_t first8[8][4];
#pragma unroll
		for (int r = 0; r < 8; ++r) issue(first8[r]);
#pragma unroll
		for (int a = 0; a < AHEAD; ++a) issue(ring[a]);
#pragma unroll
		for (int m = 0; m < 4; ++m)
		{
			int h0[8], h1[8];
			hrow(first8[2 * m], h0);
			hrow(first8[2 * m + 1], h1);
#pragma unroll
			for (int c = 0; c < 8; ++c) P[m][c] = __builtin_amdgcn_perm((uint32_t)h1[c], (uint32_t)h0[c], 0x05040100);
		}
	}

	// t: this reference's second-pass sums of one row; the partner lane holds the other reference's
	auto store_row = [&](const int (&t)[8])
	{
#ifdef HEVCASM_PRED_PATTERN_ONLY
		if (stores) reinterpret_cast<U1 *>(dst + (size_t)out_off)->v = (uint32_t)(t[0] ^ t[1] ^ t[2] ^ t[3] ^ t[4] ^ t[5] ^ t[6] ^ t[7]);
		out_off += step_dst;
		return;
#endif
		int mine[4];
#pragma unroll
		for (int k = 0; k < 4; ++k)
		{
			const int lo = trunc16_shr6(t[k]), hi = trunc16_shr6(t[4 + k]);
			const int keep = which ? hi : lo, send = which ? lo : hi;          // the partner stores the other four columns
			mine[k] = keep + partner_lane<BLOCKED>(send) + 64;
		}
		const uint32_t a0 = __builtin_amdgcn_ashr_pk_u8_i32(mine[0], mine[1], 7), a1 = __builtin_amdgcn_ashr_pk_u8_i32(mine[2], mine[3], 7);
		const uint32_t px = __builtin_amdgcn_perm(a1, a0, 0x05040100);
		if (stores) reinterpret_cast<U1 *>(dst + (size_t)out_off)->v = px;
		out_off += step_dst;
	};
	// output rows 2 i and 2 i + 1; the pair whose rows they start on is slot i % 4, the pair that arrives (rows 2 i + 8, 2 i + 9) replaces it
	auto two_rows = [&](auto loads_tag, const int ii)
	{
		constexpr bool LOADS = decltype(loads_tag)::value;
		const int s0 = ii % 4, s1 = (ii + 1) % 4, s2 = (ii + 2) % 4, s3 = (ii + 3) % 4;
		int odd[8], even[8];
#pragma unroll
		for (int c = 0; c < 8; ++c)
		{
#ifdef HEVCASM_PRED_PATTERN_ONLY
			odd[c] = (int)P[s0][c]; even[c] = (int)(P[s0][c] ^ P[s3][c]);
#else
			odd[c] = dot2_first(P[s0][c], O0);
			even[c] = dot2(P[s3][c], E3, dot2(P[s2][c], E2, dot2(P[s1][c], E1, dot2_first(P[s0][c], E0))));
#endif
		}
		store_row(even);
		int h0[8], h1[8];
		hrow(ring[(2 * ii) % AHEAD], h0);
		hrow(ring[(2 * ii + 1) % AHEAD], h1);
		if (LOADS)
		{
			issue(ring[(2 * ii) % AHEAD]);
			issue(ring[(2 * ii + 1) % AHEAD]);
		}
#pragma unroll
		for (int c = 0; c < 8; ++c)
		{
#ifdef HEVCASM_PRED_PATTERN_ONLY
			P[s0][c] = (uint32_t)(h1[c] ^ h0[c]);
			odd[c] ^= (int)P[s0][c];
#else
			P[s0][c] = __builtin_amdgcn_perm((uint32_t)h1[c], (uint32_t)h0[c], 0x05040100);
			odd[c] = dot2(P[s0][c], O4, dot2(P[s3][c], O3, dot2(P[s2][c], O2, dot2(P[s1][c], O1, odd[c]))));
#endif
		}
		store_row(odd);
	};
	const int groups = H / 8;
	for (int g = 0; g < groups - 1; ++g)
	{
#pragma unroll
		for (int ii = 0; ii < 4; ++ii) two_rows(Flag<true>(), ii);
	}
	// the last group: the rows the ring does not hold yet (none with 8 rows ahead)
	two_rows(Flag<(AHEAD < 8)>(), 0);
	two_rows(Flag<(AHEAD < 6)>(), 1);
	two_rows(Flag<false>(), 2);
	two_rows(Flag<false>(), 3);
}

// Where it pays (same binary, 64 frames of 1080p per launch, profiles/r04_pred_hfirst_bi_ab.txt): 16-wide PUs of 16 rows and more -5 ... -9 %, 32-wide of 24 and
// more -3 ... -7 %, 64x64 -2 % (-7 % at 256 frames per launch).  Short PUs lose (the first 8 rows are filtered before the first output row: 32x8 +19 %,
// 16x8 +5 %, 32x16 +4 %), 64-wide ones of fewer than 64 rows +5 ... +14 % (16 lanes to a PU row and 2 bytes loaded per pixel where the stream kernel's 9 lanes
// load 1.1), 24- / 48-wide +9 % / +-0.
#ifndef HEVCASM_HFIRST_AHEAD
#define HEVCASM_HFIRST_AHEAD 8
#endif
static bool hfirst_bi_applies(int W, int H)
{
	static const bool off = getenv("HEVCASM_PRED_NOHFIRST_BI") != nullptr;      // measurement switches, read once per process
	static const bool all = getenv("HEVCASM_PRED_HFIRST_BI_ALL") != nullptr;    // every size the kernel can take
	if (off || W % 8 || H % 8 || W < 16 || W > 64) return false;
	return all || (W == 16 && H >= 16) || (W == 32 && H >= 24) || (W == 64 && H >= 64);
}

static int launch_pred_hfirst_bi8(uint8_t *dst, ptrdiff_t sd, const uint8_t *ref0, const uint8_t *ref1, ptrdiff_t sr, const uint32_t *dof, const uint32_t *r0, const uint32_t *r1,
		const uint8_t *f0, const uint8_t *f1, int W, int H, size_t n, hipStream_t st, const Planes2 &p2)
{
	static const bool interleaved = getenv("HEVCASM_PRED_HFIRST_INTERLEAVED") != nullptr;      // measurement switch, read once per process
	const int strips = W / 8, ppw = kWave / (2 * strips);
	const uint32_t grid = ceil_div(ceil_div(n, (size_t)ppw), (size_t)(kThreads / kWave));
#define HFIRST_BI(B) hipLaunchKernelGGL((pred_hfirst_bi8_kernel<HEVCASM_HFIRST_AHEAD, B>), dim3(grid, p2.count), dim3(kThreads), 0, st, dst, sd, ref0, ref1, sr, dof, r0, r1, f0, f1, W, H, strips, ppw, (uint32_t)n, p2.dst, p2.ref0, p2.ref1)
	if (interleaved || (strips != 2 && strips != 4 && strips != 8)) HFIRST_BI(0);
	else if (strips == 2) HFIRST_BI(2);
	else if (strips == 4) HFIRST_BI(4);
	else HFIRST_BI(8);
#undef HFIRST_BI
	return launch_status("pred_hfirst_bi8_kernel");
}

template <int TAPS, bool BI>
static int launch_pred(uint8_t *dst, ptrdiff_t sd, const uint8_t *ref0, const uint8_t *ref1, ptrdiff_t sr,
		const uint32_t *dof, const uint32_t *r0, const uint32_t *r1, const uint8_t *f0, const uint8_t *f1,
		int W, int H, size_t n, hipStream_t st, const Planes2 &p2 = Planes2{ 1, 0, 0, 0 })
{
	// 8 bytes per lane per row halves the load instructions (the texture-address unit handles 4 lanes
	// per cycle whatever their width); PUs narrower than 8 keep 4 bytes so that no lane is half idle.
	// The window must hold at least one full lane load: W + T - 1 >= 8 for the wide variant.
	// (4 bytes per lane for bi-prediction, 95 VGPRs, was measured too: 10 % slower at 16x16, equal at 64x64.)
	// 4 taps: one lane per 8-column strip (needs a wave's windows to span 16 bytes: at least two rows a 16-byte stride apart)
	static const bool no_strip = getenv("HEVCASM_PRED_NOSTRIP") != nullptr;      // measurement switch, read once per process
	// Where it pays (same box, 256 frames of 960x540 per launch, profiles/r03_pred_strip_ab.txt): 8-wide -13 ... -21 %, 2- and 6-wide -18 ... -26 %,
	// 16-wide -7 ... -10 %, 4-wide -11 % (uni) / +4 % (bi), 12-wide +16 ... +24 % (two strips = the stream kernel's two lanes, more arithmetic),
	// 24- / 32-wide +-0 ... +20 % (the stream kernel's lanes share the vertical pass there).
	const bool strip_pays = (W <= 8 && !(BI && W == 4)) || W == 16;
	if (TAPS == 4 && strip_pays && H >= 2 && (sr >= 16 || sr <= -16) && !no_strip) return launch_pred_strip4<BI>(dst, sd, ref0, ref1, sr, dof, r0, r1, f0, f1, W, H, n, st, p2);
	if constexpr (TAPS == 8 && BI) { if (hfirst_bi_applies(W, H)) return launch_pred_hfirst_bi8(dst, sd, ref0, ref1, sr, dof, r0, r1, f0, f1, W, H, n, st, p2); }
	if (W >= 8) return launch_pred_ndw<TAPS, 2, BI>(dst, sd, ref0, ref1, sr, dof, r0, r1, f0, f1, W, H, n, st, p2);
	return launch_pred_ndw<TAPS, 1, BI>(dst, sd, ref0, ref1, sr, dof, r0, r1, f0, f1, W, H, n, st, p2);
}

static int check_pred_args(int taps, int W, int H, size_t n)
{
	if (taps != 8 && taps != 4) return set_error(HEVCASM_HIP_EINVAL, "taps must be 8 (luma) or 4 (chroma)");
	const int maxdim = taps == 8 ? 64 : 32;  // the reference's intermediate buffers (pred_inter.c:172, :221)
	if (W < 1 || H < 1 || W > maxdim || H > maxdim) return set_error(HEVCASM_HIP_EINVAL, "PU size outside the reference's domain");
	if (n > 0xffffffffull) return set_error(HEVCASM_HIP_EINVAL, "n_blocks exceeds 2^32-1");
	return 0;
}

// host side of the runs entry points: the run table, validated like a one-size call each
template <bool BI>
static int launch_pred_runs(const hevcasm_hip_pred_plane *planes, int n_planes, const hevcasm_hip_pred_run *runs, int n_runs,
		const uint32_t *dof, const uint32_t *r0, const uint32_t *r1, const uint8_t *f0, const uint8_t *f1, hipStream_t st)
{
	if (n_planes < 1 || n_planes > 4) return set_error(HEVCASM_HIP_EINVAL, "1 to 4 planes");
	if (n_runs < 0) return set_error(HEVCASM_HIP_EINVAL, "negative run count");
	if (n_runs == 0) return 0;
	if (!planes || !runs || !dof || !r0 || !f0 || (BI && (!r1 || !f1))) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	for (int p = 0; p < n_planes; ++p)
		if (!planes[p].dst || !planes[p].ref0 || (BI && !planes[p].ref1)) return set_error(HEVCASM_HIP_EINVAL, "null plane pointer");
	size_t launch_waves = 0;      // the whole launch's waves decide whether tall PUs are cut into row segments (launch_pred_ndw's rule)
	for (int i = 0; i < n_runs; ++i)
	{
		const hevcasm_hip_pred_run &r = runs[i];
		if (int e = check_pred_args(r.taps, r.nPbW, r.nPbH, r.count)) return e;
		if (r.plane >= n_planes) return set_error(HEVCASM_HIP_EINVAL, "run names a plane that was not given");
		const int ndw = r.nPbW >= 8 ? 2 : 1, lw = (r.nPbW + r.taps - 1 + 4 * ndw - 1) / (4 * ndw);
		launch_waves += ceil_div(r.count, (size_t)(kWave / lw));
	}
	// Each run's geometry; then the runs are dealt to the grid LONGEST WAVES FIRST (rows per wave, descending): the launch ends when its last wave does, and a
	// 64-row PU that starts in the last generation of workgroups would be the whole tail.
	std::vector<PredRunDev> table;
	table.reserve((size_t)n_runs);
	for (int i = 0; i < n_runs; ++i)
	{
		const hevcasm_hip_pred_run &r = runs[i];
		if (r.count == 0) continue;
		const int T = r.taps, W = r.nPbW, H = r.nPbH;
		const int ndw = W >= 8 ? 2 : 1, lw = (W + T - 1 + 4 * ndw - 1) / (4 * ndw), ppw = kWave / lw;
		size_t all = launch_waves;
		int log2_segments = 0;
		while (log2_segments < 2 && all < 16384 && H % (2 << log2_segments) == 0 && (H >> (log2_segments + 1)) >= 2 * T)
		{
			++log2_segments;
			all *= 2;
		}
		PredRunDev d = {};
		d.first = r.first; d.count = r.count;
		d.variant = (uint8_t)((T == 8 ? 0 : 2) + (ndw == 2 ? 0 : 1));
		d.W = (uint8_t)W; d.H = (uint8_t)H; d.lw = (uint8_t)lw; d.ppw = (uint8_t)ppw; d.log2_segments = (uint8_t)log2_segments; d.plane = r.plane;
		table.push_back(d);
	}
	std::stable_sort(table.begin(), table.end(), [](const PredRunDev &x, const PredRunDev &y) { return (x.H >> x.log2_segments) > (y.H >> y.log2_segments); });
	for (size_t base = 0; base < table.size(); base += kMaxPredRuns)      // more runs than one table holds: the next launch
	{
		PredRunsArgs a = {};
		for (int p = 0; p < n_planes; ++p) a.planes[p] = PredPlaneDev{ planes[p].dst, planes[p].ref0, BI ? planes[p].ref1 : planes[p].ref0, planes[p].stride_dst, planes[p].stride_ref };
		a.dst_off = dof; a.ref0_off = r0; a.ref1_off = BI ? r1 : r0; a.frac0 = f0; a.frac1 = BI ? f1 : f0;
		uint32_t wgs = 0, used = 0;
		for (int i = 0; i < kMaxPredRuns; ++i) a.wg_start[i] = 0xffffffffu;
		for (size_t i = base; i < table.size() && i < base + kMaxPredRuns; ++i)
		{
			PredRunDev &d = a.runs[used];
			d = table[i];
			a.wg_start[used++] = wgs;
			const size_t waves = ceil_div(d.count, (size_t)d.ppw) << d.log2_segments;
			d.wg_count = ceil_div(waves, (size_t)(kThreads / kWave));
			wgs += d.wg_count;
		}
		a.n_runs = used; a.total_wgs = wgs;
		hipLaunchKernelGGL(pred_runs_kernel<BI>, dim3(wgs), dim3(kThreads), 0, st, a);
		if (int e = launch_status("pred_runs_kernel")) return e;
	}
	return 0;
}

}  // namespace hevc

using namespace hevc;

extern "C" int hevcasm_hip_pred_uni_8to8_runs(
	const hevcasm_hip_pred_plane *planes, int n_planes, const hevcasm_hip_pred_run *runs, int n_runs,
	const uint32_t *dst_off, const uint32_t *ref_off, const uint8_t *frac, void *stream)
{
	if (int e = check_device()) return e;
	return launch_pred_runs<false>(planes, n_planes, runs, n_runs, dst_off, ref_off, ref_off, frac, frac, as_stream(stream));
}

extern "C" int hevcasm_hip_pred_bi_8to8_runs(
	const hevcasm_hip_pred_plane *planes, int n_planes, const hevcasm_hip_pred_run *runs, int n_runs,
	const uint32_t *dst_off, const uint32_t *ref0_off, const uint32_t *ref1_off, const uint8_t *frac0, const uint8_t *frac1, void *stream)
{
	if (int e = check_device()) return e;
	return launch_pred_runs<true>(planes, n_planes, runs, n_runs, dst_off, ref0_off, ref1_off, frac0, frac1, as_stream(stream));
}

#ifdef PRED_STAMPS
extern "C" int hevcasm_hip_debug_pred_stamps(unsigned long long *buf)
{
	return (int)hipMemcpyToSymbol(HIP_SYMBOL(hevc::g_pred_stamps), &buf, sizeof(buf));
}
#endif

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

// Two planes with one PU list (Cb and Cr: same motion vectors, same geometry): one launch, gridDim.y = 2.
extern "C" int hevcasm_hip_pred_uni_8to8_batch_2planes(
	uint8_t *dst_a, uint8_t *dst_b, ptrdiff_t stride_dst, const uint8_t *ref_a, const uint8_t *ref_b, ptrdiff_t stride_ref,
	const uint32_t *dst_off, const uint32_t *ref_off, const uint8_t *frac,
	int taps, int nPbW, int nPbH, size_t n_blocks, void *stream)
{
	if (int e = check_device()) return e;
	if (int e = check_pred_args(taps, nPbW, nPbH, n_blocks)) return e;
	if (n_blocks == 0) return 0;
	if (!dst_a || !dst_b || !ref_a || !ref_b || !dst_off || !ref_off || !frac) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	hipStream_t st = as_stream(stream);
	const Planes2 p2 = { 2, plane_distance(dst_b, dst_a), plane_distance(ref_b, ref_a), plane_distance(ref_b, ref_a) };
	return taps == 8
		? launch_pred<8, false>(dst_a, stride_dst, ref_a, ref_a, stride_ref, dst_off, ref_off, ref_off, frac, frac, nPbW, nPbH, n_blocks, st, p2)
		: launch_pred<4, false>(dst_a, stride_dst, ref_a, ref_a, stride_ref, dst_off, ref_off, ref_off, frac, frac, nPbW, nPbH, n_blocks, st, p2);
}

extern "C" int hevcasm_hip_pred_bi_8to8_batch_2planes(
	uint8_t *dst_a, uint8_t *dst_b, ptrdiff_t stride_dst,
	const uint8_t *ref0_a, const uint8_t *ref0_b, const uint8_t *ref1_a, const uint8_t *ref1_b, ptrdiff_t stride_ref,
	const uint32_t *dst_off, const uint32_t *ref0_off, const uint32_t *ref1_off,
	const uint8_t *frac0, const uint8_t *frac1,
	int taps, int nPbW, int nPbH, size_t n_blocks, void *stream)
{
	if (int e = check_device()) return e;
	if (int e = check_pred_args(taps, nPbW, nPbH, n_blocks)) return e;
	if (n_blocks == 0) return 0;
	if (!dst_a || !dst_b || !ref0_a || !ref0_b || !ref1_a || !ref1_b || !dst_off || !ref0_off || !ref1_off || !frac0 || !frac1)
		return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	hipStream_t st = as_stream(stream);
	const Planes2 p2 = { 2, plane_distance(dst_b, dst_a), plane_distance(ref0_b, ref0_a), plane_distance(ref1_b, ref1_a) };
	return taps == 8
		? launch_pred<8, true>(dst_a, stride_dst, ref0_a, ref1_a, stride_ref, dst_off, ref0_off, ref1_off, frac0, frac1, nPbW, nPbH, n_blocks, st, p2)
		: launch_pred<4, true>(dst_a, stride_dst, ref0_a, ref1_a, stride_ref, dst_off, ref0_off, ref1_off, frac0, frac1, nPbW, nPbH, n_blocks, st, p2);
}
