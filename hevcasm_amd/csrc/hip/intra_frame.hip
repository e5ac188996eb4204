// intra_frame.hip -- whole intra planes reconstructed on the device in wavefront order (second half; include/hevcasm_hip.h,
// hevcasm_hip_intra_reconstruct_frames).
//
// Per transform unit, in decoding order: neighbour record from the reconstructed samples (reference layout pred_intra.c:59-65, corner as
// f265/intra.asm:126-131) -> substitution of unavailable samples (H.265 8.4.4.2.2) -> smoothing decision + filter (8.4.4.2.3) -> prediction
// (intra_predict.hpp: INTRA_DC = pred_intra.c:68-103) -> residual -> forward transform (residual_decode.c:592-892) -> quantise (quantize.c:160-186)
// -> dequantise (quantize.c:53-62) -> inverse transform + add (residual_decode.c:69-413).  The arithmetic of every step the reference has is
// restated exactly (32-bit sums, the reference's rounding, truncating forward stores, int16 clips of the inverse passes, the int16 narrowing of
// hevcasm_clip); pinned for all-DC frames against the composition of the reference's own functions (tests/golden: iframe_dc_*).
//
// Structure (MI355X): ONE persistent launch of single-wavefront workgroups.  A workgroup draws a ticket (atomic counter), looks the ticket up in a table
// that lists every (frame, CTU) in wavefront order -- key cx + 2 cy, the two-CTU lag of HEVC's wavefront parallel processing, frames interleaved inside
// a step -- waits until the up to four neighbouring CTUs it reads from have published their edges, then runs the CTU's TUs in list order with the CTU's
// reconstruction, its source pixels and all scratch in LDS (the intra-CTU dependencies are then just program order of one wave: LDS operations of a wave
// complete in order, so no barrier is needed, only compiler fences).  Tickets are a topological order of the dependency graph, so a workgroup only waits
// for CTUs that are held by workgroups already running: no co-residency requirement, no deadlock, any grid size.
// Hand-off between workgroups (guide: cdna_hip_programming.md section 6, Guideline 16, form R2 "the data IS the flag"): a finished CTU stores its bottom
// row and right column as 8-byte granules { tag = 1, four samples } with agent-scope atomic stores (write-through, no fence); a consumer's wave re-reads its
// (at most 41) granules with agent-scope atomic loads until every tag is set.  A granule is written once, by one store, and is never read before its CTU's
// ticket was drawn; the granule area is zeroed by a memset node in front of every launch.  Every spin is bounded (wall clock) and checks an abort word.
// The neighbour records never exist in HBM: the 4.9x record traffic of the flat intra batch (profiles/r05_new_kernel_traffic.txt) is gone by construction.

#include "common.hpp"
#include "dct_matrix.hpp"
#include "intra_predict.hpp"

#include <algorithm>

namespace hevc {

constexpr int kLanes = 64;            // one wavefront per workgroup
constexpr int kTileStride = 112;      // LDS reconstruction tile: row y at (y + 1), sample x at column (x + 16): x = -1 .. 95, rows -1 .. 63
constexpr int kTileRows = 65;
constexpr int kMaxCtuTus = 256;       // a 64x64 CTU of 4x4 TUs
constexpr uint32_t kStatusBadTu = 1u, kStatusTimeout = 2u;
constexpr unsigned long long kSpinLimitTicks = 200000000ull;      // s_memrealtime runs at 100 MHz: 2 s

struct IntraFramesArgs
{
	uint8_t *recon; const uint8_t *src; uint8_t *pred_out;
	ptrdiff_t stride_recon, stride_src;
	size_t pitch_recon, pitch_src;
	int16_t *levels; int *cbf;
	const hevcasm_hip_intra_tu *tus; const uint32_t *ctu_first;
	hevcasm_hip_intra_qp qp[4];
	int frame_w, frame_h, ctu, cols, rows, cIdx, strong;
	uint32_t n_frames, n_items;
	uint32_t *head;                    // [0] ticket counter, [1] status, [2] abort
	uint32_t *order;                   // ticket -> frame << 20 | cy << 10 | cx
	unsigned long long *granules;      // per (frame, CTU): ctu / 4 bottom-row granules, then ctu / 4 right-column granules
};

__constant__ Matrix32 kDctDevice = make_matrix32();
// DST-VII of 4x4 luma blocks as a matrix: residual_decode.c:592-610 (forward: row k of this times the input), :69-88 (inverse: its transpose)
__constant__ signed char kDst4[4][4] = { { 29, 55, 74, 84 }, { 74, 74, 0, -74 }, { 84, -29, -74, 55 }, { 55, -84, 74, -29 } };

typedef short i16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void wave_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t pack16(int lo, int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }

// ---- coefficient tables in LDS: pairs (W[b][2k], W[b][2k + 1]) for v_dot2_i32_i16, one row per output index b -----------------------------------
// forward: W[b][i] = M_N[b][i] = M32[b * 32 / N][i]: rows of ONE table serve every size.  inverse: W[b][i] = M_N[i][b]: a table per size.
struct Tables
{
	uint32_t wf[32 * 17];       // row pitch 17 dwords: rows on different banks
	uint32_t wi32[32 * 17];
	uint32_t wi16[16 * 9];
	uint32_t wi8[8 * 5];
	uint32_t wi4[4 * 3];
	uint32_t df[4 * 3], di[4 * 3];
};

__device__ __forceinline__ void build_tables(Tables &t, int lane)
{
	for (int e = lane; e < 32 * 16; e += kLanes)
	{
		const int b = e / 16, k = e % 16;
		t.wf[b * 17 + k] = pack16(kDctDevice.m[b][2 * k], kDctDevice.m[b][2 * k + 1]);
		t.wi32[b * 17 + k] = pack16(kDctDevice.m[2 * k][b], kDctDevice.m[2 * k + 1][b]);
	}
	for (int e = lane; e < 16 * 8; e += kLanes)
	{
		const int b = e / 8, k = e % 8;
		t.wi16[b * 9 + k] = pack16(kDctDevice.m[2 * (2 * k)][b], kDctDevice.m[2 * (2 * k + 1)][b]);
	}
	if (lane < 8 * 4)
	{
		const int b = lane / 4, k = lane % 4;
		t.wi8[b * 5 + k] = pack16(kDctDevice.m[4 * (2 * k)][b], kDctDevice.m[4 * (2 * k + 1)][b]);
	}
	if (lane < 4 * 2)
	{
		const int b = lane / 2, k = lane % 2;
		t.wi4[b * 3 + k] = pack16(kDctDevice.m[8 * (2 * k)][b], kDctDevice.m[8 * (2 * k + 1)][b]);
		t.df[b * 3 + k] = pack16(kDst4[b][2 * k], kDst4[b][2 * k + 1]);
		t.di[b * 3 + k] = pack16(kDst4[2 * k][b], kDst4[2 * k + 1][b]);
	}
}

template <int N> __device__ __forceinline__ void load_weights(const Tables &t, bool inverse, bool dst, int b, uint32_t (&w)[N / 2])
{
	const uint32_t *row;
	if (N == 4 && dst) row = (inverse ? t.di : t.df) + b * 3;
	else if (!inverse) row = t.wf + b * (32 / N) * 17;
	else row = N == 32 ? t.wi32 + b * 17 : N == 16 ? t.wi16 + b * 9 : N == 8 ? t.wi8 + b * 5 : t.wi4 + b * 3;
#pragma unroll
	for (int k = 0; k < N / 2; ++k) w[k] = row[k];
}

// One 1-D pass over an N x N int16 tile in LDS (row pitch N + 4): out(a, b) = (sum_i W[b][i] * in[a][i] + round) >> shift for every row a, stored at
// [b][a] (TRANSPOSE_OUT) or [a][b].  Forward passes store the low 16 bits (residual_decode.c:631-634), inverse passes clip to int16 (:113-116).
// Lane = (b, a group): the lane's weight row stays in registers; the rows it reads are the same address in every lane of the group (LDS broadcast).
template <int N, bool INVERSE, bool TRANSPOSE_OUT>
__device__ __forceinline__ void transform_pass(const int16_t *in, int16_t *out, const uint32_t (&w)[N / 2], int shift, int lane)
{
	constexpr int P = N + 4;
	constexpr int G = (kLanes / N) < N ? (kLanes / N) : N;      // row groups: 4x4: 4 (16 lanes busy), 8x8: 8, 16x16: 4, 32x32: 2
	constexpr int PER = N / G;                                  // rows per lane: 1, 1, 4, 16
	// A lane's sum is a chain of N / 2 dependent v_dot2_i32_i16, and this wave is alone on its SIMD most of the time (the kernel is a dependency chain, not a
	// throughput loop): one chain at a time runs at the instruction's latency, not its issue rate.  So FOUR rows' sums advance together (independent chains).
	constexpr int CH = PER < 4 ? PER : 4;
	const int b = lane % N, g = min(lane / N, G - 1);      // (4x4: lanes 16 .. 63 repeat group 3 -- same values to the same places, no branch)
	{
		const int add = 1 << (shift - 1);
#pragma unroll
		for (int i0 = 0; i0 < PER; i0 += CH)
		{
			int acc[CH];
#pragma unroll
			for (int c = 0; c < CH; ++c) acc[c] = add;
#pragma unroll
			for (int k = 0; k < N / 2; ++k)
			{
#pragma unroll
				for (int c = 0; c < CH; ++c)
				{
					const uint32_t *row = reinterpret_cast<const uint32_t *>(in + (g * PER + i0 + c) * P);
					acc[c] = __builtin_amdgcn_sdot2(__builtin_bit_cast(i16x2, row[k]), __builtin_bit_cast(i16x2, w[k]), acc[c], false);
				}
			}
#pragma unroll
			for (int c = 0; c < CH; ++c)
			{
				const int a = g * PER + i0 + c;
				int v = acc[c] >> shift;
				if (INVERSE) v = clip3(-32768, 32767, v);
				if (TRANSPOSE_OUT) out[b * P + a] = (int16_t)v;
				else out[a * P + b] = (int16_t)v;
			}
		}
	}
}

// ---- one CTU's state in LDS ------------------------------------------------------------------------------------------------------------------------

// ENCODE: the kernel is built twice.  The encoder's TU loop then holds no global LOAD at all: on this hardware loads and stores share one counter (vmcnt), and
// the wait in front of a load's first use also waits for every store issued before it -- the previous TU's reconstruction and levels (with one kernel for both
// directions the compiler put such a wait on the encoder's path too, to guard the register the decoder loads into: 4x4 TU 0.79 -> 0.72 us without it).  The
// decoder fetches the CTU's levels into LDS with the rest of the CTU's inputs, for the same reason; and every `if (encode)` is decided at compile time.
template <bool ENCODE> struct CtuLds
{
	Tables tables;
	uint8_t tile[kTileRows * kTileStride];     // reconstruction of this CTU + the row above (x = -1 .. 95) + the column to the left
	uint8_t srcs[ENCODE ? 64 * 64 : 16];       // source pixels of the CTU (encoder)
	int16_t levs[ENCODE ? 8 : 64 * 64];        // the CTU's levels as they lie in memory: 16 per 4x4 cell in z-scan order (decoder)
	uint8_t done[16 * 16];                     // per 4x4 cell: reconstructed
	uint8_t raw[144], flt[144];                // compact neighbour record, and its smoothed form: [2N - 1 - y] left, [2N + x] top, [4N] corner
	uint8_t ref[128];                          // predictor scratch (intra_predict.hpp)
	uint8_t ptile[32 * 32];
	int16_t buf_a[32 * 36], buf_b[32 * 36];    // residual / coefficients, row pitch N + 4
};

struct CtuView
{
	int x0, y0, ctu, frame_w, frame_h;         // CTU origin in the plane
};

// H.265 6.4.1 / 8.4.4.2.2 for a sample at CTU-local (px, py) as the TU now being predicted sees it.  No branches: a lone wave pays an issue slot for every
// exec-mask instruction around a lane conditional, and this runs once per TU.
template <typename LDS> __device__ __forceinline__ bool sample_available(const LDS &L, const CtuView &v, int px, int py)
{
	const int gx = v.x0 + px, gy = v.y0 + py;
	const bool in_picture = (unsigned)gx < (unsigned)v.frame_w & (unsigned)gy < (unsigned)v.frame_h;
	const bool above = py < 0;                                        // the CTU row above: the three CTUs a TU can reach into precede in the wavefront
	const bool left = !above & (px < 0) & (py < v.ctu);              // the CTU to the left; the one below it comes later
	const bool inside = !above & (px >= 0) & (px < v.ctu) & (py < v.ctu);      // to the right / below: later
	const int cell = inside ? (py >> 2) * 16 + (px >> 2) : 0;
	return in_picture & (above | left | (inside & (L.done[cell] != 0)));
}

template <typename LDS> __device__ __forceinline__ uint8_t &tile_at(LDS &L, int px, int py) { return L.tile[(py + 1) * kTileStride + px + 16]; }

// The 4N + 1 neighbouring samples of the N x N TU at CTU-local (tx, ty) into L.raw, unavailable ones substituted.  A lane takes one UNIT in the text's search
// order -- lanes 0 .. N/2 - 1: four samples of the left column, bottom-up; lane N/2: the corner; lanes N/2 + 1 .. N: four samples of the top row --
// because availability is a property of 4x4 cells.  Substitution: an unavailable unit repeats the LAST sample of the nearest available unit before it in
// search order; the units before the first available one repeat its FIRST sample; nothing available: 128.
template <int N, typename LDS> __device__ __forceinline__ void build_neighbours(LDS &L, const CtuView &v, int tx, int ty, int lane)
{
	constexpr int U = N / 2;
	bool have = false;
	uint32_t val = 0;
	{
		// one code path for the three kinds of unit: four bytes `step` apart from `first` -- a left unit walks UP its column from its bottom-most row (memory order
		// of the record = search order: the lowest byte is the bottom-most row), a top unit walks right, the corner repeats its byte; lanes beyond the record
		// repeat the last unit (their result is not used).  The samples are read whether or not they count: the reads travel with the availability lookup
		// (every position a unit can have lies inside the LDS allocation).
		const int u = min(lane, 2 * U);
		const bool is_left = u < U, is_corner = u == U;
		const int px = is_left | is_corner ? tx - 1 : tx + 4 * (u - U - 1);
		const int py = is_left ? ty + 2 * N - 4 - 4 * u : ty - 1;
		const int first = (py + 1 + (is_left ? 3 : 0)) * kTileStride + px + 16;
		const int step = is_left ? -kTileStride : is_corner ? 0 : 1;
		val = (uint32_t)L.tile[first] | ((uint32_t)L.tile[first + step] << 8) | ((uint32_t)L.tile[first + 2 * step] << 16) | ((uint32_t)L.tile[first + 3 * step] << 24);
		have = sample_available(L, v, px, py) & (lane <= 2 * U);
	}
	const unsigned long long mask = __ballot(have);
	const unsigned long long below = mask & ((1ull << lane) - 1ull);
	const int from = mask == 0 ? 0 : below ? 63 - __clzll((long long)below) : __ffsll((long long)mask) - 1;
	const uint32_t theirs = (uint32_t)__shfl((int)val, from, kLanes);
	// (selects, not branches; the corner's byte [4N] is the low byte of dword N of the record, so every unit is ONE dword store)
	const uint32_t sub = mask == 0 ? 128u : from == U ? (theirs & 0xffu) : below ? (theirs >> 24) : (theirs & 0xffu);
	val = have ? val : sub * 0x01010101u;
	uint32_t *raw32 = reinterpret_cast<uint32_t *>(L.raw);
	if (lane <= 2 * U) raw32[lane < U ? lane : lane == U ? N : lane - 1] = val;
}

// H.265 8.4.4.2.3 on the record: [1 2 1] / 4 along the run left column (bottom-up) - corner - top row, end samples kept; 32x32 with strong smoothing and two
// flat edges: the bi-linear form
template <int N, typename LDS> __device__ __forceinline__ void smooth_neighbours(LDS &L, int strong, int lane)
{
	const int corner = L.raw[4 * N];
	bool bi = false;
	if (N == 32 && strong)
	{
		const int a = corner + L.raw[2 * N + 63] - 2 * L.raw[2 * N + 31], b = corner + L.raw[2 * N - 1 - 63] - 2 * L.raw[2 * N - 1 - 31];
		bi = abs(a) < 8 && abs(b) < 8;
	}
	// sample i of the run (4N + 1 of them) lies at byte at(i) of the record: the corner, logically between the two edges, is stored behind them
	auto at = [](int i) { return i < 2 * N ? i : i == 2 * N ? 4 * N : i - 1; };
	const int end_left = L.raw[0], end_top = L.raw[4 * N - 1];
	for (int i0 = 0; i0 <= 4 * N; i0 += kLanes)      // selects only: 1, 2 or 3 trips (8x8, 16x16, 32x32)
	{
		const int i = min(i0 + lane, 4 * N), below = max(i - 1, 0), above = min(i + 1, 4 * N);
		const int cur = L.raw[at(i)];
		const int smooth = ((int)L.raw[at(below)] + 2 * cur + (int)L.raw[at(above)] + 2) >> 2;
		const int k = i < 2 * N ? 2 * N - 1 - i : i - 2 * N - 1;      // distance from the corner along the edge (bi-linear form: 0 .. 62 inside an edge)
		const int linear = ((63 - k) * corner + (k + 1) * (i < 2 * N ? end_left : end_top) + 32) >> 6;
		const bool keep = i == 0 | i == 4 * N | (bi & (i == 2 * N));      // the two end samples always; the corner in the bi-linear form
		L.flt[at(i)] = (uint8_t)(keep ? cur : bi ? linear : smooth);
	}
}

__device__ __forceinline__ int wave_or(int v)
{
	v |= __builtin_amdgcn_mov_dpp(v, 0xb1 /* quad_perm [1, 0, 3, 2] */, 0xf, 0xf, true);
	v |= __builtin_amdgcn_mov_dpp(v, 0x4e /* quad_perm [2, 3, 0, 1] */, 0xf, 0xf, true);
	v |= __builtin_amdgcn_mov_dpp(v, 0x141 /* row_half_mirror */, 0xf, 0xf, true);
	v |= __builtin_amdgcn_mov_dpp(v, 0x140 /* row_mirror */, 0xf, 0xf, true);
	return __builtin_amdgcn_readlane(v, 0) | __builtin_amdgcn_readlane(v, 16) | __builtin_amdgcn_readlane(v, 32) | __builtin_amdgcn_readlane(v, 48);
}

// One TU, start to finish.  (tx, ty): position inside the CTU.
// lev: the TU's N x N levels -- in global memory (encoder: written) or in the CTU's LDS copy (decoder: read)
template <int N, bool ENCODE>
__device__ __forceinline__ void reconstruct_tu(CtuLds<ENCODE> &L, const CtuView &v, const IntraFramesArgs &A, uint8_t *recon, uint8_t *pred_out,
	int tx, int ty, int mode, int16_t *lev, int *cbf_out, int lane)
{
	constexpr bool encode = ENCODE;
	// pixels per lane: a 16x16 TU over all 64 lanes (4 each: the flat kernel's deal of 16 lanes x 16 pixels left three quarters of this wave idle through the
	// prediction, the residual and the reconstruction -- a third of a 16x16 TU's time)
	constexpr int PX = N == 16 ? 4 : IntraGeom<N>::PX, LPB = N * N / PX, LOG2 = IntraGeom<N>::LOG2, P = N + 4;
	const bool luma = A.cIdx == 0;

#ifndef IF_KNOCK_NB
	build_neighbours<N>(L, v, tx, ty, lane);
	wave_sync();
#endif
	// filterFlag (8.4.4.2.3): luma, not DC, not 4x4, and far enough from the horizontal / vertical modes for the size
	bool filter = false;
	if (N > 4 && luma && mode != 1)
	{
		const int dist = min(abs(mode - 26), abs(mode - 10));
		filter = dist > (N == 8 ? 7 : N == 16 ? 1 : 0);
	}
	if (filter)
	{
		smooth_neighbours<N>(L, A.strong, lane);
		wave_sync();
	}
	Chunk<PX> pred;
#ifndef IF_KNOCK_PRED
	intra_predict_block<N, kLanes, PX>(filter ? L.flt : L.raw, L.ref, L.ptile, lane, mode, luma && N < 32, wave_sync, pred);
#else
	for (int k = 0; k < PX / 4; ++k) pred.d[k] = (uint32_t)mode * 0x01010101u;
#endif

	// the lane's PX bytes sit at row r, columns c0 .. of the block (row-major position PX * lane)
	const bool owner = lane < LPB;
	const int r = owner ? PX * lane / N : 0, c0 = owner ? PX * lane % N : 0;
	const bool dst = luma && N == 4;                 // trType 1: DST-VII for 4x4 luma blocks of intra CUs
	const hevcasm_hip_intra_qp q = A.qp[LOG2 - 2];
	uint32_t w[N / 2];

	if (encode)
	{
		if (owner)
		{
			const uint32_t *s32 = reinterpret_cast<const uint32_t *>(L.srcs + (ty + r) * 64 + tx + c0);
			uint32_t *a32 = reinterpret_cast<uint32_t *>(L.buf_a + r * P + c0);
#pragma unroll
			for (int k = 0; k < PX / 4; ++k)
			{
				const uint32_t s = s32[k], p = pred.d[k];
				a32[2 * k] = pack16((int)(s & 0xff) - (int)(p & 0xff), (int)((s >> 8) & 0xff) - (int)((p >> 8) & 0xff));
				a32[2 * k + 1] = pack16((int)((s >> 16) & 0xff) - (int)((p >> 16) & 0xff), (int)(s >> 24) - (int)(p >> 24));
			}
		}
		load_weights<N>(L.tables, false, dst, lane % N, w);
		wave_sync();
		// residual_decode.c:855-892: rows of the residual with shift log2 - 1, then the transposed intermediate with shift log2 + 6
#ifndef IF_KNOCK_FWD
		transform_pass<N, false, true>(L.buf_a, L.buf_b, w, LOG2 - 1, lane);
		wave_sync();
		transform_pass<N, false, true>(L.buf_b, L.buf_a, w, LOG2 + 6, lane);
		wave_sync();
#endif
	}

	// quantise (encoder) or fetch (decoder) the lane's levels; dequantise; the dequantised coefficients go to buf_b TRANSPOSED (the inverse passes sum along rows).
	// These steps are per coefficient, so ALL 64 lanes share them: QPX = N * N / 64 consecutive coefficients of a row each (8x8: 1, 16x16: 4, 32x32: 16) -- dealt
	// like the pixels (PX a lane, N * N / PX lanes) an 8x8 TU kept 8 lanes busy with 8 coefficients each, one after the other, and a 16x16 TU 16 with 16:
	// a quarter of the TU's time (profiles/r06_intra_knockout_split.txt).  (4x4 TUs come here only under -DIF_TU4_GENERAL: 16 lanes, one coefficient each.)
	constexpr int QPX = N == 4 ? 1 : N * N / kLanes;
	const bool q_owner = QPX * lane < N * N;
	const int qr = q_owner ? QPX * lane / N : 0, qc0 = q_owner ? QPX * lane % N : 0;
	int cbf = 0;
#ifndef IF_KNOCK_QUANT
	if (q_owner)
	{
		int level[QPX];
		if (encode)
		{
			const int offset = q.q_offset << (q.q_shift - 16);
#pragma unroll
			for (int b = 0; b < QPX; ++b)
			{
				int x = L.buf_a[qr * P + qc0 + b];
				const bool neg = x < 0;
				x = neg ? -x : x;                                           // abs(-32768) = 32768 in int (quantize.c:176)
				x = (__mul24(x, q.q_scale) + offset) >> q.q_shift;               // both factors below 2^16 (quantize.c:162-168): the full-rate 24-bit multiply is exact
				x = neg ? -x : x;
				x = clip3(-32768, 32767, x);
				cbf |= x;
				level[b] = x;
			}
			// 2 QPX bytes of levels, contiguous over the lanes
			if constexpr (QPX == 1) lev[qr * N + qc0] = (int16_t)level[0];
			else
			{
#pragma unroll
				for (int h = 0; h < QPX; h += 8)
				{
					Chunk<(QPX < 8 ? 2 * QPX : 16)> piece;
#pragma unroll
					for (int k = 0; k < (QPX < 8 ? QPX : 8) / 2; ++k) piece.d[k] = pack16(level[h + 2 * k], level[h + 2 * k + 1]);
					piece.store(reinterpret_cast<uint8_t *>(lev + qr * N + qc0 + h));
				}
			}
		}
		else
		{
#pragma unroll
			for (int b = 0; b < QPX; ++b) level[b] = lev[qr * N + qc0 + b];      // (the CTU's LDS copy)
		}
		const int dq_add = 1 << (q.dq_shift - 1);
#pragma unroll
		for (int b = 0; b < QPX; ++b)
			L.buf_b[(qc0 + b) * P + qr] = (int16_t)clip3(-32768, 32767, (__mul24(level[b], q.dq_scale) + dq_add) >> q.dq_shift);      // int16 x int16: exact in 24-bit factors
	}
	if (encode && cbf_out)
	{
		cbf = wave_or(cbf);
		if (lane == 0) *cbf_out = cbf;
	}
#endif
	load_weights<N>(L.tables, true, dst, lane % N, w);
	wave_sync();
	// residual_decode.c:371-413: columns of the coefficients with shift 7 (clipped, stored transposed), then the same with shift 12
#ifndef IF_KNOCK_INV
	transform_pass<N, true, true>(L.buf_b, L.buf_a, w, 7, lane);
	wave_sync();
	transform_pass<N, true, false>(L.buf_a, L.buf_b, w, 12, lane);
	wave_sync();
#endif

	if (owner)
	{
		const uint32_t *b32 = reinterpret_cast<const uint32_t *>(L.buf_b + r * P + c0);
		Chunk<PX> rec;
#pragma unroll
		for (int k = 0; k < PX / 4; ++k)
		{
			uint32_t outw = 0;
#pragma unroll
			for (int b = 0; b < 4; ++b)
			{
				const uint32_t pair = b32[2 * k + b / 2];
				const int res = (b & 1) ? (int)pair >> 16 : (int)(int16_t)(pair & 0xffff);
				const int sum = (int)(int16_t)(((pred.d[k] >> (8 * b)) & 0xff) + res);      // hevcasm_clip(int16_t x, ...) narrows first (residual_decode.c:350-356, :365)
				outw |= (uint32_t)clip3(0, 255, sum) << (8 * b);
			}
			rec.d[k] = outw;
		}
		uint32_t *t32 = reinterpret_cast<uint32_t *>(&tile_at(L, tx + c0, ty + r));
#pragma unroll
		for (int k = 0; k < PX / 4; ++k) t32[k] = rec.d[k];
		const ptrdiff_t at = (ptrdiff_t)(v.y0 + ty + r) * A.stride_recon + v.x0 + tx + c0;
		rec.store(recon + at);
		if (pred_out) pred.store(pred_out + at);
	}
	if (lane < (N / 4) * (N / 4)) L.done[((ty >> 2) + lane / (N / 4)) * 16 + (tx >> 2) + lane % (N / 4)] = 1;
	wave_sync();
}

// ---- 4x4 TUs entirely in registers (round 6, second form) ---------------------------------------------------------------------------------------------
//
// A 4x4 TU through the general path above is a dozen dependent LDS round trips (record, reference array, tile transposes, four transform passes) and about
// 700 instructions -- for 16 pixels, on a wave that is alone on its SIMD: 1.5 us, and a mixed frame's CTU is mostly 4x4 TUs.  Here lane 4 y + x owns pixel
// (x, y) (lanes 16 .. 63 mirror lanes 0 .. 15: same values, only lanes 0 .. 15 store):
//   * lane j <= 16 reads neighbour sample j in the text's search order (0 .. 7 left column bottom-up, 8 corner, 9 .. 16 top row) and its availability: ONE LDS
//     round trip, shared with the source pixel; a ballot gives the availability mask;
//   * every pixel needs two neighbour samples (a, b) whatever the mode: their record indices are arithmetic on (mode, x, y), substitution (8.4.4.2.2) is
//     arithmetic on the mask (the nearest available sample below in search order, else the lowest available one), and the samples arrive by ds_bpermute from
//     the lanes that hold them: the SECOND and last round trip.  No neighbour record, no reference array, no transposition (a horizontal mode is the vertical
//     one with x and y exchanged in the index arithmetic);
//   * the four 1-D transform passes are DPP: a horizontal pass reads the quad's four lanes (quad_perm broadcasts), a vertical pass the lanes 4, 8, 12 further
//     round the row of 16 (row_ror), against per-lane weights held in registers for the whole kernel.
// Same integer arithmetic as the general path, step for step (truncating forward stores, int16 clips of the inverse passes, the int16 narrowing before the
// final clip); tests/test_intra_frames.py and the golden iframe_* cases run through it.
struct Tu4Lane
{
	int x, y;                  // the lane's pixel
	int ndx, ndy;              // neighbour sample min(lane, 16) relative to the TU's origin
	int wfh[4], wfv[4];        // forward: horizontal pass W[x][k]; vertical pass W[y][(y - k) & 3] (the lane k rows up the rotation)
	int wiv[4], wih[4];        // inverse: vertical pass M[(y - k) & 3][y]; horizontal pass M[k][x]
	ptrdiff_t at;              // the pixel's offset from the TU's first in the reconstructed plane
};

__device__ __forceinline__ Tu4Lane make_tu4_lane(int lane, bool dst, ptrdiff_t stride_recon)
{
	Tu4Lane K;
	K.x = lane & 3; K.y = (lane >> 2) & 3;
	K.at = K.y * stride_recon + K.x;
	const int j = min(lane, 16);
	K.ndx = j <= 8 ? -1 : j - 9;
	K.ndy = j < 8 ? 7 - j : -1;
	// (narrowed to a signed char, which every coefficient of the two 4x4 matrices fits: the compiler then KNOWS the factor is small and multiplies with the
	// full-rate 24-bit instruction; an int of unknown range made it a quarter-rate v_mul_lo_u32 per tap)
	auto M = [&](int b, int i) { return (int)(signed char)(dst ? (int)kDst4[b][i] : (int)kDctDevice.m[8 * b][i]); };
#pragma unroll
	for (int k = 0; k < 4; ++k)
	{
		K.wfh[k] = M(K.x, k);
		K.wfv[k] = M(K.y, (K.y - k) & 3);
		K.wiv[k] = M((K.y - k) & 3, K.y);
		K.wih[k] = M(k, K.x);
	}
	return K;
}

// Diagnostic build only (-DIF_STAMPS, tools/jobs/r06_intra_stamps.sh): s_memtime deltas of the 4x4 path's phases, summed over all waves into a buffer nothing else reads
#ifdef IF_STAMPS
__device__ unsigned long long g_if_stamps[8];
__device__ unsigned long long *g_if_acc_dummy;
#define IF_STAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); if_acc[k] += now_ - last_; last_ = now_; } while (0)
#define IF_STAMP_ARG , unsigned long long (&if_acc)[4]
#define IF_STAMP_PASS , if_acc
#define IF_STAMP_BEGIN() unsigned long long last_ = __builtin_amdgcn_s_memtime()
#else
#define IF_STAMP(k) do { } while (0)
#define IF_STAMP_BEGIN() do { } while (0)
#define IF_STAMP_ARG
#define IF_STAMP_PASS
#endif

template <int CTRL> __device__ __forceinline__ int dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
// sum_k w[k] * (the value of lane k of this lane's quad)
__device__ __forceinline__ int quad_dot(int v, const int (&w)[4])
{
	return __mul24(w[0], dpp<0x00>(v)) + __mul24(w[1], dpp<0x55>(v)) + __mul24(w[2], dpp<0xaa>(v)) + __mul24(w[3], dpp<0xff>(v));
}
// sum_k w[k] * (the value of the lane 4 k below this one, round the row of 16)
__device__ __forceinline__ int column_dot(int v, const int (&w)[4])
{
	return __mul24(w[0], v) + __mul24(w[1], dpp<0x124>(v)) + __mul24(w[2], dpp<0x128>(v)) + __mul24(w[3], dpp<0x12c>(v));      // row_ror:4, 8, 12
}

// angles: intra_angle(mode) in the low half, intra_inv_angle(mode) (0 unless the angle is negative) in the high half -- digested with the TU list
template <bool ENCODE>
__device__ __forceinline__ void reconstruct_tu4(CtuLds<ENCODE> &L, const CtuView &v, const IntraFramesArgs &A, const Tu4Lane &K, uint8_t *recon, uint8_t *pred_out,
	int tx, int ty, int mode, uint32_t angles, int16_t *lev, int *cbf_out, int lane IF_STAMP_ARG)
{
	constexpr bool encode = ENCODE;
	IF_STAMP_BEGIN();
	const bool luma = A.cIdx == 0;
	const hevcasm_hip_intra_qp q = A.qp[0];
	int level = 0;
	if (!encode) level = lev[lane & 15];      // (decoder: from the CTU's LDS copy)

	// neighbour sample min(lane, 16), its availability, the lane's source pixel: one round trip
	// (sample_available() spelled out so that its LDS read is issued WITH the other two; the scheduling barrier keeps the compiler from sinking them behind the wait)
	const int px = tx + K.ndx, py = ty + K.ndy;
	const bool in_picture = (unsigned)(v.x0 + px) < (unsigned)v.frame_w & (unsigned)(v.y0 + py) < (unsigned)v.frame_h;
	const bool above = py < 0, beside = !above & (px < 0) & (py < v.ctu), inside = !above & (px >= 0) & (px < v.ctu) & (py < v.ctu);
	const int nb = tile_at(L, px, py);
	const int src_px = L.srcs[(ty + K.y) * 64 + tx + K.x];
	const int cell_done = L.done[inside ? (py >> 2) * 16 + (px >> 2) : 0];
	__builtin_amdgcn_sched_barrier(0);
	const bool have = in_picture & (above | beside | (inside & (cell_done != 0))) & (lane <= 16);
	const uint32_t m = (uint32_t)__ballot(have);      // 17 bits
	IF_STAMP(0);
	const int lowest = m ? __builtin_ctz(m) : 0;
	// record index -> the lane whose sample stands in for it (8.4.4.2.2 per sample: availability is per 4-sample unit, so "the last sample of the nearest
	// available unit below" is the nearest available sample below, and "the first sample of the first available unit" the lowest available sample)
	auto stand_in = [&](int i) { const uint32_t t = m & ((2u << i) - 1u); return t ? 31 - (int)__clz(t) : lowest; };
	// (& 0xff: a sample IS a byte, and the compiler must know it to multiply at full rate)
	auto fetch = [&](int i) { const int s = __builtin_amdgcn_ds_bpermute(stand_in(i) << 2, nb) & 0xff; return m ? s : 128; };
	auto fetch_uniform = [&](int i) { const int s = __builtin_amdgcn_readlane(nb, __builtin_amdgcn_readfirstlane(stand_in(i))) & 0xff; return m ? s : 128; };

	int pred;
	if (mode >= 2)
	{
		// 8.4.4.2.6 in family space: row fv, column fu; reference entry r >= 0 is the corner (0) and the main edge (r - 1), r < 0 the other edge projected
		const int angle = (int)(int16_t)(angles & 0xffffu), inv = (int)angles >> 16;
		const bool vertical = mode >= 18;
		// (the main edge runs up the record -- top row, 9 .. -- or down it -- left column, 7 ..)
		const int fv = vertical ? K.y : K.x, fu = vertical ? K.x : K.y;
		const int tt = (fv + 1) * angle, idx = tt >> 5, f = tt & 31;
		const int r = fu + idx + 1;
		auto record = [&](int r)      // (selects only)
		{
			const int k = min((__mul24(r, inv) + 128) >> 8, 8), along = r >= 0 ? r : -k;
			return min(max(8 + (vertical ? along : -along), 0), 16);
		};
		const int a = fetch(record(r));
		const int b = fetch(angle == 0 ? 8 - (vertical ? fv + 1 : -(fv + 1)) : record(r + 1));      // (modes 10 / 26: b has weight 0 and carries the other edge's sample for the boundary filter)
		pred = ((32 - f) * a + f * b + 16) >> 5;
		if (luma && angle == 0)
		{
			const int corner = fetch_uniform(8);
			pred = fu == 0 ? clip3(0, 255, a + ((b - corner) >> 1)) : pred;
		}
	}
	else if (mode == 0)
	{
		const int left = fetch(7 - K.y), top = fetch(9 + K.x), tn = fetch_uniform(13), ln = fetch_uniform(3);      // 8.4.4.2.4
		pred = ((3 - K.x) * left + (K.x + 1) * tn + (3 - K.y) * top + (K.y + 1) * ln + 4) >> 3;
	}
	else
	{
		const int top = fetch(9 + K.x), left = fetch(7 - K.y);      // pred_intra.c:68-103
		int st = top + dpp<0xb1>(top);       // top depends on x only: the quad's sum is the row's
		st += dpp<0x4e>(st);
		int sl = left + dpp<0x124>(left);    // left depends on y only: round the row of 16 in steps of 4
		sl += dpp<0x128>(sl);
		const int dc = (st + sl + 4) >> 3;
		pred = dc;
		if (luma)
		{
			const int edge = K.y == 0 ? top : left;
			const int filtered = (K.x == 0 & K.y == 0) ? (left + 2 * dc + top + 2) >> 2 : (edge + 3 * dc + 2) >> 2;
			pred = (K.x == 0 | K.y == 0) ? filtered : dc;
		}
	}

	int cbf = 0;
#ifdef IF_STAMPS
	pred = __builtin_amdgcn_readfirstlane(pred) == 0x7fffffff ? 0 : pred;      // (forces the prediction to be complete at the stamp)
#endif
	IF_STAMP(1);
	if (encode)
	{
		// residual_decode.c:855-892 (4x4: :592-610 for the DST): rows with shift 1, columns with shift 8, both stored as int16 (truncating)
		// The reference stores both intermediates as int16 (truncating).  For 4x4 blocks of 8-bit video nothing is ever cut off: |residual| <= 255 and the largest
		// absolute row sum of either matrix is 256 (DCT) / 242 (DST), so |first pass| <= (256 * 255 + 1) >> 1 = 32640 and |second pass| <= (256 * 32640 + 128) >> 8
		// = 32640: the casts are identities and are left out (written out, the compiler folded shift + cast into the weights -- w << 15 -- which no longer
		// fit the full-rate 24-bit multiply).
		const int t1 = (quad_dot(src_px - pred, K.wfh) + 1) >> 1;
		const int c = (column_dot(t1, K.wfv) + 128) >> 8;
		// quantize.c:160-186
		const bool neg = c < 0;
		int x = neg ? -c : c;
		x = (__mul24(x, q.q_scale) + (q.q_offset << (q.q_shift - 16))) >> q.q_shift;
		x = neg ? -x : x;
		level = clip3(-32768, 32767, x);
		cbf = level;
	}
	// quantize.c:53-62, then residual_decode.c:371-413: columns with shift 7, rows with shift 12, both clipped to int16
	const int d = clip3(-32768, 32767, (__mul24(level, q.dq_scale) + (1 << (q.dq_shift - 1))) >> q.dq_shift);
	const int x1 = clip3(-32768, 32767, (column_dot(d, K.wiv) + 64) >> 7);
	const int res = clip3(-32768, 32767, (quad_dot(x1, K.wih) + 2048) >> 12);
	const int rec = clip3(0, 255, (int)(int16_t)(pred + res));      // hevcasm_clip narrows to int16 first (residual_decode.c:350-356, :365)
	if (encode && cbf_out)
	{
		cbf |= dpp<0xb1>(cbf);
		cbf |= dpp<0x4e>(cbf);
		cbf |= dpp<0x141>(cbf);
		cbf |= dpp<0x140>(cbf);
		cbf = __builtin_amdgcn_readlane(cbf, 0);      // (the OR of the sixteen levels, as the general path reports it)
	}
#ifdef IF_STAMPS
	if (__builtin_amdgcn_readfirstlane(rec) == 0x7fffffff) return;
#endif
	IF_STAMP(2);
	if (lane < 16)
	{
		tile_at(L, tx + K.x, ty + K.y) = (uint8_t)rec;
		const ptrdiff_t at = (ptrdiff_t)(v.y0 + ty) * A.stride_recon + (v.x0 + tx) + K.at;      // (a scalar product + the lane's constant)
		recon[at] = (uint8_t)rec;
		if (pred_out) pred_out[at] = (uint8_t)pred;
		if (encode) lev[lane] = (int16_t)level;
		if (lane == 0)
		{
			L.done[(ty >> 2) * 16 + (tx >> 2)] = 1;
			if (encode && cbf_out) *cbf_out = cbf;
		}
	}
	wave_sync();
	IF_STAMP(3);
}

// ---- 8x8 TUs: neighbours and prediction in registers, one pixel per lane (round 6) -------------------------------------------------------------------
//
// The same idea one size up: lane 8 y + x owns pixel (x, y); lane j <= 32 reads neighbour sample j (0 .. 15 left column bottom-up, 16 corner, 17 .. 32 top row).
// 8x8 TUs can be smoothed (8.4.4.2.3), so the substituted record is materialised first -- sample j in lane j, by ds_bpermute from its stand-in -- smoothed
// with the two neighbouring lanes (DPP wave_shr / wave_shl), and the pixels gather their two samples from that: three LDS round trips where the general path
// has about ten (record, reference array, transposing tile).  The transforms are register passes too (dct8_forward / dct8_inverse below);
// residual, quantiser and reconstruction are one element per lane.
struct Tu8Lane
{
	int x, y, j;               // the lane's pixel; its neighbour sample min(lane, 32)
	int ndx, ndy;              // that sample relative to the TU's origin
	ptrdiff_t at;              // the pixel's offset from the TU's first in the reconstructed plane
	// the 8-point passes in registers (dct8_forward / dct8_inverse): weights by the lane's place in its row of 8, the transposing lane, the lane's coefficient
	int wf[4], wi[4];
	int transposed;            // lane 8 x + y
	int coefficient;           // after the forward passes the lane (row p, column q) holds coefficient (vertical frequency f(q), horizontal f(p)): 8 f(q) + f(p)
};

// The 8-point DCT along a row of 8 lanes by its even / odd halves (residual_decode.c:305-348 has the same structure): with m = the mirrored lane's value, lanes
// 0 .. 3 form e[x] = v[x] + v[7 - x] and lanes 4 .. 7 o[7 - x] = v[7 - x] - v[x]; even outputs are a 4-point transform of e, odd ones of o, and each quad holds
// exactly the four values its outputs need: four quad_perm broadcasts.  Output k of quad 0 is frequency 2 k, of quad 1 frequency 2 k + 1: f(place) below.
__device__ __forceinline__ int dct8_forward(int v, const int (&w)[4], bool upper, int add, int shift)
{
	const int m = dpp<0x141>(v);      // row_half_mirror
	return (quad_dot(upper ? m - v : v + m, w) + add) >> shift;
}
// the inverse: quad 0 turns the even frequencies it holds into ee[x], quad 1 the odd ones into oo[7 - x]; sample x = ee[x] + oo[x], sample 7 - x = ee[x] - oo[x]
__device__ __forceinline__ int dct8_inverse(int v, const int (&w)[4], bool upper, int add, int shift)
{
	const int s = quad_dot(v, w), m = dpp<0x141>(s);
	return clip3(-32768, 32767, ((upper ? m - s : s + m) + add) >> shift);
}

__device__ __forceinline__ Tu8Lane make_tu8_lane(int lane, ptrdiff_t stride_recon)
{
	Tu8Lane K;
	K.x = lane & 7; K.y = lane >> 3; K.j = min(lane, 32);
	K.ndx = K.j <= 16 ? -1 : K.j - 17;
	K.ndy = K.j < 16 ? 15 - K.j : -1;
	K.at = K.y * stride_recon + K.x;
	auto M = [&](int b, int i) { return (int)(signed char)kDctDevice.m[4 * b][i]; };      // the 8-point matrix: every fourth row of the 32-point one
	auto f = [](int place) { return place < 4 ? 2 * place : 2 * (place - 4) + 1; };
	const int place = K.x, k = place & 3;
#pragma unroll
	for (int q = 0; q < 4; ++q)
	{
		K.wf[q] = place < 4 ? M(2 * k, q) : M(2 * k + 1, 3 - q);          // quad 1's lane q holds o[3 - q]
		K.wi[q] = place < 4 ? M(2 * q, k) : M(2 * q + 1, 3 - k);          // quad 1's lane k computes oo[3 - k]
	}
	K.transposed = 8 * K.x + K.y;
	K.coefficient = 8 * f(K.x) + f(K.y);
	return K;
}

template <bool ENCODE>
__device__ __forceinline__ void reconstruct_tu8(CtuLds<ENCODE> &L, const CtuView &v, const IntraFramesArgs &A, const Tu8Lane &K, uint8_t *recon, uint8_t *pred_out,
	int tx, int ty, int mode, uint32_t angles, int16_t *lev, int *cbf_out, int lane)
{
	constexpr bool encode = ENCODE;
	constexpr int N = 8, P = N + 4;
	const bool luma = A.cIdx == 0;
	const hevcasm_hip_intra_qp q = A.qp[1];

	// round trip 1: neighbour sample, its availability, the source pixel
	const int px = tx + K.ndx, py = ty + K.ndy;
	const bool in_picture = (unsigned)(v.x0 + px) < (unsigned)v.frame_w & (unsigned)(v.y0 + py) < (unsigned)v.frame_h;
	const bool above = py < 0, beside = !above & (px < 0) & (py < v.ctu), inside = !above & (px >= 0) & (px < v.ctu) & (py < v.ctu);
	const int nb = tile_at(L, px, py);
	const int src_px = encode ? L.srcs[(ty + K.y) * 64 + tx + K.x] : 0;
	const int cell_done = L.done[inside ? (py >> 2) * 16 + (px >> 2) : 0];
	__builtin_amdgcn_sched_barrier(0);
	const bool have = in_picture & (above | beside | (inside & (cell_done != 0))) & (lane <= 32);
	const unsigned long long m = __ballot(have);      // 33 bits
	const uint32_t m_lo = (uint32_t)m;
	const bool last_there = (m >> 32) != 0;

	// round trip 2: the substituted record (8.4.4.2.2, per sample as in reconstruct_tu4), sample j in lane j; then 8.4.4.2.3 on it
	const uint32_t t = m_lo & ((2u << min(K.j, 31)) - 1u);
	int stand_in = t ? 31 - (int)__clz(t) : (m_lo ? __builtin_ctz(m_lo) : 32);
	stand_in = (K.j == 32 & last_there) ? 32 : stand_in;
	int sample = __builtin_amdgcn_ds_bpermute(stand_in << 2, nb) & 0xff;
	sample = m ? sample : 128;
	const int dist = min(abs(mode - 26), abs(mode - 10));
	const bool filter = luma & (mode != 1) & (dist > 7);      // filterFlag for 8x8 luma: planar and the three diagonals
	const int smooth = (dpp<0x138>(sample) /* wave_shr:1 = lane j - 1 */ + 2 * sample + dpp<0x130>(sample) /* wave_shl:1 = lane j + 1 */ + 2) >> 2;
	const int use = (filter & (K.j != 0) & (K.j != 32)) ? smooth : sample;

	// round trip 3: each pixel's two samples
	auto fetch = [&](int i) { return __builtin_amdgcn_ds_bpermute(i << 2, use) & 0xff; };
	int pred;
	if (mode >= 2)
	{
		const int angle = (int)(int16_t)(angles & 0xffffu), inv = (int)angles >> 16;
		const bool vertical = mode >= 18;
		const int fv = vertical ? K.y : K.x, fu = vertical ? K.x : K.y;
		const int tt = (fv + 1) * angle, idx = tt >> 5, f = tt & 31;
		const int r = fu + idx + 1;
		auto record = [&](int r)
		{
			const int k = min((__mul24(r, inv) + 128) >> 8, 16), along = r >= 0 ? r : -k;
			return min(max(16 + (vertical ? along : -along), 0), 32);
		};
		const int a = fetch(record(r));
		const int b = fetch(angle == 0 ? 16 - (vertical ? fv + 1 : -(fv + 1)) : record(r + 1));
		pred = ((32 - f) * a + f * b + 16) >> 5;
		if (luma && angle == 0)
		{
			const int corner = __builtin_amdgcn_readlane(use, 16) & 0xff;
			pred = fu == 0 ? clip3(0, 255, a + ((b - corner) >> 1)) : pred;
		}
	}
	else if (mode == 0)
	{
		const int left = fetch(15 - K.y), top = fetch(17 + K.x), tn = __builtin_amdgcn_readlane(use, 25) & 0xff, ln = __builtin_amdgcn_readlane(use, 7) & 0xff;
		pred = ((7 - K.x) * left + (K.x + 1) * tn + (7 - K.y) * top + (K.y + 1) * ln + 8) >> 4;
	}
	else
	{
		const int top = fetch(17 + K.x), left = fetch(15 - K.y);
		int st = top + dpp<0xb1>(top);       // top depends on x only: quad, then the other quad of the row of 8
		st += dpp<0x4e>(st);
		st += dpp<0x141>(st);                // row_half_mirror
		int sl = left + dpp<0x128>(left);    // left depends on y only: the row's other half (y ^ 1), then the four rows of 16
		sl = __builtin_amdgcn_readlane(sl, 0) + __builtin_amdgcn_readlane(sl, 16) + __builtin_amdgcn_readlane(sl, 32) + __builtin_amdgcn_readlane(sl, 48);
		const int dc = (st + sl + 8) >> 4;
		pred = dc;
		if (luma)
		{
			const int edge = K.y == 0 ? top : left;
			const int filtered = (K.x == 0 & K.y == 0) ? (left + 2 * dc + top + 2) >> 2 : (edge + 3 * dc + 2) >> 2;
			pred = (K.x == 0 | K.y == 0) ? filtered : dc;
		}
	}

	// Residual -> coefficients -> levels -> residual, all in registers.  Forward (residual_decode.c:855-892): along the rows with shift 2, transpose (one
	// ds_bpermute), along what were the columns with shift 9; the reference's int16 stores of both intermediates are identities here as for 4x4 blocks (largest
	// absolute row sum of the 8-point matrix 512: |first| <= (512 * 255 + 2) >> 2 = 32640, |second| <= (512 * 32640 + 256) >> 9 = 32640).  The lane (row p,
	// column q) then holds the coefficient of vertical frequency f(q) and horizontal f(p).  Inverse (:371-413): along q (the columns of the block) with shift 7,
	// transpose, along p with shift 12, both clipped: the lanes are pixels again.
	const bool upper = (lane & 4) != 0;
	int level, cbf = 0;
	if (encode)
	{
		const int t1 = dct8_forward(src_px - pred, K.wf, upper, 2, 2);
		const int c = dct8_forward(__builtin_amdgcn_ds_bpermute(K.transposed << 2, t1), K.wf, upper, 256, 9);
		const bool neg = c < 0;
		int x = neg ? -c : c;
		x = (__mul24(x, q.q_scale) + (q.q_offset << (q.q_shift - 16))) >> q.q_shift;
		x = neg ? -x : x;
		level = clip3(-32768, 32767, x);
		lev[K.coefficient] = (int16_t)level;
		cbf = level;
	}
	else level = lev[K.coefficient];
	const int d = clip3(-32768, 32767, (__mul24(level, q.dq_scale) + (1 << (q.dq_shift - 1))) >> q.dq_shift);
	const int x1 = dct8_inverse(d, K.wi, upper, 64, 7);
	const int res = dct8_inverse(__builtin_amdgcn_ds_bpermute(K.transposed << 2, x1), K.wi, upper, 2048, 12);
	if (encode && cbf_out)
	{
		cbf = wave_or(cbf);
		if (lane == 0) *cbf_out = cbf;
	}

	const int rec = clip3(0, 255, (int)(int16_t)(pred + res));
	tile_at(L, tx + K.x, ty + K.y) = (uint8_t)rec;
	const ptrdiff_t at = (ptrdiff_t)(v.y0 + ty) * A.stride_recon + (v.x0 + tx) + K.at;
	recon[at] = (uint8_t)rec;
	if (pred_out) pred_out[at] = (uint8_t)pred;
	if (lane < 4) L.done[((ty >> 2) + lane / 2) * 16 + (tx >> 2) + lane % 2] = 1;
	wave_sync();
}

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) uint32_t gu32;

__device__ __forceinline__ unsigned long long granule_load(const unsigned long long *p)
{
	return __hip_atomic_load((const gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void granule_store(unsigned long long *p, uint32_t value)
{
	__hip_atomic_store((gu64 *)p, (1ull << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ticket -> (frame, CTU): the wavefront order.  Step s = cx + 2 cy holds the CTUs (s - 2j, j), j = jmin(s) .. ; steps in order, inside a step frame after frame.
__global__ __launch_bounds__(256) void intra_frames_order_kernel(uint32_t *order, int cols, int rows, uint32_t n_frames)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t per_frame = (uint32_t)cols * rows;
	if (t >= per_frame * n_frames) return;
	const uint32_t f = t / per_frame, c = t % per_frame;
	const int cx = (int)(c % cols), cy = (int)(c / cols), s = cx + 2 * cy;
	uint32_t before = 0;                                           // CTUs of one frame in earlier steps: row j holds min(max(s - 2j, 0), cols) of them
	for (int j = 0; j < rows; ++j) before += (uint32_t)min(max(s - 2 * j, 0), cols);
	const int jmin = max(0, (s - cols + 2) / 2), jmax = min(rows - 1, s / 2);
	const uint32_t in_step = (uint32_t)(jmax - jmin + 1);
	order[before * n_frames + f * in_step + (uint32_t)(cy - jmin)] = (f << 20) | ((uint32_t)cy << 10) | (uint32_t)cx;
}

template <bool ENCODE>
__device__ __forceinline__ void intra_frames_body(const IntraFramesArgs &A)
{
	__shared__ __attribute__((aligned(16))) CtuLds<ENCODE> L;
	const int lane = (int)threadIdx.x;
	const int ctu = A.ctu, per_edge = ctu / 4;
	const uint32_t n_ctus = (uint32_t)A.cols * A.rows;
	constexpr bool encode = ENCODE;
	build_tables(L.tables, lane);
	wave_sync();
	const Tu4Lane K4 = make_tu4_lane(lane, A.cIdx == 0, A.stride_recon);      // trType 1 (DST-VII) for 4x4 luma blocks of intra CUs
	const Tu8Lane K8 = make_tu8_lane(lane, A.stride_recon);

	for (;;)
	{
		uint32_t ticket = 0;
		if (lane == 0) ticket = __hip_atomic_fetch_add((gu32 *)A.head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);
		if (ticket >= A.n_items) return;
		const uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane((int)A.order[ticket]);
		const int cx = (int)(item & 1023u), cy = (int)((item >> 10) & 1023u);
		const uint32_t f = item >> 20, c = (uint32_t)cy * A.cols + cx;
		CtuView v = { cx * ctu, cy * ctu, ctu, A.frame_w, A.frame_h };
		uint8_t *recon = A.recon + f * A.pitch_recon;
		uint8_t *pred_out = A.pred_out ? A.pred_out + f * A.pitch_recon : nullptr;
		const uint32_t first = A.ctu_first[f * n_ctus + c], last = A.ctu_first[f * n_ctus + c + 1];
		const uint32_t n_tus = min(last - first, (uint32_t)kMaxCtuTus);

		// the CTU's TU list and source pixels: requested before the wait for the neighbours, they land meanwhile
		// (each lane checks and digests its TUs here, all at once, so that the TU loop -- a chain on ONE wave -- starts a TU with one LDS read and a few bit-field extracts)
		uint32_t digs[kMaxCtuTus / kLanes];      // TU i of the CTU: lane i % 64 of register i / 64 (the loop below fetches it with one v_readlane: no LDS round trip per TU)
		uint32_t angs[kMaxCtuTus / kLanes];      // its mode's angle and inverse angle (reconstruct_tu4): 40 scalar instructions per TU when derived in the TU loop
		{
			bool bad = false;
#pragma unroll
			for (int k = 0; k < kMaxCtuTus / kLanes; ++k)
			{
				const uint32_t i = (uint32_t)(k * kLanes + lane);
				digs[k] = 0; angs[k] = 0;
				if (i >= n_tus) continue;
				const hevcasm_hip_intra_tu d = A.tus[first + i];
				const int log2n = d.log2TrafoSize, mode = d.intraPredMode, tx = (int)d.x - v.x0, ty = (int)d.y - v.y0, n = 1 << (log2n & 7);
				const bool good = log2n >= 2 && log2n <= 5 && n <= ctu && mode <= 34 && tx >= 0 && ty >= 0 && tx + n <= ctu && ty + n <= ctu && tx % n == 0 && ty % n == 0 &&
					v.x0 + tx + n <= A.frame_w && v.y0 + ty + n <= A.frame_h;
				const uint32_t ux = (uint32_t)tx >> 2, uy = (uint32_t)ty >> 2;
				uint32_t z = 0;      // z-scan index of the TU's first 4x4 cell inside the CTU: where its levels lie
#pragma unroll
				for (int b = 0; b < 4; ++b) z |= ((ux >> b) & 1u) << (2 * b) | ((uy >> b) & 1u) << (2 * b + 1);
				// valid << 31 | z-scan index << 16 | mode << 10 | (log2 - 2) << 8 | (ty / 4) << 4 | tx / 4
				digs[k] = good ? (1u << 31) | (z << 16) | ((uint32_t)mode << 10) | ((uint32_t)(log2n - 2) << 8) | (uy << 4) | ux : 0u;
				const int angle = mode >= 2 && mode <= 34 ? intra_angle(mode) : 0;
				angs[k] = ((uint32_t)angle & 0xffffu) | ((uint32_t)(angle < 0 ? intra_inv_angle(mode) : 0) << 16);
				bad |= !good;
			}
			if (__any(bad) && lane == 0) __hip_atomic_fetch_or((gu32 *)(A.head + 1), kStatusBadTu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // a malformed TU is skipped and flagged
		}
		if constexpr (encode)
		{
			const uint8_t *src = A.src + f * A.pitch_src;
			for (int e = lane; e < ctu * ctu / 4; e += kLanes)      // dword pieces (the picture edge is a multiple of 4, rows need not be 16-byte multiples)
			{
				const int y = e / (ctu / 4), x = 4 * (e % (ctu / 4));
				if (v.x0 + x < A.frame_w && v.y0 + y < A.frame_h)
					*reinterpret_cast<uint32_t *>(L.srcs + y * 64 + x) = reinterpret_cast<const U1 *>(src + (ptrdiff_t)(v.y0 + y) * A.stride_src + v.x0 + x)->v;
			}
		}
		int16_t *ctu_levels = A.levels + ((size_t)f * n_ctus + c) * (size_t)(ctu * ctu);
		if constexpr (!encode)
		{
			for (int e = lane; e < ctu * ctu / 8; e += kLanes)      // 16-byte pieces; the CTU's block of the level array is ctu * ctu entries whatever the picture edge cuts off
				*reinterpret_cast<u32x4 *>(L.levs + 8 * e) = reinterpret_cast<const U4 *>(ctu_levels + 8 * e)->v;
		}
		for (int e = lane; e < 256 / 4; e += kLanes) reinterpret_cast<uint32_t *>(L.done)[e] = 0;

		// the edges this CTU reads: right column of the CTU to the left, bottom row of the one above (+ the first 32 samples of the one above-right), the corner
		// sample of the one above-left.  One granule per lane; a lane without one is done at once.
		{
			const int reach = min(32, ctu) / 4;
			const unsigned long long *g = nullptr;
			int kind = -1, k = 0;      // 0 left column, 1 top row, 2 top-right row, 3 corner
			if (lane < per_edge) { kind = 0; k = lane; if (cx > 0) g = A.granules + ((size_t)f * n_ctus + c - 1) * 2 * per_edge + per_edge + k; }
			else if (lane < 2 * per_edge) { kind = 1; k = lane - per_edge; if (cy > 0) g = A.granules + ((size_t)f * n_ctus + c - A.cols) * 2 * per_edge + k; }
			else if (lane < 2 * per_edge + reach) { kind = 2; k = lane - 2 * per_edge; if (cy > 0 && cx + 1 < A.cols) g = A.granules + ((size_t)f * n_ctus + c - A.cols + 1) * 2 * per_edge + k; }
			else if (lane == 2 * per_edge + reach) { kind = 3; if (cy > 0 && cx > 0) g = A.granules + ((size_t)f * n_ctus + c - A.cols - 1) * 2 * per_edge + per_edge - 1; }
			unsigned long long got = 0;
			const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
			for (uint32_t spins = 0;; ++spins)
			{
				if (g) got = granule_load(g);
				if (__all(!g || (got >> 32) == 1ull)) break;
				__builtin_amdgcn_s_sleep(2);
				if ((spins & 255u) == 255u)
				{
					const bool late = __builtin_amdgcn_s_memrealtime() - t0 > kSpinLimitTicks;
					const uint32_t aborted = __hip_atomic_load((const gu32 *)(A.head + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					if (late || aborted)
					{
						if (lane == 0)
						{
							__hip_atomic_fetch_or((gu32 *)(A.head + 1), kStatusTimeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
							__hip_atomic_store((gu32 *)(A.head + 2), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						}
						return;      // every wave that sees the abort word leaves: the grid drains
					}
				}
			}
			if (g)
			{
				const uint32_t val = (uint32_t)got;
				if (kind == 0)
				{
#pragma unroll
					for (int b = 0; b < 4; ++b) tile_at(L, -1, 4 * k + b) = (uint8_t)(val >> (8 * b));
				}
				else if (kind == 1) *reinterpret_cast<uint32_t *>(&tile_at(L, 4 * k, -1)) = val;
				else if (kind == 2) *reinterpret_cast<uint32_t *>(&tile_at(L, ctu + 4 * k, -1)) = val;
				else tile_at(L, -1, -1) = (uint8_t)(val >> 24);
			}
		}
		wave_sync();

		uint32_t holder = 0, holder_angles = 0;
#ifdef IF_STAMPS
		const unsigned long long loop_t0 = __builtin_amdgcn_s_memtime();
		unsigned long long if_acc[4] = { 0, 0, 0, 0 };
#endif
		for (uint32_t i = 0; i < n_tus; ++i)
		{
			if (i % kLanes == 0)      // (every 64th TU: the registers that hold the next 64)
			{
				const uint32_t which = i / kLanes;
				holder = which == 0 ? digs[0] : which == 1 ? digs[1] : which == 2 ? digs[2] : digs[3];
				holder_angles = which == 0 ? angs[0] : which == 1 ? angs[1] : which == 2 ? angs[2] : angs[3];
			}
			const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)holder, (int)(i % kLanes));
			const uint32_t angles = (uint32_t)__builtin_amdgcn_readlane((int)holder_angles, (int)(i % kLanes));
			if (!(d >> 31)) continue;
			const int tx = (int)(d & 15u) << 2, ty = (int)((d >> 4) & 15u) << 2, log2n = (int)((d >> 8) & 3u) + 2, mode = (int)((d >> 10) & 63u);
			const uint32_t z = (d >> 16) & 255u;
			int16_t *lev = (encode ? ctu_levels : L.levs) + 16u * z;
			int *cbf_out = A.cbf ? A.cbf + first + i : nullptr;
			switch (log2n)
			{
#ifndef IF_TU4_GENERAL
			case 2: reconstruct_tu4<ENCODE>(L, v, A, K4, recon, pred_out, tx, ty, mode, angles, lev, cbf_out, lane IF_STAMP_PASS); break;
#else
			case 2: reconstruct_tu<4, ENCODE>(L, v, A, recon, pred_out, tx, ty, mode, lev, cbf_out, lane); break;      // (A/B switch: the general path for 4x4 too)
#endif
#ifndef IF_TU8_GENERAL
			case 3: reconstruct_tu8<ENCODE>(L, v, A, K8, recon, pred_out, tx, ty, mode, angles, lev, cbf_out, lane); break;
#else
			case 3: reconstruct_tu<8, ENCODE>(L, v, A, recon, pred_out, tx, ty, mode, lev, cbf_out, lane); break;
#endif
			case 4: reconstruct_tu<16, ENCODE>(L, v, A, recon, pred_out, tx, ty, mode, lev, cbf_out, lane); break;
			default: reconstruct_tu<32, ENCODE>(L, v, A, recon, pred_out, tx, ty, mode, lev, cbf_out, lane); break;
			}
		}

#ifdef IF_STAMPS
		if (lane == 0)
		{
			atomicAdd(&g_if_stamps[4], __builtin_amdgcn_s_memtime() - loop_t0); atomicAdd(&g_if_stamps[5], (unsigned long long)n_tus);
			for (int k = 0; k < 4; ++k) atomicAdd(&g_if_stamps[k], if_acc[k]);
		}
#endif
		// publish: bottom row, then right column; four samples + the tag in one 8-byte store each (written once, by this lane only)
		{
			unsigned long long *mine = A.granules + ((size_t)f * n_ctus + c) * 2 * per_edge;
			if (lane < per_edge) granule_store(mine + lane, *reinterpret_cast<const uint32_t *>(&tile_at(L, 4 * lane, ctu - 1)));
			else if (lane < 2 * per_edge)
			{
				const int k = lane - per_edge;
				granule_store(mine + lane, (uint32_t)tile_at(L, ctu - 1, 4 * k) | ((uint32_t)tile_at(L, ctu - 1, 4 * k + 1) << 8) |
					((uint32_t)tile_at(L, ctu - 1, 4 * k + 2) << 16) | ((uint32_t)tile_at(L, ctu - 1, 4 * k + 3) << 24));
			}
		}
		wave_sync();
	}
}

// Two builds of the loop.  The encoder's needs 271 registers (256 + 15 spilled to accumulation registers): ONE wave per SIMD, four workgroups per CU -- the fastest
// form of a wave, and enough waves whenever the wavefront offers at most 4 x CUs CTUs at a time (64 frames of 1080p: 960).  `dense` holds the same code to 256
// registers (13 spills to scratch, 3 - 6 % slower per wave) so that two waves share a SIMD and six workgroups a CU: what a call of hundreds of frames wants
// (256 frames of 1080p: 9.2 -> 7.2 ms; 64 frames 4.4 -> 4.7 ms, hence not always: profiles/r06_intra_waves_ab.txt).  The host picks by the call's concurrency.
template <bool ENCODE>
__global__ __launch_bounds__(kLanes) void intra_frames_kernel(const IntraFramesArgs A)
{
	intra_frames_body<ENCODE>(A);
}

template <bool ENCODE>
__global__ __launch_bounds__(kLanes) __attribute__((amdgpu_waves_per_eu(2, 2))) void intra_frames_dense_kernel(const IntraFramesArgs A)
{
	intra_frames_body<ENCODE>(A);
}

}  // namespace hevc

using namespace hevc;

static size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }

struct WorkspaceLayout { size_t head, order, granules, total; uint32_t n_items; int cols, rows; };

static WorkspaceLayout workspace_layout(int frame_w, int frame_h, int ctu, size_t n_frames)
{
	WorkspaceLayout w = {};
	w.cols = (frame_w + ctu - 1) / ctu; w.rows = (frame_h + ctu - 1) / ctu;
	const size_t items = (size_t)w.cols * w.rows * n_frames;
	w.n_items = (uint32_t)items;
	// the zeroed part first (head + granules, one memset from the start of the allocation, a multiple of 16 bytes), then the order table
	w.head = 0;
	w.granules = 64;
	w.order = round_up(w.granules + items * 2 * (ctu / 4) * 8, 64);
	w.total = round_up(w.order + 4 * items, 64);
	return w;
}

static int check_geometry(int frame_w, int frame_h, int ctu, size_t n_frames)
{
	if (ctu != 16 && ctu != 32 && ctu != 64) return set_error(HEVCASM_HIP_EINVAL, "ctu must be 16, 32 or 64");
	if (frame_w <= 0 || frame_h <= 0 || (frame_w & 3) || (frame_h & 3) || frame_w > 65532 || frame_h > 65532) return set_error(HEVCASM_HIP_EINVAL, "frame dimensions must be multiples of 4 below 65536");
	if ((frame_w + ctu - 1) / ctu > 1023 || (frame_h + ctu - 1) / ctu > 1023) return set_error(HEVCASM_HIP_EINVAL, "more than 1023 CTUs in a row or column");
	if (n_frames > 4096) return set_error(HEVCASM_HIP_EINVAL, "at most 4096 frames per call");
	return 0;
}

extern "C" size_t hevcasm_hip_intra_frames_workspace_bytes(int frame_w, int frame_h, int ctu, size_t n_frames)
{
	if (check_geometry(frame_w, frame_h, ctu, n_frames)) return 0;
	return workspace_layout(frame_w, frame_h, ctu, n_frames).total;
}

extern "C" int hevcasm_hip_intra_reconstruct_frames(
	uint8_t *recon, ptrdiff_t stride_recon, size_t frame_pitch_recon,
	const uint8_t *src, ptrdiff_t stride_src, size_t frame_pitch_src,
	int16_t *levels, int *cbf, uint8_t *pred_out,
	const hevcasm_hip_intra_tu *tus, const uint32_t *ctu_first, const hevcasm_hip_intra_qp *qp,
	int frame_w, int frame_h, int ctu, int cIdx, int flags, size_t n_frames,
	void *workspace, void *stream)
{
	if (int e = check_device()) return e;
	if (int e = check_geometry(frame_w, frame_h, ctu, n_frames)) return e;
	if (n_frames == 0) return 0;
	if (!recon || !levels || !tus || !ctu_first || !qp || !workspace) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	if (reinterpret_cast<uintptr_t>(workspace) & 15) return set_error(HEVCASM_HIP_EINVAL, "workspace must be 16-byte aligned");
	if (reinterpret_cast<uintptr_t>(levels) & 3) return set_error(HEVCASM_HIP_EINVAL, "levels must be 4-byte aligned");
	if (stride_recon < frame_w || (src && stride_src < frame_w)) return set_error(HEVCASM_HIP_EINVAL, "strides must be at least the frame width");
	for (int k = 0; k < 4; ++k)
	{
		// the reference's own preconditions (quantize.c:162-168) and the int16 scale of its dequantiser's SIMD form
		if (src && (qp[k].q_scale < 0 || qp[k].q_scale >= 0x8000 || qp[k].q_offset < 0 || qp[k].q_offset >= 0x8000 || qp[k].q_shift < 16 || qp[k].q_shift > 27))
			return set_error(HEVCASM_HIP_EINVAL, "quantiser parameters outside 0 <= scale, offset < 0x8000, 16 <= shift <= 27");
		if (qp[k].dq_scale < -32768 || qp[k].dq_scale > 32767 || qp[k].dq_shift < 1 || qp[k].dq_shift > 31)
			return set_error(HEVCASM_HIP_EINVAL, "dequantiser parameters outside int16 scale, 1 <= shift <= 31");
	}
	const WorkspaceLayout w = workspace_layout(frame_w, frame_h, ctu, n_frames);
	hipStream_t st = as_stream(stream);
	uint8_t *ws = static_cast<uint8_t *>(workspace);

	IntraFramesArgs a = {};
	a.recon = recon; a.src = src; a.pred_out = pred_out;
	a.stride_recon = stride_recon; a.stride_src = stride_src; a.pitch_recon = frame_pitch_recon; a.pitch_src = frame_pitch_src;
	a.levels = levels; a.cbf = cbf; a.tus = tus; a.ctu_first = ctu_first;
	for (int k = 0; k < 4; ++k) a.qp[k] = qp[k];
	a.frame_w = frame_w; a.frame_h = frame_h; a.ctu = ctu; a.cols = w.cols; a.rows = w.rows; a.cIdx = cIdx; a.strong = (flags & HEVCASM_HIP_INTRA_STRONG_SMOOTHING) ? 1 : 0;
	a.n_frames = (uint32_t)n_frames; a.n_items = w.n_items;
	a.head = reinterpret_cast<uint32_t *>(ws + w.head);
	a.order = reinterpret_cast<uint32_t *>(ws + w.order);
	a.granules = reinterpret_cast<unsigned long long *>(ws + w.granules);

	HEVC_HIP_TRY(hipMemsetAsync(ws, 0, w.order, st));      // ticket counter, status, abort word and every granule's tag: zero before every launch
	hipLaunchKernelGGL(intra_frames_order_kernel, dim3(ceil_div(w.n_items, 256)), dim3(256), 0, st, a.order, w.cols, w.rows, a.n_frames);
	if (int e = launch_status("intra_frames_order_kernel")) return e;
	// Single-wave workgroups, 22.4 KiB of LDS each (decoder: 26.4); any grid size is correct (tickets), more than the work is pointless.  The wavefront offers
	// min(rows, ceil(cols / 2)) CTUs of a frame at a time: when all frames' together fit one wave per SIMD that is the grid (and the one-wave build of the loop),
	// otherwise every workgroup slot of the device (six per CU, decoder five) with the two-waves-per-SIMD build.
	const size_t cus = (size_t)device_cus();
	const size_t concurrency = n_frames * (size_t)std::min(w.rows, (w.cols + 1) / 2);
	const bool dense = concurrency > 4 * cus;
	const uint32_t grid = (uint32_t)std::min<size_t>(w.n_items, cus * (dense ? (src ? 6 : 5) : 4));
	if (src && dense) hipLaunchKernelGGL(intra_frames_dense_kernel<true>, dim3(grid), dim3(kLanes), 0, st, a);
	else if (src) hipLaunchKernelGGL(intra_frames_kernel<true>, dim3(grid), dim3(kLanes), 0, st, a);
	else if (dense) hipLaunchKernelGGL(intra_frames_dense_kernel<false>, dim3(grid), dim3(kLanes), 0, st, a);
	else hipLaunchKernelGGL(intra_frames_kernel<false>, dim3(grid), dim3(kLanes), 0, st, a);
	return launch_status("intra_frames_kernel");
}

extern "C" int hevcasm_hip_intra_frames_status(const void *workspace, void *stream)
{
	if (int e = check_device()) return e;
	if (!workspace) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	uint32_t head[4] = { 0, 0, 0, 0 };
	HEVC_HIP_TRY(hipMemcpyAsync(head, workspace, sizeof(head), hipMemcpyDeviceToHost, as_stream(stream)));
	HEVC_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
	if (head[1] & kStatusTimeout) return set_error(HEVCASM_HIP_EINVAL, "intra frames: a wait for a neighbouring CTU timed out (TU list not in decoding order, or the device was lost)");
	if (head[1] & kStatusBadTu) return set_error(HEVCASM_HIP_EINVAL, "intra frames: a malformed TU descriptor was skipped (size, mode, alignment or position)");
	return 0;
}

#ifdef IF_STAMPS
extern "C" int hevcasm_hip_debug_intra_stamps(unsigned long long *out8, int reset)
{
	if (out8) HEVC_HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(hevc::g_if_stamps), 64));
	if (reset) { unsigned long long z[8] = {}; HEVC_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(hevc::g_if_stamps), z, 64)); }
	return 0;
}
#endif
