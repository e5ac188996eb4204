// subpel.hip -- fused quarter-pel refinement: luma interpolation -> SAD or Hadamard SATD -> arg-min, all 49
// displacements (qx, qy) in [-3, 3]^2 of a block in one pass.
//
// The composition it replaces is 49 x { hevcasm_pred_uni_8to8 (pred_inter.h:56, C bodies
// pred_inter.c:141-228) into a scratch block, hevcasm_sad (sad.c:47-60) against the source }.
// Done that way every candidate prediction is written to and read back from memory; here the
// (W + 8) x (H + 8) reference window is fetched once and nothing but {qx, qy, cost} is written.
//
// All four of the reference's filter paths are one formula: with the "fraction 0" filter taken as
// {0,0,0,64,0,0,0,0}, pred = sat8((sum_v cv[t] * sum_h ch[u] * p + 2048) >> 12).  That is exact:
// copy, H-only ((h + 32) >> 6) and V-only ((v + 32) >> 6) are the same expression with a factor 64
// that the shift divides back out, and the 8-bit intermediate needs no rounding (shift 0,
// pred_inter.c:141-171), so vertical-then-horizontal equals the reference's horizontal-then-vertical.
//
// Per block (LPB lanes, several blocks per workgroup when the block is small):
//   - window, source block -> LDS, once;
//   - for each of the 7 qy: V pass (v_pk_mad_i16 on unpacked byte pairs, coefficients in SGPRs)
//     writes an H x (W + 8) int16 plane to LDS; H pass: a lane reads the 12 plane values around 4
//     output pixels once and evaluates all 7 qx from registers -- v_dot2_i32_i16 against literal
//     coefficient pairs, two pixels per v_ashr_pk_u8_i32, one v_sad_u8 per 4 pixels;
//   - the 7 partial SADs are summed over DPP groups of <= 16 lanes, group leaders ds_add into
//     cost[49]; after the last qy lane 0 picks the minimum in raster order of (qy, qx).
// The int16 plane is double-buffered: one barrier per qy.  VALU-bound (about 300 lane-ops per
// pixel), not HBM-bound: algorithmic bytes per block are (W + 8)(H + 8) + W H + 8 + 12.

#include "common.hpp"
#include <algorithm>
#include <vector>

namespace hevc {

typedef short s16x2 __attribute__((ext_vector_type(2)));

// luma filter, pred_inter.c:53-83 (fraction 0 = pass-through scaled by 64, see above)
struct LumaTaps { int c[4][8]; };
constexpr LumaTaps kLuma = { {
	{ 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 },
	{ -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } } };

struct LumaPairs { uint32_t v[4][8]; };
constexpr LumaPairs make_luma_pairs()
{
	LumaPairs t = {};
	for (int f = 0; f < 4; ++f)
		for (int k = 0; k < 8; ++k) t.v[f][k] = ((uint32_t)kLuma.c[f][k] & 0xffffu) | ((uint32_t)kLuma.c[f][k] << 16);
	return t;
}
__constant__ LumaPairs kLumaPairs = make_luma_pairs();   // (c, c): one coefficient for both halves of v_pk_mad_i16

__device__ __forceinline__ uint32_t pk_mad16(uint32_t a, uint32_t b, uint32_t c)
{
	const s16x2 r = __builtin_bit_cast(s16x2, a) * __builtin_bit_cast(s16x2, b) + __builtin_bit_cast(s16x2, c);
	return __builtin_bit_cast(uint32_t, r);
}

// sum_t c[FX][t] * v[S + t], v = the 12 int16 held pairwise in d[0..5]
template <int FX, int S> __device__ __forceinline__ int hsum(const uint32_t (&d)[6], int acc)
{
	constexpr int N = (S & 1) ? 5 : 4;
#pragma unroll
	for (int k = 0; k < N; ++k)
	{
		const int lo = (S & 1) ? (k == 0 ? 0 : kLuma.c[FX][2 * k - 1]) : kLuma.c[FX][2 * k];
		const int hi = (S & 1) ? (k == 4 ? 0 : kLuma.c[FX][2 * k]) : kLuma.c[FX][2 * k + 1];
		if (lo == 0 && hi == 0) continue;
		const uint32_t c = ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16);
		acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, d[S / 2 + k]), __builtin_bit_cast(s16x2, c), acc, false);
	}
	return acc;
}

// SAD of 4 predicted pixels of displacement QX (quarter pels, -3..3) against the source dword
template <int QX> __device__ __forceinline__ uint32_t candidate_sad(const uint32_t (&d)[6], uint32_t src4, uint32_t acc)
{
	constexpr int FX = QX & 3, IX = QX < 0 ? -1 : 0;
	const int o0 = hsum<FX, 0 + IX + 1>(d, 2048), o1 = hsum<FX, 1 + IX + 1>(d, 2048);
	const int o2 = hsum<FX, 2 + IX + 1>(d, 2048), o3 = hsum<FX, 3 + IX + 1>(d, 2048);
	const uint32_t lo = __builtin_amdgcn_ashr_pk_u8_i32(o0, o1, 12);
	const uint32_t hi = __builtin_amdgcn_ashr_pk_u8_i32(o2, o3, 12);
	return __builtin_amdgcn_sad_u8(src4, __builtin_amdgcn_perm(hi, lo, 0x05040100), acc);
}

template <int CTRL> __device__ __forceinline__ uint32_t add_dpp(uint32_t v)
{
	return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}

// ---- Hadamard SATD of the candidate (metric 1) ------------------------------------------------
//
// hevcasm_hadamard_satd (hadamard.c:76-111) over TILE x TILE tiles.  Lanes are dealt so that the four
// rows of a 4x4 sub-block are the four lanes of a quad, and for 8x8 tiles the four 4x4 sub-blocks of
// a tile are the four quads of one 16-lane DPP row.  A lane holds its row's 4 differences as two
// packed int16 pairs (every Hadamard coefficient of an 8x8 block of byte differences is within
// +-16320): horizontal 4-point butterflies in the lane, vertical ones across the quad by DPP, and --
// H8 = H2 (x) H4 -- two more butterfly stages across the quads (ds_swizzle: lane ^ 4, lane ^ 8).
// The sum of magnitudes does not depend on the order of the basis, so any complete set of
// butterflies gives the reference's sum; its rounding (sum + n/4) / (n/2) is applied per tile.
__device__ __forceinline__ uint32_t pk_add16(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (s16x2)(__builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b))); }
__device__ __forceinline__ uint32_t pk_sub16(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (s16x2)(__builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b))); }
// v * sign + p with sign = (+1, +1) or (-1, -1): one v_pk_mad_i16
__device__ __forceinline__ uint32_t butterfly(uint32_t v, uint32_t sign, uint32_t p) { return pk_mad16(v, sign, p); }

struct SatdLane
{
	uint32_t s02, s13;             // the source row's pixels (0, 2) and (1, 3) as int16 pairs
	uint32_t sign1, sign2, sign4, sign8;   // (-1, -1) where the lane is the upper partner of a butterfly, else (+1, +1)
	bool leader;                   // first lane of its tile
};

template <int QX, int TILE> __device__ __forceinline__ uint32_t candidate_satd(const uint32_t (&d)[6], const SatdLane &ln, uint32_t acc)
{
	constexpr int FX = QX & 3, IX = QX < 0 ? -1 : 0;
	const int o0 = hsum<FX, 0 + IX + 1>(d, 2048), o1 = hsum<FX, 1 + IX + 1>(d, 2048);
	const int o2 = hsum<FX, 2 + IX + 1>(d, 2048), o3 = hsum<FX, 3 + IX + 1>(d, 2048);
	// predicted pixels (0, 2) and (1, 3), saturated to bytes, widened to int16 pairs
	const uint32_t p02 = __builtin_amdgcn_perm(0, __builtin_amdgcn_ashr_pk_u8_i32(o0, o2, 12), 0x0c010c00);
	const uint32_t p13 = __builtin_amdgcn_perm(0, __builtin_amdgcn_ashr_pk_u8_i32(o1, o3, 12), 0x0c010c00);
	const uint32_t d02 = pk_sub16(ln.s02, p02), d13 = pk_sub16(ln.s13, p13);
	// row transform: (d0+d1, d2+d3), (d0-d1, d2-d3) -> all four sign patterns
	const uint32_t e = pk_add16(d02, d13), f = pk_sub16(d02, d13);
	const uint32_t x = __builtin_amdgcn_perm(f, e, 0x05040100), y = __builtin_amdgcn_perm(f, e, 0x07060302);
	uint32_t v[2] = { pk_add16(x, y), pk_sub16(x, y) };
#pragma unroll
	for (int k = 0; k < 2; ++k)
	{
		// column transform across the quad's four rows
		v[k] = butterfly(v[k], ln.sign1, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v[k], 0xB1, 0xf, 0xf, true));
		v[k] = butterfly(v[k], ln.sign2, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v[k], 0x4E, 0xf, 0xf, true));
		if (TILE == 8)
		{
			v[k] = butterfly(v[k], ln.sign4, (uint32_t)__builtin_amdgcn_ds_swizzle((int)v[k], 0x101f));   // lane ^ 4
			v[k] = butterfly(v[k], ln.sign8, (uint32_t)__builtin_amdgcn_ds_swizzle((int)v[k], 0x201f));   // lane ^ 8
		}
	}
	int sum = 0;
#pragma unroll
	for (int k = 0; k < 2; ++k)
	{
		const s16x2 a = __builtin_bit_cast(s16x2, v[k]);
		const s16x2 m = __builtin_elementwise_max(a, (s16x2)(-a));
		sum = __builtin_amdgcn_sdot2(m, (s16x2){ 1, 1 }, sum, false);
	}
	uint32_t t = (uint32_t)sum;
	t = add_dpp<0xB1>(t);
	t = add_dpp<0x4E>(t);
	if (TILE == 8) { t = add_dpp<0x141>(t); t = add_dpp<0x140>(t); }
	const uint32_t cost = (t + TILE / 4) / (TILE / 2);
	return acc + (ln.leader ? cost : 0u);
}

// (row, column) of item `it` in a grid `w` items wide, walked with a fixed stride: one division when the
// walk starts, additions afterwards (an integer division per item is a dozen quarter-rate instructions --
// a fifth of an item's work in the H pass)
struct Walk
{
	int y, x, dy, dx, w;
	__device__ __forceinline__ Walk(int first, int stride, int width) : y(first / width), x(first % width), dy(stride / width), dx(stride % width), w(width) {}
	__device__ __forceinline__ void next()
	{
		y += dy; x += dx;
		if (x >= w) { x -= w; ++y; }
	}
};

struct SubpelGeometry
{
	int W, H;
	int lpb;        // lanes per block
	int bpw;        // blocks per workgroup
	int seg;        // lanes per DPP reduction group: gcd(lpb, 16)
	int blk_bytes;  // LDS bytes per block
	int off_src, off_plane, off_cost;   // byte offsets inside a block's LDS
};

// a W x H block's geometry (host: the one-size launcher; device: the runs kernel, from its run's size)
__host__ __device__ inline SubpelGeometry subpel_geometry(int W, int H)
{
	SubpelGeometry g;
	g.W = W; g.H = H;
	const int items = W * H / 4;
	g.lpb = items < kThreads ? items : kThreads;
	g.bpw = kThreads / g.lpb;
	g.seg = (g.lpb % 16 == 0) ? 16 : (g.lpb % 8 == 0) ? 8 : 4;
	const int win_bytes = (H + 8) * (W + 8), src_bytes = W * H, plane_bytes = 2 * H * (W + 8) * 2;
	g.off_src = (win_bytes + 7) & ~7;
	g.off_plane = (g.off_src + src_bytes + 7) & ~7;
	g.off_cost = g.off_plane + plane_bytes;
	g.blk_bytes = (g.off_cost + 49 * 4 + 15) & ~15;
	return g;
}

// METRIC 0 = SAD, 1 = Hadamard SATD over TILE x TILE tiles (TILE = 8 when both dimensions allow, else 4)
// `wg`: the workgroup's index among the ceil(n_blocks / g.bpw) of this geometry
template <int METRIC, int TILE>
__device__ __forceinline__ void subpel_refine_workgroup(
	const uint8_t *__restrict__ src, ptrdiff_t stride_src, const uint8_t *__restrict__ ref, ptrdiff_t stride_ref,
	const uint32_t *__restrict__ src_off, const uint32_t *__restrict__ ref_off, int *__restrict__ result,
	const SubpelGeometry &g, uint32_t n_blocks, uint32_t wg, uint8_t *lds)
{
	const int W = g.W, H = g.H;
	const int w4 = W / 4;              // output dwords per row
	const int wd = (W + 8) / 4;        // window dwords per row
	const int pitch = W + 8;           // plane row, int16 elements
	const int b_local = threadIdx.x / g.lpb, l = threadIdx.x - b_local * g.lpb;
	const uint32_t blk = wg * g.bpw + b_local;
	const bool active = b_local < g.bpw && blk < n_blocks;

	uint8_t *mine = lds + (active ? b_local : 0) * g.blk_bytes;
	uint32_t *win = reinterpret_cast<uint32_t *>(mine);                       // (H + 8) x wd dwords, pixel (x, y) at row y + 4, byte x + 4
	uint32_t *srcT = reinterpret_cast<uint32_t *>(mine + g.off_src);          // H x w4 dwords
	uint8_t *plane = mine + g.off_plane;                                       // 2 x H x pitch int16
	uint32_t *cost = reinterpret_cast<uint32_t *>(mine + g.off_cost);         // 49
	const int plane_bytes = H * pitch * 2;

	if (active)
	{
		const uint8_t *rp = ref + ref_off[blk] - 4 * stride_ref - 4;
		for (int it = l; it < (H + 8) * wd; it += g.lpb)
		{
			const int r = it / wd, x = it - r * wd;
			win[it] = reinterpret_cast<const U1 *>(rp + (ptrdiff_t)r * stride_ref + 4 * x)->v;
		}
		const uint8_t *sp = src + src_off[blk];
		for (int it = l; it < H * w4; it += g.lpb)
		{
			const int y = it / w4, x = it - y * w4;
			srcT[it] = reinterpret_cast<const U1 *>(sp + (ptrdiff_t)y * stride_src + 4 * x)->v;
		}
		for (int i = l; i < 49; i += g.lpb) cost[i] = 0;
	}
	__syncthreads();

	for (int qi = 0; qi < 7; ++qi)
	{
		const int qy = qi - 3;
		uint8_t *pl = plane + (qi & 1) * plane_bytes;
		if (active)
		{
			// vertical pass: plane[y][x + 4] = sum_t cv[t] * pixel(x, y + (qy >> 2) - 3 + t)
			const int row0 = qy < 0 ? 0 : 1;          // window row of tap 0 for output row 0
			uint32_t cv[8];
#pragma unroll
			for (int t = 0; t < 8; ++t) cv[t] = kLumaPairs.v[qy & 3][t];
			Walk vw(l, g.lpb, wd);
			for (int it = l; it < H * wd; it += g.lpb, vw.next())
			{
				const int y = vw.y, x = vw.x;
				uint32_t even = 0, odd = 0;           // (v0, v2), (v1, v3)
#pragma unroll
				for (int t = 0; t < 8; ++t)
				{
					const uint32_t px = win[(y + row0 + t) * wd + x];
					even = pk_mad16(__builtin_amdgcn_perm(0, px, 0x0c020c00), cv[t], even);
					odd = pk_mad16(__builtin_amdgcn_perm(0, px, 0x0c030c01), cv[t], odd);
				}
				u32x2 out;
				out.x = __builtin_amdgcn_perm(odd, even, 0x05040100);   // (v0, v1)
				out.y = __builtin_amdgcn_perm(odd, even, 0x07060302);   // (v2, v3)
				*reinterpret_cast<u32x2 *>(pl + 2 * (y * pitch + 4 * x)) = out;
			}
		}
		__syncthreads();
		if (active)
		{
			uint32_t acc[7] = { 0, 0, 0, 0, 0, 0, 0 };
			// METRIC 0: item = (row, dword column).  SATD: item = (tile, lane in tile); tiles are walked, the
			// lane's place inside its tile is fixed (lpb is a multiple of the tile's lanes)
			constexpr int TL = METRIC == 0 ? 1 : TILE == 4 ? 4 : 16;      // lanes per tile
			Walk hw(l / TL, g.lpb / TL, METRIC == 0 ? w4 : TILE == 4 ? w4 : w4 / 2);
			for (int it = l; it < H * w4; it += g.lpb, hw.next())
			{
				int y, x;
				if (METRIC == 0)
				{
					y = hw.y; x = hw.x;
				}
				else if (TILE == 4)
				{
					// a quad of lanes = the four rows of one 4x4 tile
					y = 4 * hw.y + (l & 3); x = hw.x;
				}
				else
				{
					// 16 lanes = one 8x8 tile: quad q = its 4x4 sub-block (q >> 1 down, q & 1 across)
					const int q = (l >> 2) & 3;
					y = 8 * hw.y + 4 * (q >> 1) + (l & 3); x = 2 * hw.x + (q & 1);
				}
				// plane columns 4x .. 4x + 11 = pixels 4x - 4 .. 4x + 7
				const u32x2 *p = reinterpret_cast<const u32x2 *>(pl + 2 * (y * pitch + 4 * x));
				const u32x2 a = p[0], b = p[1], c = p[2];
				const uint32_t d[6] = { a.x, a.y, b.x, b.y, c.x, c.y };
				const uint32_t s4 = srcT[y * w4 + x];
				if (METRIC == 0)
				{
					acc[0] = candidate_sad<-3>(d, s4, acc[0]);
					acc[1] = candidate_sad<-2>(d, s4, acc[1]);
					acc[2] = candidate_sad<-1>(d, s4, acc[2]);
					acc[3] = candidate_sad<0>(d, s4, acc[3]);
					acc[4] = candidate_sad<1>(d, s4, acc[4]);
					acc[5] = candidate_sad<2>(d, s4, acc[5]);
					acc[6] = candidate_sad<3>(d, s4, acc[6]);
				}
				else
				{
					SatdLane ln;
					ln.s02 = __builtin_amdgcn_perm(0, s4, 0x0c020c00);
					ln.s13 = __builtin_amdgcn_perm(0, s4, 0x0c030c01);
					ln.sign1 = (l & 1) ? 0xffffffffu : 0x00010001u;
					ln.sign2 = (l & 2) ? 0xffffffffu : 0x00010001u;
					ln.sign4 = (l & 4) ? 0xffffffffu : 0x00010001u;
					ln.sign8 = (l & 8) ? 0xffffffffu : 0x00010001u;
					ln.leader = (l & (TILE == 8 ? 15 : 3)) == 0;
					acc[0] = candidate_satd<-3, TILE>(d, ln, acc[0]);
					acc[1] = candidate_satd<-2, TILE>(d, ln, acc[1]);
					acc[2] = candidate_satd<-1, TILE>(d, ln, acc[2]);
					acc[3] = candidate_satd<0, TILE>(d, ln, acc[3]);
					acc[4] = candidate_satd<1, TILE>(d, ln, acc[4]);
					acc[5] = candidate_satd<2, TILE>(d, ln, acc[5]);
					acc[6] = candidate_satd<3, TILE>(d, ln, acc[6]);
				}
			}
#pragma unroll
			for (int k = 0; k < 7; ++k)
			{
				uint32_t v = acc[k];
				v = add_dpp<0xB1>(v);                      // quad_perm [1,0,3,2]
				v = add_dpp<0x4E>(v);                      // quad_perm [2,3,0,1]
				if (g.seg >= 8) v = add_dpp<0x141>(v);     // row_half_mirror
				if (g.seg >= 16) v = add_dpp<0x140>(v);    // row_mirror
				if ((l & (g.seg - 1)) == 0) atomicAdd(&cost[qi * 7 + k], v);
			}
		}
	}
	__syncthreads();
	if (active && l == 0)
	{
		uint32_t best = 0xffffffffu;
		for (int i = 0; i < 49; ++i) best = min(best, (cost[i] << 6) | (uint32_t)i);   // SAD <= 64*64*255 < 2^20; SATD <= 64 tiles * 2^18 = 2^24
		const int i = (int)(best & 63);
		result[3 * (size_t)blk + 0] = i % 7 - 3;
		result[3 * (size_t)blk + 1] = i / 7 - 3;
		result[3 * (size_t)blk + 2] = (int)(best >> 6);
	}
}

template <int METRIC, int TILE>
__global__ __launch_bounds__(kThreads) void subpel_refine_kernel(
	const uint8_t *__restrict__ src, ptrdiff_t stride_src, const uint8_t *__restrict__ ref, ptrdiff_t stride_ref,
	const uint32_t *__restrict__ src_off, const uint32_t *__restrict__ ref_off, int *__restrict__ result,
	SubpelGeometry g, uint32_t n_blocks)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	subpel_refine_workgroup<METRIC, TILE>(src, stride_src, ref, stride_ref, src_off, ref_off, result, g, n_blocks, xcd_contiguous(blockIdx.x, gridDim.x), lds);
}

// ---- one launch for a frame's partition: runs of equally sized blocks (round 6; see me.hip's me_full_search_runs_kernel) ---------------------------
constexpr int kMaxSubpelRuns = 96;
struct SubpelRunDev { uint32_t wg_count, first, count; uint16_t W, H; };
struct SubpelRunsArgs
{
	const uint8_t *src, *ref;
	ptrdiff_t stride_src, stride_ref;
	const uint32_t *src_off, *ref_off;
	int *result;
	uint32_t wg_start[kMaxSubpelRuns];      // first workgroup of run i; unused entries hold 0xffffffff
	SubpelRunDev runs[kMaxSubpelRuns];
};

template <int METRIC>
__global__ __launch_bounds__(kThreads) void subpel_refine_runs_kernel(const SubpelRunsArgs a)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	uint32_t at_or_before = 0, start = 0;
#pragma unroll
	for (int i = 0; i < kMaxSubpelRuns; ++i)
	{
		const bool in = a.wg_start[i] <= blockIdx.x;
		at_or_before += in ? 1u : 0u;
		start = in ? a.wg_start[i] : start;      // starts ascend: the last one taken is the run's
	}
	const SubpelRunDev r = a.runs[at_or_before - 1];
	const SubpelGeometry g = subpel_geometry(r.W, r.H);
	const uint32_t wg = xcd_contiguous(blockIdx.x - start, r.wg_count);
#define HEVC_RUN(TILE) subpel_refine_workgroup<METRIC, TILE>(a.src, a.stride_src, a.ref, a.stride_ref, a.src_off + r.first, a.ref_off + r.first, a.result + 3 * (size_t)r.first, g, r.count, wg, lds)
	if (METRIC == 0) HEVC_RUN(4);
	else if (r.W % 8 == 0 && r.H % 8 == 0) HEVC_RUN(8);
	else HEVC_RUN(4);
#undef HEVC_RUN
}

}  // namespace hevc

using namespace hevc;

extern "C" int hevcasm_hip_subpel_refine_runs(
	const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref,
	const uint32_t *src_off, const uint32_t *ref_off, int *result,
	const hevcasm_hip_cost_run *runs, int n_runs, int metric, void *stream)
{
	if (int e = check_device()) return e;
	if (n_runs < 0) return set_error(HEVCASM_HIP_EINVAL, "negative run count");
	if (metric != 0 && metric != 1) return set_error(HEVCASM_HIP_EINVAL, "metric must be 0 (SAD) or 1 (Hadamard SATD)");
	if (n_runs == 0) return 0;
	if (!runs || !src || !ref || !src_off || !ref_off || !result) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	std::vector<SubpelRunDev> table;
	table.reserve((size_t)n_runs);
	size_t shmem = 0;
	for (int i = 0; i < n_runs; ++i)
	{
		const int W = (int)(runs[i].rect >> 8), H = (int)(runs[i].rect & 0xff);
		if (W < 4 || W > 64 || H < 4 || H > 64 || (W & 3) || (H & 3) || (runs[i].rect >> 16)) return set_error(HEVCASM_HIP_EINVAL, "block size must be a multiple of 4 in 4..64");
		if ((uint64_t)runs[i].first + runs[i].count > 0x7fffffffull / 3) return set_error(HEVCASM_HIP_EINVAL, "n_blocks too large");
		const SubpelGeometry g = subpel_geometry(W, H);
		const size_t lds = (size_t)g.blk_bytes * g.bpw;
		if (lds > 64 * 1024) return set_error(HEVCASM_HIP_EINVAL, "block does not fit in 64 KiB of LDS");
		if (runs[i].count == 0) continue;
		shmem = lds > shmem ? lds : shmem;
		table.push_back(SubpelRunDev{ ceil_div((size_t)runs[i].count, (size_t)g.bpw), runs[i].first, runs[i].count, (uint16_t)W, (uint16_t)H });
	}
	// the largest blocks first: the launch ends when its last workgroup does
	std::stable_sort(table.begin(), table.end(), [](const SubpelRunDev &x, const SubpelRunDev &y) { return (int)x.W * x.H > (int)y.W * y.H; });
	for (size_t base = 0; base < table.size(); base += kMaxSubpelRuns)      // more runs than one table holds: the next launch
	{
		SubpelRunsArgs a = {};
		a.src = src; a.ref = ref; a.stride_src = stride_src; a.stride_ref = stride_ref; a.src_off = src_off; a.ref_off = ref_off; a.result = result;
		uint32_t wgs = 0, used = 0;
		for (int i = 0; i < kMaxSubpelRuns; ++i) a.wg_start[i] = 0xffffffffu;
		for (size_t i = base; i < table.size() && i < base + kMaxSubpelRuns; ++i)
		{
			a.runs[used] = table[i];
			a.wg_start[used++] = wgs;
			wgs += table[i].wg_count;
		}
		if (metric == 0) hipLaunchKernelGGL(subpel_refine_runs_kernel<0>, dim3(wgs), dim3(kThreads), shmem, as_stream(stream), a);
		else hipLaunchKernelGGL(subpel_refine_runs_kernel<1>, dim3(wgs), dim3(kThreads), shmem, as_stream(stream), a);
		if (int e = launch_status("subpel_refine_runs_kernel")) return e;
	}
	return 0;
}

extern "C" int hevcasm_hip_subpel_refine_batch(
	const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref,
	const uint32_t *src_off, const uint32_t *ref_off, int *result,
	uint32_t rect, int metric, size_t n_blocks, void *stream)
{
	if (int e = check_device()) return e;
	const int W = (int)(rect >> 8), H = (int)(rect & 0xff);
	if (W < 4 || W > 64 || H < 4 || H > 64 || (W & 3) || (H & 3)) return set_error(HEVCASM_HIP_EINVAL, "block size must be a multiple of 4 in 4..64");
	if (metric != 0 && metric != 1) return set_error(HEVCASM_HIP_EINVAL, "metric must be 0 (SAD) or 1 (Hadamard SATD)");
	if (n_blocks == 0) return 0;
	if (n_blocks > 0x7fffffffull / 3) return set_error(HEVCASM_HIP_EINVAL, "n_blocks too large");
	if (!src || !ref || !src_off || !ref_off || !result) return set_error(HEVCASM_HIP_EINVAL, "null pointer");

	const SubpelGeometry g = subpel_geometry(W, H);
	const size_t shmem = (size_t)g.blk_bytes * g.bpw;
	if (shmem > 64 * 1024) return set_error(HEVCASM_HIP_EINVAL, "block does not fit in 64 KiB of LDS");
	const uint32_t grid = ceil_div(n_blocks, g.bpw);
	hipStream_t st = as_stream(stream);
	if (metric == 0)
		hipLaunchKernelGGL((subpel_refine_kernel<0, 4>), dim3(grid), dim3(kThreads), shmem, st, src, stride_src, ref, stride_ref, src_off, ref_off, result, g, (uint32_t)n_blocks);
	else if (W % 8 == 0 && H % 8 == 0)
		hipLaunchKernelGGL((subpel_refine_kernel<1, 8>), dim3(grid), dim3(kThreads), shmem, st, src, stride_src, ref, stride_ref, src_off, ref_off, result, g, (uint32_t)n_blocks);
	else
		hipLaunchKernelGGL((subpel_refine_kernel<1, 4>), dim3(grid), dim3(kThreads), shmem, st, src, stride_src, ref, stride_ref, src_off, ref_off, result, g, (uint32_t)n_blocks);
	return launch_status("subpel_refine_kernel");
}
