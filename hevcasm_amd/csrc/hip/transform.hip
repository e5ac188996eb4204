// transform.hip -- HEVC inverse transform + add (optionally fused with dequantisation) and
// forward transform for gfx950.
//
// Replaces hevcasm_inverse_transform_add (reference residual_decode.h:54; C path
// residual_decode.c:69-413), the composition hevcasm_quantize_inverse -> inverse_transform_add
// (quantize.c:53-62 then the above), and hevcasm_transform (residual_decode.h:82; C path
// residual_decode.c:592-892) for whole batches of N x N blocks, N = 4 (DCT or DST-VII), 8, 16, 32.
//
// Mapping: N lanes own one block (lane t = column t in the vertical pass, row t in the
// horizontal pass), so a 256-thread workgroup carries 256 / N blocks.  Each lane runs one
// N-point 1-D transform entirely in registers using the even/odd ("partial butterfly")
// factorisation with compile-time coefficients: 32-bit accumulation, no intermediate rounding,
// so results equal the plain matrix product bit for bit.  The two passes exchange data through an
// LDS tile per block (the transpose), padded by 4 halfwords per row so that both the column walk
// (consecutive lanes, consecutive halfwords) and the row walk (lanes 2(N+4) bytes apart, 8-byte
// reads) are bank-conflict free.  Rounding, int16 clipping between passes (inverse) and
// truncation without clipping (forward) follow the reference exactly.
// Algorithmic bytes per block: 4 N^2 (2 N^2 coefficients + N^2 prediction + N^2 output).

#include "common.hpp"
#include "dct_matrix.hpp"
#include "tu_runs.hpp"

namespace hevc {

// ---- coefficients ------------------------------------------------------------------------

// entry (k, i) of the N-point matrix = row k * 32 / N of the 32-point one
template <int N> __device__ __forceinline__ constexpr int dct(int k, int i) { return kDct.m[k * (32 / N)][i]; }

// ---- 1-D transforms in registers ----------------------------------------------------------

// ---- inverse 1-D transform on packed int16 pairs --------------------------------------------
//
// The even/odd factorisation consumes its inputs in a fixed pair order: the odd rows of the
// N-point stage two by two, then the odd rows of the N/2-point stage (rows 2, 6, 10, ...), and so
// on down to the pair (0, N/2).  Both LDS tiles store the data already paired in that order, so a
// lane's 1-D transform is N/2 dword reads followed by v_dot2_i32_i16 against compile-time
// coefficient pairs: two multiply-accumulates per instruction, exact 32-bit sums.
typedef short s16x2 __attribute__((ext_vector_type(2)));

template <int N> struct PairOrder
{
	int first[N / 2], second[N / 2];   // pair p = rows (first[p], second[p])
	int pair_of[N], half_of[N];        // row k sits in pair pair_of[k], half half_of[k] (0 = low 16 bits)
};

template <int N> constexpr PairOrder<N> make_pair_order()
{
	PairOrder<N> o = {};
	int idx = 0;
	for (int step = 1; step < N / 2; step *= 2)
		for (int r = step; r < N; r += 4 * step)
		{
			o.first[idx] = r; o.second[idx] = r + 2 * step; ++idx;
		}
	o.first[idx] = 0; o.second[idx] = N / 2;
	for (int p = 0; p < N / 2; ++p)
	{
		o.pair_of[o.first[p]] = p; o.half_of[o.first[p]] = 0;
		o.pair_of[o.second[p]] = p; o.half_of[o.second[p]] = 1;
	}
	return o;
}

// byte offset of row k inside a pair-ordered row: 4 * pair + 2 * half
struct PairBytes { unsigned char at[4][32]; };
constexpr PairBytes make_pair_bytes()
{
	PairBytes t = {};
	constexpr PairOrder<4> o4 = make_pair_order<4>();
	constexpr PairOrder<8> o8 = make_pair_order<8>();
	constexpr PairOrder<16> o16 = make_pair_order<16>();
	constexpr PairOrder<32> o32 = make_pair_order<32>();
	for (int k = 0; k < 4; ++k) t.at[0][k] = (unsigned char)(4 * o4.pair_of[k] + 2 * o4.half_of[k]);
	for (int k = 0; k < 8; ++k) t.at[1][k] = (unsigned char)(4 * o8.pair_of[k] + 2 * o8.half_of[k]);
	for (int k = 0; k < 16; ++k) t.at[2][k] = (unsigned char)(4 * o16.pair_of[k] + 2 * o16.half_of[k]);
	for (int k = 0; k < 32; ++k) t.at[3][k] = (unsigned char)(4 * o32.pair_of[k] + 2 * o32.half_of[k]);
	return t;
}
__constant__ PairBytes kPairBytes = make_pair_bytes();

__device__ __forceinline__ int dot2(uint32_t pair, int c_lo, int c_hi, int acc)
{
	const uint32_t c = ((uint32_t)c_lo & 0xffffu) | ((uint32_t)c_hi << 16);
	return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, pair), __builtin_bit_cast(s16x2, c), acc, false);
}

// The FIRST product of a sum.  Left to the compiler, every sum starts as v_mov acc, 0 (or bias) + v_dot2c_i32_i16 acc, literal, pair: the VOP2
// form accumulates in place and takes its coefficient pair as a literal, so it needs its accumulator preset -- 74 v_mov in the 32x32 kernel, a tenth of
// its vector instructions, in a kernel whose VALU is 73-84 % busy at large launches (profiles/r04_pred_transform_counters.txt).  The VOP3P form takes
// the addend as an operand (inline 0, or the bias held in a register) and the coefficient pair from an SGPR (an s_mov on the scalar unit): no v_mov.
#ifndef HEVCASM_DOT2_FIRST_PLAIN
__device__ __forceinline__ int dot2_first(uint32_t pair, int c_lo, int c_hi)
{
	int r;
	const uint32_t c = ((uint32_t)c_lo & 0xffffu) | ((uint32_t)c_hi << 16);
	asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(pair), "s"(c));
	return r;
}
__device__ __forceinline__ int dot2_first(uint32_t pair, int c_lo, int c_hi, int bias /* in a VGPR */)
{
	int r;
	const uint32_t c = ((uint32_t)c_lo & 0xffffu) | ((uint32_t)c_hi << 16);
	asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(pair), "s"(c), "v"(bias));
	return r;
}
#else
__device__ __forceinline__ int dot2_first(uint32_t pair, int c_lo, int c_hi) { return dot2(pair, c_lo, c_hi, 0); }
__device__ __forceinline__ int dot2_first(uint32_t pair, int c_lo, int c_hi, int bias) { return dot2(pair, c_lo, c_hi, bias); }
#endif

// n-point stage of the N-point inverse transform; its inputs are every (N / n)-th row, found in
// pairs p[base ...].  out[i] = bias + sum over those rows.
template <int N, int n> struct InversePacked
{
	static __device__ __forceinline__ void run(const uint32_t (&p)[N / 2], int base, int bias, int (&out)[n])
	{
		int e[n / 2];
		InversePacked<N, n / 2>::run(p, base + n / 4, bias, e);
#pragma unroll
		for (int k = 0; k < n / 2; ++k)
		{
			int o = dot2_first(p[base], dct<n>(1, k), dct<n>(3, k));
#pragma unroll
			for (int q = 1; q < n / 4; ++q) o = dot2(p[base + q], dct<n>(4 * q + 1, k), dct<n>(4 * q + 3, k), o);
			out[k] = e[k] + o;
			out[n - 1 - k] = e[k] - o;
		}
	}
};
template <int N> struct InversePacked<N, 2>
{
	static __device__ __forceinline__ void run(const uint32_t (&p)[N / 2], int base, int bias, int (&out)[2])
	{
		out[0] = dot2_first(p[base], 64, 64, bias);
		out[1] = dot2_first(p[base], 64, -64, bias);
	}
};

// 4-point inverse DST-VII on the same pair order as the 4-point DCT: p[0] = (s1, s3), p[1] = (s0, s2)
__device__ __forceinline__ void dst4_inverse_packed(const uint32_t (&p)[2], int bias, int (&o)[4])
{
	o[0] = dot2(p[1], 29, 84, dot2_first(p[0], 74, 55, bias));
	o[1] = dot2(p[1], 55, -29, dot2_first(p[0], 74, -84, bias));
	o[2] = dot2(p[1], 74, -74, dot2_first(p[0], 0, 74, bias));
	o[3] = dot2(p[1], 84, 55, dot2_first(p[0], -74, -29, bias));
}

template <int N, bool DST> __device__ __forceinline__ void inverse_packed(const uint32_t (&p)[N / 2], int bias, int (&out)[N])
{
	if constexpr (DST) dst4_inverse_packed(p, bias, out); else InversePacked<N, N>::run(p, 0, bias, out);
}

// out[k] = sum_n dct<N>(k, n) * in[n]     (forward)
template <int N> struct Forward1D
{
	static __device__ __forceinline__ void run(const int (&in)[N], int (&out)[N])
	{
		int s[N / 2], d[N / 2], sub[N / 2];
#pragma unroll
		for (int n = 0; n < N / 2; ++n)
		{
			s[n] = in[n] + in[N - 1 - n];
			d[n] = in[n] - in[N - 1 - n];
		}
		Forward1D<N / 2>::run(s, sub);
#pragma unroll
		for (int k = 0; k < N / 2; ++k) out[2 * k] = sub[k];
#pragma unroll
		for (int k = 1; k < N; k += 2)
		{
			int a = 0;
#pragma unroll
			for (int n = 0; n < N / 2; ++n) a += dct<N>(k, n) * d[n];
			out[k] = a;
		}
	}
};
template <> struct Forward1D<2>
{
	static __device__ __forceinline__ void run(const int (&in)[2], int (&out)[2])
	{
		out[0] = 64 * (in[0] + in[1]);
		out[1] = 64 * (in[0] - in[1]);
	}
};

// 4-point DST-VII, basis rows {29,55,74,84} {74,74,0,-74} {84,-29,-74,55} {55,-84,74,-29}
__device__ __forceinline__ void dst4_forward(const int (&s)[4], int (&o)[4])
{
	o[0] = 29 * s[0] + 55 * s[1] + 74 * s[2] + 84 * s[3];
	o[1] = 74 * (s[0] + s[1] - s[3]);
	o[2] = 84 * s[0] - 29 * s[1] - 74 * s[2] + 55 * s[3];
	o[3] = 55 * s[0] - 84 * s[1] + 74 * s[2] - 29 * s[3];
}

template <int N, bool DST> __device__ __forceinline__ void forward_1d(const int (&in)[N], int (&out)[N])
{
	if constexpr (DST) dst4_forward(in, out); else Forward1D<N>::run(in, out);
}

// ---- LDS tile helpers -----------------------------------------------------------------------

template <int N> struct Tile
{
	static constexpr int S = N + 4;             // row stride in halfwords
	static constexpr int HALFWORDS = N * S;
};

// lane-private row <-> registers, 8 bytes at a time (rows start 8-byte aligned: S is a multiple of 4)
template <int N> __device__ __forceinline__ void row_to_regs(const int16_t *row, int (&v)[N])
{
#pragma unroll
	for (int q = 0; q < N / 4; ++q)
	{
		const u32x2 d = *reinterpret_cast<const u32x2 *>(row + 4 * q);
		v[4 * q + 0] = (int)(short)(d.x & 0xffff);
		v[4 * q + 1] = (int)d.x >> 16;
		v[4 * q + 2] = (int)(short)(d.y & 0xffff);
		v[4 * q + 3] = (int)d.y >> 16;
	}
}
template <int N> __device__ __forceinline__ void regs_to_row(int16_t *row, const int (&v)[N])
{
#pragma unroll
	for (int q = 0; q < N / 4; ++q)
	{
		u32x2 d;
		d.x = ((uint32_t)v[4 * q + 0] & 0xffffu) | ((uint32_t)v[4 * q + 1] << 16);
		d.y = ((uint32_t)v[4 * q + 2] & 0xffffu) | ((uint32_t)v[4 * q + 3] << 16);
		*reinterpret_cast<u32x2 *>(row + 4 * q) = d;
	}
}

// 2N contiguous bytes of global memory (no alignment promise) <-> N ints
template <int N, bool STREAM = false> __device__ __forceinline__ void global_row_load(const int16_t *p, int (&v)[N])
{
	constexpr int CB = N >= 8 ? 16 : 8;
#pragma unroll
	for (int q = 0; q < 2 * N / CB; ++q)
	{
		Chunk<CB> c;
		if (STREAM) c.load_stream(reinterpret_cast<const uint8_t *>(p) + q * CB); else c.load(reinterpret_cast<const uint8_t *>(p) + q * CB);
#pragma unroll
		for (int k = 0; k < CB / 4; ++k)
		{
			v[q * (CB / 2) + 2 * k] = (int)(short)(c.d[k] & 0xffff);
			v[q * (CB / 2) + 2 * k + 1] = (int)c.d[k] >> 16;
		}
	}
}
// the same as N / 2 raw dwords (int16 pairs as they lie in memory)
template <int N, bool STREAM = false> __device__ __forceinline__ void global_row_load_raw(const int16_t *p, uint32_t (&d)[N / 2])
{
	constexpr int CB = N >= 8 ? 16 : 8;
#pragma unroll
	for (int q = 0; q < 2 * N / CB; ++q)
	{
		Chunk<CB> c;
		if (STREAM) c.load_stream(reinterpret_cast<const uint8_t *>(p) + q * CB); else c.load(reinterpret_cast<const uint8_t *>(p) + q * CB);
#pragma unroll
		for (int k = 0; k < CB / 4; ++k) d[q * (CB / 4) + k] = c.d[k];
	}
}
template <int N, bool STREAM = false> __device__ __forceinline__ void global_row_store(int16_t *p, const int (&v)[N])
{
	constexpr int CB = N >= 8 ? 16 : 8;
#pragma unroll
	for (int q = 0; q < 2 * N / CB; ++q)
	{
		Chunk<CB> c;
#pragma unroll
		for (int k = 0; k < CB / 4; ++k)
			c.d[k] = ((uint32_t)v[q * (CB / 2) + 2 * k] & 0xffffu) | ((uint32_t)v[q * (CB / 2) + 2 * k + 1] << 16);
		if (STREAM) c.store_stream(reinterpret_cast<uint8_t *>(p) + q * CB); else c.store(reinterpret_cast<uint8_t *>(p) + q * CB);
	}
}

// ---- inverse transform + add ---------------------------------------------------------------

// LDS per block: tile P1 holds the coefficient block as [pair of rows][column] (one dword = the two
// rows of a pair at that column; row stride N + 1 dwords, so the column walk of the first pass and
// the 16-bit scatter that fills it are conflict free), tile P2 holds the first-pass result as
// [row][pair of columns] (row stride N/2 (+4) dwords: 16-byte aligned, conflict-free b128 reads).
template <int N> struct InverseTiles
{
	static constexpr int S1 = N + 1;
	static constexpr int S2 = N / 2 + (N >= 16 ? 4 : 0);
	static constexpr int P1_DWORDS = (N / 2) * S1;
	static constexpr int P2_DWORDS = N * S2;
	// P2 overlays P1: every lane has read its P1 column into registers (and a barrier has passed)
	// before anyone writes P2, which halves the LDS footprint and doubles the resident workgroups.
	static constexpr int DWORDS = ((P1_DWORDS > P2_DWORDS ? P1_DWORDS : P2_DWORDS) + 3) & ~3;
};

template <int N, bool DST, bool DEQUANT>
__global__ __launch_bounds__(kThreads) void inverse_transform_add_kernel(
	uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *pred, ptrdiff_t stride_pred,   // dst may alias pred (in-place reconstruction): no __restrict__
	const uint32_t *__restrict__ dst_off, const uint32_t *__restrict__ pred_off,
	const int16_t *__restrict__ coeffs, int scale, int shift, uint32_t n_blocks)
{
#define HEVC_WG xcd_contiguous(blockIdx.x, gridDim.x)
#include "transform_inverse_tile.inc"
#undef HEVC_WG
}

// ---- 4x4 / 8x8 inverse transform + add, one lane per block ------------------------------------
//
// A 4x4 block is 32 bytes of coefficients and an 8x8 block 128: small enough for one lane to hold
// and transform entirely in registers, with no LDS tile and no barrier.  Lane = block, so
// consecutive lanes read consecutive coefficient blocks, and for raster-ordered blocks
// neighbouring lanes read / write one contiguous run of each prediction / destination row.
// Pairs for the column pass are picked out of the row-major dwords with v_perm_b32; the first
// pass's int16 clip and the re-pairing for the row pass are one v_cvt_pk_i16_i32 (saturating).
// Columns are processed two at a time in pair order so that only 2N first-pass sums are live.
template <int N, bool DST, bool DEQUANT>
__global__ __launch_bounds__(kThreads) void inverse_lane_kernel(
	uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *pred, ptrdiff_t stride_pred,   // dst may alias pred (in-place reconstruction): no __restrict__
	const uint32_t *__restrict__ dst_off, const uint32_t *__restrict__ pred_off,
	const int16_t *__restrict__ coeffs, int scale, int shift, uint32_t n_blocks)
{
#define HEVC_WG xcd_contiguous(blockIdx.x, gridDim.x)
#include "transform_inverse_lane.inc"
#undef HEVC_WG
}

// ---- forward transform ----------------------------------------------------------------------

template <int N, bool DST>
__global__ __launch_bounds__(kThreads) void forward_transform_kernel(
	int16_t *__restrict__ coeffs, const int16_t *__restrict__ src, ptrdiff_t src_stride,
	const uint32_t *__restrict__ src_off, uint32_t n_blocks)
{
#define HEVC_WG xcd_contiguous(blockIdx.x, gridDim.x)
#include "transform_forward_tile.inc"
#undef HEVC_WG
}

// 4x4 forward transform, one lane per block: as inverse_lane_kernel, everything in registers.
// The first stage's truncation to int16 (residual_decode.c: the intermediate array is int16) is a
// sign-extension of the low half; 32-bit sums are exact for any int16 input.
template <bool DST>
__global__ __launch_bounds__(kThreads) void forward4_lane_kernel(
	int16_t *__restrict__ coeffs, const int16_t *__restrict__ src, ptrdiff_t src_stride,
	const uint32_t *__restrict__ src_off, uint32_t n_blocks)
{
#define HEVC_WG xcd_contiguous(blockIdx.x, gridDim.x)
#include "transform_forward4_lane.inc"
#undef HEVC_WG
}

// ---- 32x32 forward transform on the matrix cores ---------------------------------------------
//
// A 32-point transform of 32 rows IS a 32x32x32 integer matrix product, and with packed butterflies
// ruled out (the even/odd sums need 17 bits) the VALU form above is issue-bound at 0.38 of the HBM
// peak.  v_mfma_i32_32x32x32_i8 takes int8 operands; the coefficient matrix fits (|c| <= 90) and an
// int16 operand is split exactly into bytes, x = 256 * hi + (lo - 128) + 128 with hi signed and
// lo - 128 signed, so   X C' = 256 (Hi C') + (Lo' C') + 128 * rowsum(C)   in exact 32-bit arithmetic
// (|Hi C'| <= 32 * 128 * 90, far below 2^31 / 256).  One wave per block, no LDS:
//   stage 1  T = X C'   A operand = the residual rows as they lie in memory (lane = row r, half h:
//            16 consecutive int16 = two 16-byte loads), B operand = C', a per-lane constant;
//            rounding constant and the 128 * rowsum term ride in the accumulator;
//   stage 2  Out = C A' where A' = trunc16(T >> 4).  T's accumulator tile has its column (k) on the
//            lane and its 16 rows in registers, which is exactly a B operand that sums over the row
//            index: no lane movement.  The k order inside the instruction is then the accumulator's
//            row order (j&3) + 8 (j>>2) + 4 h, and the constant A operand is laid out to match.
// Both operands of an MFMA only have to AGREE on which k a byte carries (tools/ubench/mfma_i8_layout.hip),
// so stage 1 simply calls byte j of lane half h "k = 16 h + j".
typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef int v16i32 __attribute__((ext_vector_type(16)));

struct Forward32Tables
{
	uint32_t b1[64][4];    // stage 1 B operand: lane (k' = l & 31, h = l >> 5), byte j = C[k'][16 h + j]
	uint32_t a2[64][4];    // stage 2 A operand: lane (m  = l & 31, h),          byte j = C[m][(j & 3) + 8 (j >> 2) + 4 h]
	int rowsum[32];        // sum_n C[k][n]
};
constexpr Forward32Tables make_forward32_tables()
{
	Forward32Tables t = {};
	for (int l = 0; l < 64; ++l)
	{
		const int r = l & 31, h = l >> 5;
		for (int j = 0; j < 16; ++j)
		{
			const uint32_t c1 = (uint32_t)(unsigned char)kDct.m[r][16 * h + j];
			const uint32_t c2 = (uint32_t)(unsigned char)kDct.m[r][(j & 3) + 8 * (j >> 2) + 4 * h];
			t.b1[l][j / 4] |= c1 << (8 * (j % 4));
			t.a2[l][j / 4] |= c2 << (8 * (j % 4));
		}
	}
	for (int k = 0; k < 32; ++k)
		for (int n = 0; n < 32; ++n) t.rowsum[k] += kDct.m[k][n];
	return t;
}
__constant__ Forward32Tables kFwd32 = make_forward32_tables();

// 8 int16 in 4 dwords -> their low bytes (minus 128, as int8) and high bytes, 2 dwords each
__device__ __forceinline__ void split_bytes(const uint32_t (&w)[4], uint32_t (&lo)[2], uint32_t (&hi)[2])
{
	lo[0] = __builtin_amdgcn_perm(w[1], w[0], 0x06040200) ^ 0x80808080u;
	lo[1] = __builtin_amdgcn_perm(w[3], w[2], 0x06040200) ^ 0x80808080u;
	hi[0] = __builtin_amdgcn_perm(w[1], w[0], 0x07050301);
	hi[1] = __builtin_amdgcn_perm(w[3], w[2], 0x07050301);
}

constexpr int kFwd32BlocksPerWave = 2;   // 1, 2, 4, 8 measured: within noise of each other (0.068-0.075 ms per 65 536 blocks)

__global__ __launch_bounds__(kThreads) void forward32_mfma_kernel(
	int16_t *__restrict__ coeffs, const int16_t *__restrict__ src, ptrdiff_t src_stride,
	const uint32_t *__restrict__ src_off, uint32_t n_blocks)
{
#define HEVC_WG xcd_contiguous(blockIdx.x, gridDim.x)
#include "transform_forward32_mfma.inc"
#undef HEVC_WG
}

template <int N, bool DST, bool DEQUANT>
static int launch_inverse(uint8_t *dst, ptrdiff_t sd, const uint8_t *pred, ptrdiff_t sp, const uint32_t *dof, const uint32_t *pof,
		const int16_t *c, int scale, int shift, size_t n, hipStream_t st)
{
	const uint32_t grid = ceil_div(n, kThreads / N);
	hipLaunchKernelGGL((inverse_transform_add_kernel<N, DST, DEQUANT>), dim3(grid), dim3(kThreads), 0, st,
			dst, sd, pred, sp, dof, pof, c, scale, shift, (uint32_t)n);
	return launch_status("inverse_transform_add_kernel");
}

template <bool DEQUANT>
static int inverse_entry(uint8_t *dst, ptrdiff_t sd, const uint8_t *pred, ptrdiff_t sp, const uint32_t *dof, const uint32_t *pof,
		const int16_t *c, int scale, int shift, int trType, int log2n, size_t n, void *stream)
{
	if (int e = check_device()) return e;
	if (log2n < 2 || log2n > 5 || (trType && log2n != 2)) return set_error(HEVCASM_HIP_EINVAL, "transform size/type not supported by the reference");
	if (DEQUANT && (shift < 1 || shift > 31)) return set_error(HEVCASM_HIP_EINVAL, "shift out of range");
	if (DEQUANT && (scale < -32768 || scale > 32767)) return set_error(HEVCASM_HIP_EINVAL, "scale must fit int16 (the reference's SIMD path multiplies with pmaddwd, quantize_a.asm:53-109)");
	if (n == 0) return 0;
	if (n > 0xffffffffull) return set_error(HEVCASM_HIP_EINVAL, "n_blocks exceeds 2^32-1");
	if (!dst || !pred || !dof || !pof || !c) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	hipStream_t st = as_stream(stream);
	if (log2n <= 3)
	{
		const uint32_t grid = ceil_div(n, kThreads);
		if (trType) hipLaunchKernelGGL((inverse_lane_kernel<4, true, DEQUANT>), dim3(grid), dim3(kThreads), 0, st, dst, sd, pred, sp, dof, pof, c, scale, shift, (uint32_t)n);
		else if (log2n == 2) hipLaunchKernelGGL((inverse_lane_kernel<4, false, DEQUANT>), dim3(grid), dim3(kThreads), 0, st, dst, sd, pred, sp, dof, pof, c, scale, shift, (uint32_t)n);
		else hipLaunchKernelGGL((inverse_lane_kernel<8, false, DEQUANT>), dim3(grid), dim3(kThreads), 0, st, dst, sd, pred, sp, dof, pof, c, scale, shift, (uint32_t)n);
		return launch_status("inverse_lane_kernel");
	}
	switch (log2n)
	{
	case 4: return launch_inverse<16, false, DEQUANT>(dst, sd, pred, sp, dof, pof, c, scale, shift, n, st);
	default: return launch_inverse<32, false, DEQUANT>(dst, sd, pred, sp, dof, pof, c, scale, shift, n, st);
	}
}

template <int N, bool DST>
static int launch_forward(int16_t *coeffs, const int16_t *src, ptrdiff_t ss, const uint32_t *sof, size_t n, hipStream_t st)
{
	const uint32_t grid = ceil_div(n, kThreads / N);
	hipLaunchKernelGGL((forward_transform_kernel<N, DST>), dim3(grid), dim3(kThreads), 0, st, coeffs, src, ss, sof, (uint32_t)n);
	return launch_status("forward_transform_kernel");
}

// ---- one launch for a frame's transform units: runs of equally sized TUs (round 6; tu_runs.hpp) ----------------------------------------------------
//
// The workgroup bodies above once more, as functions of (batch, workgroup index inside the batch); the one-size kernels keep their own copies of the same text
// (an inlined function is optimised before it meets its kernel's launch bounds and came out scheduled differently: the measured kernels stay as they were).
template <int N, bool DST, bool DEQUANT>
__device__ __forceinline__ void inverse_tile_workgroup(uint32_t *lds,
	uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *pred, ptrdiff_t stride_pred, const uint32_t *__restrict__ dst_off, const uint32_t *__restrict__ pred_off,
	const int16_t *__restrict__ coeffs, int scale, int shift, uint32_t n_blocks, const uint32_t wg_in_batch)
{
#define HEVC_WG wg_in_batch
#define HEVC_INV_TILES_DECL uint32_t *tiles = lds      // one buffer for the runs kernel's sizes in turn, not a static array per size
#define HEVC_INV_ROW_IN(blk, t, v, raw) \
	if (DEQ_OPSEL) global_row_load_raw<N, N == 16>(coeffs + (size_t)blk * (N * N) + t * N, raw); \
	else global_row_load<N, N == 16>(coeffs + (size_t)blk * (N * N) + t * N, v)
#include "transform_inverse_tile.inc"
#undef HEVC_INV_TILES_DECL
#undef HEVC_INV_ROW_IN
#undef HEVC_WG
}
template <int N, bool DST, bool DEQUANT>
__device__ __forceinline__ void inverse_lane_workgroup(
	uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *pred, ptrdiff_t stride_pred, const uint32_t *__restrict__ dst_off, const uint32_t *__restrict__ pred_off,
	const int16_t *__restrict__ coeffs, int scale, int shift, uint32_t n_blocks, const uint32_t wg_in_batch)
{
#define HEVC_WG wg_in_batch
#include "transform_inverse_lane.inc"
#undef HEVC_WG
}
template <int N, bool DST>
__device__ __forceinline__ void forward_tile_workgroup(uint32_t *lds,
	int16_t *__restrict__ coeffs, const int16_t *__restrict__ src, ptrdiff_t src_stride, const uint32_t *__restrict__ src_off, uint32_t n_blocks, const uint32_t wg_in_batch)
{
#define HEVC_WG wg_in_batch
#define HEVC_FWD_TILES_DECL int16_t *tiles = reinterpret_cast<int16_t *>(lds)
#define HEVC_FWD_ROW_IN(off1, t1, v) global_row_load<N>(src + off1 + (ptrdiff_t)t1 * src_stride, v)
#define HEVC_FWD_ROW_OUT(blk, t, v, live) if (live) global_row_store<N, true>(coeffs + (size_t)blk * (N * N) + t * N, v)
#include "transform_forward_tile.inc"
#undef HEVC_FWD_TILES_DECL
#undef HEVC_FWD_ROW_IN
#undef HEVC_FWD_ROW_OUT
#undef HEVC_WG
}
template <bool DST>
__device__ __forceinline__ void forward4_lane_workgroup(
	int16_t *__restrict__ coeffs, const int16_t *__restrict__ src, ptrdiff_t src_stride, const uint32_t *__restrict__ src_off, uint32_t n_blocks, const uint32_t wg_in_batch)
{
#define HEVC_WG wg_in_batch
#include "transform_forward4_lane.inc"
#undef HEVC_WG
}
__device__ __forceinline__ void forward32_mfma_workgroup(
	int16_t *__restrict__ coeffs, const int16_t *__restrict__ src, ptrdiff_t src_stride, const uint32_t *__restrict__ src_off, uint32_t n_blocks, const uint32_t wg_in_batch)
{
#define HEVC_WG wg_in_batch
#include "transform_forward32_mfma.inc"
#undef HEVC_WG
}

// variant = 0: 4x4 DST, 1: 4x4 DCT, 2: 8x8, 3: 16x16, 4: 32x32 (both directions); blocks per workgroup of each body
constexpr uint32_t kInverseRunBlocksPerWg[5] = { kThreads, kThreads, kThreads, kThreads / 16, kThreads / 32 };
constexpr uint32_t kForwardRunBlocksPerWg[5] = { kThreads, kThreads, kThreads / 8, kThreads / 16, (kThreads / kWave) * kFwd32BlocksPerWave };

struct InverseRunsArgs
{
	uint8_t *dst;
	const uint8_t *pred;
	ptrdiff_t stride_dst, stride_pred;
	const uint32_t *dst_off, *pred_off;
	const int16_t *coeffs;
	TuRunTable table;
};
struct ForwardRunsArgs
{
	int16_t *coeffs;
	const int16_t *src;
	ptrdiff_t src_stride;
	const uint32_t *src_off;
	TuRunTable table;
};

template <bool DEQUANT>
__global__ __launch_bounds__(kThreads) void inverse_runs_kernel(const InverseRunsArgs a)
{
	__shared__ __attribute__((aligned(16))) uint32_t lds[(kThreads / 32) * InverseTiles<32>::DWORDS];      // 20 KiB: the larger of the two tile sizes
	static_assert((kThreads / 32) * InverseTiles<32>::DWORDS >= (kThreads / 16) * InverseTiles<16>::DWORDS, "inverse runs LDS");
	uint32_t wg;
	const TuRunDev r = find_tu_run(a.table, wg);
	const uint32_t *dof = a.dst_off + r.first, *pof = a.pred_off + r.first;
	const int16_t *c = a.coeffs + r.coeff_first;
	switch (r.variant)
	{
	case 0: inverse_lane_workgroup<4, true, DEQUANT>(a.dst, a.stride_dst, a.pred, a.stride_pred, dof, pof, c, r.scale, r.shift, r.count, wg); break;
	case 1: inverse_lane_workgroup<4, false, DEQUANT>(a.dst, a.stride_dst, a.pred, a.stride_pred, dof, pof, c, r.scale, r.shift, r.count, wg); break;
	case 2: inverse_lane_workgroup<8, false, DEQUANT>(a.dst, a.stride_dst, a.pred, a.stride_pred, dof, pof, c, r.scale, r.shift, r.count, wg); break;
	case 3: inverse_tile_workgroup<16, false, DEQUANT>(lds, a.dst, a.stride_dst, a.pred, a.stride_pred, dof, pof, c, r.scale, r.shift, r.count, wg); break;
	case 4: inverse_tile_workgroup<32, false, DEQUANT>(lds, a.dst, a.stride_dst, a.pred, a.stride_pred, dof, pof, c, r.scale, r.shift, r.count, wg); break;
	default: break;
	}
}

// CLASSES as for fused_tu_runs_kernel below: bit 0 = the sizes below 32x32, bit 1 = 32x32 on the matrix cores (+ 64 accumulation registers); a large call is
// two launches so that the small sizes keep their occupancy, a frame-sized one a single launch of the combined kernel.
template <int CLASSES>
__global__ __launch_bounds__(kThreads) void forward_runs_kernel(const ForwardRunsArgs a)
{
	__shared__ __attribute__((aligned(16))) uint32_t lds[(kThreads / 16) * Tile<16>::HALFWORDS / 2];      // 10 KiB: 16x16 tiles; the 8x8 tiles fit
	static_assert((kThreads / 16) * Tile<16>::HALFWORDS >= (kThreads / 8) * Tile<8>::HALFWORDS, "forward runs LDS");
	uint32_t wg;
	const TuRunDev r = find_tu_run(a.table, wg);
	int16_t *c = a.coeffs + r.coeff_first;
	const uint32_t *sof = a.src_off + r.first;
	if constexpr (CLASSES & 1)
	{
		switch (r.variant)
		{
		case 0: forward4_lane_workgroup<true>(c, a.src, a.src_stride, sof, r.count, wg); break;
		case 1: forward4_lane_workgroup<false>(c, a.src, a.src_stride, sof, r.count, wg); break;
		case 2: forward_tile_workgroup<8, false>(lds, c, a.src, a.src_stride, sof, r.count, wg); break;
		case 3: forward_tile_workgroup<16, false>(lds, c, a.src, a.src_stride, sof, r.count, wg); break;
		default: break;
		}
	}
	if constexpr (CLASSES & 2)
	{
		if (r.variant == 4) forward32_mfma_workgroup(c, a.src, a.src_stride, sof, r.count, wg);
	}
}

// a run's size and type -> variant; the checks of the one-size entry points
static int tu_variant(const hevcasm_hip_tu_run &r, uint32_t &variant, int n_planes = 1)
{
	if (r.log2TrafoSize < 2 || r.log2TrafoSize > 5 || r.trType > 1 || (r.trType && r.log2TrafoSize != 2) || (int)r.plane >= n_planes)
		return set_error(HEVCASM_HIP_EINVAL, "run: transform size/type not supported by the reference (log2TrafoSize 2..5, trType 1 only at 2), or plane beyond the call's planes");
	variant = r.log2TrafoSize == 2 ? (r.trType ? 0u : 1u) : (uint32_t)r.log2TrafoSize - 1u;
	return 0;
}

template <bool DEQUANT>
static int inverse_runs_entry(uint8_t *dst, ptrdiff_t sd, const uint8_t *pred, ptrdiff_t sp, const uint32_t *dof, const uint32_t *pof,
		const int16_t *c, const hevcasm_hip_tu_run *runs, int n_runs, void *stream)
{
	if (int e = check_device()) return e;
	if (n_runs < 0) return set_error(HEVCASM_HIP_EINVAL, "negative run count");
	if (n_runs == 0) return 0;
	if (!runs || !dst || !pred || !dof || !pof || !c) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	std::vector<TuRunTable> tables;
	std::vector<uint32_t> grids;
	if (int e = build_tu_tables(runs, n_runs, [](const hevcasm_hip_tu_run &r, uint32_t &variant, uint32_t &bpwg) {
			if (int e = tu_variant(r, variant)) return e;
			if (DEQUANT && (r.shift < 1 || r.shift > 31)) return set_error(HEVCASM_HIP_EINVAL, "run: shift out of range");
			if (DEQUANT && (r.scale < -32768 || r.scale > 32767)) return set_error(HEVCASM_HIP_EINVAL, "run: scale must fit int16 (the reference's SIMD path multiplies with pmaddwd, quantize_a.asm:53-109)");
			bpwg = kInverseRunBlocksPerWg[variant];
			return 0; }, tables, grids)) return e;
	for (size_t i = 0; i < tables.size(); ++i)
	{
		InverseRunsArgs a = { dst, pred, sd, sp, dof, pof, c, tables[i] };
		hipLaunchKernelGGL((inverse_runs_kernel<DEQUANT>), dim3(grids[i]), dim3(kThreads), 0, as_stream(stream), a);
		if (int e = launch_status("inverse_runs_kernel")) return e;
	}
	return 0;
}

// ---- the residual loop of a frame's TUs in ONE launch: source - prediction -> transform -> quantise -> dequantise -> inverse + add (round 6) ---------------
//
// An encoder reconstructs every TU it codes: hevcasm_transform, hevcasm_quantize, hevcasm_quantize_inverse, hevcasm_inverse_transform_add (residual_decode.h:82,
// quantize.h:78, :57, residual_decode.h:54) back to back.  As three launches the residual, the coefficients and the levels each cross HBM twice (16 bytes per
// pixel all told); here they never leave the workgroup: 5 bytes per pixel (source, prediction, reconstruction in, 2-byte levels out) + the descriptors.  What
// makes it cheap to write: in the forward bodies a lane ends holding ROW t of its block's coefficients -- exactly what the inverse bodies start from (4x4: a
// lane holds the whole block in both) -- so the fused body is the forward body fed from source - prediction, the quantiser and dequantiser on that row in
// registers (levels stored, the flag OR-ed over the block's lanes), and the inverse body from behind its loads.  The bodies are the .inc files above through
// their hooks; 8x8 takes the tile form of the inverse (the lane form wants a lane per block, the forward tile has a lane per row); 32x32 keeps the forward
// transform on the matrix cores and turns its column-per-lane result into rows through LDS (fused_mfma32_workgroup).  One 20 KiB LDS buffer serves forward
// and inverse tiles of every size in turn.
constexpr int kFusedLdsDwords = 8 * InverseTiles<32>::DWORDS;      // the largest tenant: eight 32x32 blocks' inverse tiles
static_assert(kFusedLdsDwords * 4 >= (kThreads / 32) * Tile<32>::HALFWORDS * 2 && kFusedLdsDwords * 4 >= (kThreads / 16) * Tile<16>::HALFWORDS * 2 &&
		kFusedLdsDwords >= (kThreads / 16) * InverseTiles<16>::DWORDS && kFusedLdsDwords >= (kThreads / 8) * InverseTiles<8>::DWORDS, "fused LDS buffer too small");

// N source bytes minus N prediction bytes
template <int N> __device__ __forceinline__ void residual_row(const uint8_t *c, const uint8_t *p, int (&v)[N])
{
	constexpr int CB = N >= 16 ? 16 : N;
#pragma unroll
	for (int q = 0; q < N / CB; ++q)
	{
		Chunk<CB> a, b;
		a.load(c + q * CB);
		b.load(p + q * CB);
#pragma unroll
		for (int k = 0; k < CB; ++k) v[q * CB + k] = (int)((a.d[k / 4] >> (8 * (k % 4))) & 0xff) - (int)((b.d[k / 4] >> (8 * (k % 4))) & 0xff);
	}
}

// quantize.c:160-186: level = sign(v) * ((|v| * scale + round) >> shift), clipped to int16.  This kernel is bound by its vector instructions, so the same number
// in four of them instead of nine: with a = v * scale (exact in 32 bits), -floor((-a + round) / 2^shift) = ceil((a - round) / 2^shift) = floor((a + 2^shift - 1 - round)
// / 2^shift), and an arithmetic shift is that floor -- so a negative v only takes another rounding constant (neg_round = 2^shift - 1 - round); -32768 needs no
// special case (the reference's abs() is an int).  The clip cannot act: |v| <= 2^15, scale < 2^15, shift >= 16, round < 2^(shift - 1) (the entry point checks the
// last three) => |level| <= 2^14.  quantize.c:53-62 below as quant.hip has it; scale24 / dscale24 carry 24 significant bits.
__device__ __forceinline__ int fused_quantize(int v, int scale24, int round, int neg_round, int shift)
{
	return (v * scale24 + (v < 0 ? neg_round : round)) >> shift;
}
__device__ __forceinline__ int fused_dequantize(int level, int dscale24, int dround, int dshift) { return clip3(-32768, 32767, (level * dscale24 + dround) >> dshift); }

template <int N>
__device__ __forceinline__ void fused_tile_workgroup(uint32_t *lds,
	uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *cur, ptrdiff_t stride_cur, const uint8_t *pred, ptrdiff_t stride_pred,
	const uint32_t *__restrict__ dst_off, const uint32_t *__restrict__ cur_off, const uint32_t *__restrict__ pred_off,
	int16_t *__restrict__ levels, int *__restrict__ cbf, int qscale, int qshift, int qoffset, int dscale, int dshift, uint32_t n_blocks, const uint32_t wg_in_batch)
{
	constexpr bool DST = false;
	int row[N];
	uint32_t row_blk = 0, row_t = 0;
	bool row_live = false;
#define HEVC_WG wg_in_batch
	{
		const uint32_t *src_off = cur_off;
#define HEVC_FWD_TILES_DECL int16_t *tiles = reinterpret_cast<int16_t *>(lds)
#define HEVC_FWD_ROW_IN(off1, t1, v) residual_row<N>(cur + off1 + (ptrdiff_t)t1 * stride_cur, pred + pred_off[blk1] + (ptrdiff_t)t1 * stride_pred, v)
#define HEVC_FWD_ROW_OUT(blk, t, v, live) \
		{ \
			_Pragma("unroll") \
			for (int k = 0; k < N; ++k) row[k] = v[k]; \
			row_blk = blk; row_t = t; row_live = live; \
		}
#include "transform_forward_tile.inc"
#undef HEVC_FWD_TILES_DECL
#undef HEVC_FWD_ROW_IN
#undef HEVC_FWD_ROW_OUT
	}
	{
		const int qround = qoffset << (qshift - 16), qneg = (1 << qshift) - 1 - qround, dround = 1 << (dshift - 1);
		const int scale24 = (qscale << 8) >> 8, dscale24 = (dscale << 8) >> 8;
		int lev[N], any = 0;
#pragma unroll
		for (int k = 0; k < N; ++k)
		{
			lev[k] = fused_quantize(row[k], scale24, qround, qneg, qshift);
			any |= lev[k];
			row[k] = fused_dequantize(lev[k], dscale24, dround, dshift);
		}
		if (row_live) global_row_store<N, true>(levels + (size_t)row_blk * (N * N) + row_t * N, lev);
#pragma unroll
		for (int d = N / 2; d >= 1; d >>= 1) any |= __shfl_xor(any, d, kWave);      // the block's N lanes are neighbours inside one wave
		if (row_live && row_t == 0) cbf[row_blk] = any;
	}
	__syncthreads();      // the forward tiles have been read: their storage becomes the inverse tiles
	{
		constexpr bool DEQUANT = false;
		const int scale = 0, shift = 1;
#define HEVC_INV_TILES_DECL uint32_t *tiles = lds
#define HEVC_INV_ROW_IN(blk, t, v, raw) \
		{ \
			_Pragma("unroll") \
			for (int k = 0; k < N; ++k) v[k] = row[k]; \
		}
#include "transform_inverse_tile.inc"
#undef HEVC_INV_TILES_DECL
#undef HEVC_INV_ROW_IN
	}
#undef HEVC_WG
}

// 32x32: the forward transform on the matrix cores (forward32_mfma_kernel's body).  A wave owns two blocks in both halves: it transforms them one after the
// other, a lane holding 16 rows of ONE column of the result -- where the coefficients are quantised (the body's own paired store then writes the LEVELS), the
// flag is OR-ed over the wave, and the dequantised values go to an LDS tile [row][column]; from there lane t of each half-wave takes ROW t, which is what the
// inverse body starts from.  (The vector-unit form of the forward pass in this place made the fused launch 1.45x SLOWER than the three launches at 32x32.)
__device__ __forceinline__ void fused_mfma32_workgroup(uint32_t *lds,
	uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *cur, ptrdiff_t stride_cur, const uint8_t *pred, ptrdiff_t stride_pred,
	const uint32_t *__restrict__ dst_off, const uint32_t *__restrict__ cur_off, const uint32_t *__restrict__ pred_off,
	int16_t *__restrict__ levels, int *__restrict__ cbf, int qscale, int qshift, int qoffset, int dscale, int dshift, uint32_t n_blocks, const uint32_t wg_in_batch)
{
	constexpr int N = 32, SS = Tile<32>::S, TILE = Tile<32>::HALFWORDS;
	const int qround = qoffset << (qshift - 16), qneg = (1 << qshift) - 1 - qround, dround = 1 << (dshift - 1);
	const int scale24 = (qscale << 8) >> 8, dscale24 = (dscale << 8) >> 8;
	int16_t *stage = reinterpret_cast<int16_t *>(lds);
#define HEVC_WG wg_in_batch
	[&]()      // (the body leaves by `return` when its wave has no block: every wave must still reach the barrier below)
	{
		int16_t *coeffs = levels;                                              // what the body stores are levels
		const ptrdiff_t src_stride = stride_cur;                               // (only the hooks read memory)
		int any = 0;
#define HEVC_FWD32_LOAD(blk, x) \
		{ \
			(void)lane_off; \
			Chunk<16> c_, p_; \
			c_.load(cur + cur_off[blk] + (ptrdiff_t)r * stride_cur + 16 * h); \
			p_.load(pred + pred_off[blk] + (ptrdiff_t)r * stride_pred + 16 * h); \
			_Pragma("unroll") \
			for (int k_ = 0; k_ < 8; ++k_) \
			{ \
				const int e_ = 2 * k_, o_ = 2 * k_ + 1; \
				const int lo_ = (int)((c_.d[e_ / 4] >> (8 * (e_ % 4))) & 0xff) - (int)((p_.d[e_ / 4] >> (8 * (e_ % 4))) & 0xff); \
				const int hi_ = (int)((c_.d[o_ / 4] >> (8 * (o_ % 4))) & 0xff) - (int)((p_.d[o_ / 4] >> (8 * (o_ % 4))) & 0xff); \
				x[k_] = ((uint32_t)lo_ & 0xffffu) | ((uint32_t)hi_ << 16); \
			} \
		}
#define HEVC_FWD32_VALUES(blk, i, v0, v1) \
		{ \
			int16_t *tile_ = stage + ((threadIdx.x >> 6) * BPW + (blk - first)) * TILE + r; \
			v0 = fused_quantize((int)(int16_t)v0, scale24, qround, qneg, qshift); \
			v1 = fused_quantize((int)(int16_t)v1, scale24, qround, qneg, qshift); \
			any |= v0 | v1; \
			tile_[((i & 3) + 8 * (i >> 2) + 4 * h) * SS] = (int16_t)fused_dequantize(v0, dscale24, dround, dshift); \
			tile_[(((i + 1) & 3) + 8 * ((i + 1) >> 2) + 4 * h) * SS] = (int16_t)fused_dequantize(v1, dscale24, dround, dshift); \
		}
#define HEVC_FWD32_BLOCK_DONE(blk) \
		{ \
			_Pragma("unroll") \
			for (int d_ = 32; d_ >= 1; d_ >>= 1) any |= __shfl_xor(any, d_, kWave); \
			if (l == 0) cbf[blk] = any; \
			any = 0; \
		}
#include "transform_forward32_mfma.inc"
#undef HEVC_FWD32_LOAD
#undef HEVC_FWD32_VALUES
#undef HEVC_FWD32_BLOCK_DONE
	}();
	__syncthreads();      // the dequantised blocks are in the tiles
	int row[N];
	row_to_regs<N>(stage + (threadIdx.x / N) * TILE + (threadIdx.x % N) * SS, row);
	__syncthreads();      // ... and in registers: the storage becomes the inverse tiles
	{
		constexpr bool DST = false, DEQUANT = false;
		const int scale = 0, shift = 1;
#define HEVC_INV_TILES_DECL uint32_t *tiles = lds
#define HEVC_INV_ROW_IN(blk, t, v, raw) \
		{ \
			_Pragma("unroll") \
			for (int k = 0; k < N; ++k) v[k] = row[k]; \
		}
#include "transform_inverse_tile.inc"
#undef HEVC_INV_TILES_DECL
#undef HEVC_INV_ROW_IN
	}
#undef HEVC_WG
}

template <bool DST>
__device__ __forceinline__ void fused_lane4_workgroup(
	uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *cur, ptrdiff_t stride_cur, const uint8_t *pred, ptrdiff_t stride_pred,
	const uint32_t *__restrict__ dst_off, const uint32_t *__restrict__ cur_off, const uint32_t *__restrict__ pred_off,
	int16_t *__restrict__ levels, int *__restrict__ cbf, int qscale, int qshift, int qoffset, int dscale, int dshift, uint32_t n_blocks, const uint32_t wg_in_batch)
{
	const int qround = qoffset << (qshift - 16), qneg = (1 << qshift) - 1 - qround, dround = 1 << (dshift - 1);
	const int scale24 = (qscale << 8) >> 8, dscale24 = (dscale << 8) >> 8;
	uint32_t lev_pairs[8], deq_pairs[8];      // row k, columns (2j, 2j + 1) in dword 2k + j, low half first: the coefficient block as it lies in memory
	int any = 0;
#define HEVC_WG wg_in_batch
	{
		const int16_t *src = reinterpret_cast<const int16_t *>(cur);      // only the hooks read memory
		const uint32_t *src_off = cur_off;
#define HEVC_FWD4_ROW_IN(p, y, v) { (void)p; residual_row<4>(cur + cur_off[blk] + (ptrdiff_t)y * stride_cur, pred + pred_off[blk] + (ptrdiff_t)y * stride_pred, v); }
#define HEVC_FWD4_ROW_OUT(blk, k, coef) \
		{ \
			int l_[4], d_[4]; \
			_Pragma("unroll") \
			for (int x = 0; x < 4; ++x) \
			{ \
				l_[x] = fused_quantize(coef[x], scale24, qround, qneg, qshift); \
				any |= l_[x]; \
				d_[x] = fused_dequantize(l_[x], dscale24, dround, dshift); \
			} \
			lev_pairs[2 * k] = ((uint32_t)l_[0] & 0xffffu) | ((uint32_t)l_[1] << 16); lev_pairs[2 * k + 1] = ((uint32_t)l_[2] & 0xffffu) | ((uint32_t)l_[3] << 16); \
			deq_pairs[2 * k] = ((uint32_t)d_[0] & 0xffffu) | ((uint32_t)d_[1] << 16); deq_pairs[2 * k + 1] = ((uint32_t)d_[2] & 0xffffu) | ((uint32_t)d_[3] << 16); \
		}
#include "transform_forward4_lane.inc"
#undef HEVC_FWD4_ROW_IN
#undef HEVC_FWD4_ROW_OUT
	}
	{
		const uint32_t blk = wg_in_batch * kThreads + threadIdx.x;      // (the forward body has returned for lanes beyond the batch)
		Chunk<16> c0, c1;
#pragma unroll
		for (int i = 0; i < 4; ++i) { c0.d[i] = lev_pairs[i]; c1.d[i] = lev_pairs[4 + i]; }
		uint8_t *out = reinterpret_cast<uint8_t *>(levels + (size_t)blk * 16);
		c0.store(out);
		c1.store(out + 16);
		cbf[blk] = any;
	}
	{
		constexpr int N = 4;
		constexpr bool DEQUANT = false;
		const int scale = 0, shift = 1;
#define HEVC_INVLANE_IN(blk, r) \
		{ \
			_Pragma("unroll") \
			for (int i = 0; i < 8; ++i) r[i] = deq_pairs[i]; \
		}
#include "transform_inverse_lane.inc"
#undef HEVC_INVLANE_IN
	}
#undef HEVC_WG
}

constexpr uint32_t kFusedRunBlocksPerWg[5] = { kThreads, kThreads, kThreads / 8, kThreads / 16, kThreads / 32 };

constexpr int kMaxTuPlanes = 4;
struct FusedRunsArgs
{
	hevcasm_hip_tu_plane planes[kMaxTuPlanes];      // a run's plane: luma, Cb, Cr of a frame in one launch
	const uint32_t *dst_off, *cur_off, *pred_off;
	int16_t *levels;
	int *cbf;
	TuRunTable table;
};

// CLASSES: bit 0 = the sizes below 32x32 (76 registers), bit 1 = 32x32 (the matrix-core path: + 64 accumulation registers, three waves per SIMD instead of
// six).  A launch small enough to be latency takes both in one kernel; a large one is two launches, so that the small sizes keep their occupancy.
template <int CLASSES>
__global__ __launch_bounds__(kThreads) void fused_tu_runs_kernel(const FusedRunsArgs a)
{
	__shared__ __attribute__((aligned(16))) uint32_t lds[kFusedLdsDwords];
	uint32_t wg;
	const TuRunDev r = find_tu_run(a.table, wg);
	const uint32_t *dof = a.dst_off + r.first, *cof = a.cur_off + r.first, *pof = a.pred_off + r.first;
	int16_t *lv = a.levels + r.coeff_first;
	int *cbf = a.cbf + r.first;
	const hevcasm_hip_tu_plane P = a.planes[r.plane];
#define HEVC_FUSED_ARGS P.rec, P.stride_rec, P.src, P.stride_src, P.pred, P.stride_pred, dof, cof, pof, lv, cbf, r.scale, r.shift, r.offset, r.dscale, r.dshift, r.count, wg
	if constexpr (CLASSES & 1)
	{
		switch (r.variant)
		{
		case 0: fused_lane4_workgroup<true>(HEVC_FUSED_ARGS); break;
		case 1: fused_lane4_workgroup<false>(HEVC_FUSED_ARGS); break;
		case 2: fused_tile_workgroup<8>(lds, HEVC_FUSED_ARGS); break;
		case 3: fused_tile_workgroup<16>(lds, HEVC_FUSED_ARGS); break;
		default: break;
		}
	}
	if constexpr (CLASSES & 2)
	{
		if (r.variant == 4) fused_mfma32_workgroup(lds, HEVC_FUSED_ARGS);
	}
#undef HEVC_FUSED_ARGS
}

}  // namespace hevc

using namespace hevc;

extern "C" int hevcasm_hip_transform_quantize_reconstruct_planes(
	const hevcasm_hip_tu_plane *planes, int n_planes,
	const uint32_t *rec_off, const uint32_t *src_off, const uint32_t *pred_off, int16_t *levels, int *cbf,
	const hevcasm_hip_tu_run *runs_quantize, const hevcasm_hip_tu_run *runs_dequantize, int n_runs, void *stream)
{
	if (int e = check_device()) return e;
	if (n_runs < 0) return set_error(HEVCASM_HIP_EINVAL, "negative run count");
	if (n_planes < 1 || n_planes > kMaxTuPlanes) return set_error(HEVCASM_HIP_EINVAL, "1 to 4 planes per call");
	if (n_runs == 0) return 0;
	if (!planes || !runs_quantize || !runs_dequantize || !rec_off || !src_off || !pred_off || !levels || !cbf) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	for (int i = 0; i < n_planes; ++i)
		if (!planes[i].rec || !planes[i].src || !planes[i].pred) return set_error(HEVCASM_HIP_EINVAL, "null plane pointer");
	// validated first; then split by class when the call is large (see fused_tu_runs_kernel)
	std::vector<hevcasm_hip_tu_run> q[2], dq[2];
	uint64_t wgs_all = 0;
	for (int i = 0; i < n_runs; ++i)
	{
		const int big = runs_quantize[i].log2TrafoSize == 5;
		q[big].push_back(runs_quantize[i]);
		dq[big].push_back(runs_dequantize[i]);
		if (runs_quantize[i].log2TrafoSize >= 2 && runs_quantize[i].log2TrafoSize <= 5)
			wgs_all += ceil_div(runs_quantize[i].count, kFusedRunBlocksPerWg[runs_quantize[i].log2TrafoSize == 2 ? 0 : runs_quantize[i].log2TrafoSize - 1]);
	}
	const bool split = !q[0].empty() && !q[1].empty() && wgs_all > (uint64_t)device_cus() * 8;
	if (!split)
	{
		q[0].insert(q[0].end(), q[1].begin(), q[1].end());
		dq[0].insert(dq[0].end(), dq[1].begin(), dq[1].end());
		q[1].clear(); dq[1].clear();
	}
	for (int part = -1; part < 2; ++part)      // part -1: every run is checked before anything is launched
	{
		if (part >= 0 && q[part].empty()) continue;
		const hevcasm_hip_tu_run *rq = part < 0 ? runs_quantize : q[part].data(), *rdq = part < 0 ? runs_dequantize : dq[part].data();
		const int n_part = part < 0 ? n_runs : (int)q[part].size();
		bool any_small = false, any_big = false;
		for (int i = 0; i < n_part; ++i) { if (rq[i].count == 0) continue; (rq[i].log2TrafoSize == 5 ? any_big : any_small) = true; }
		std::vector<TuRunTable> tables;
		std::vector<uint32_t> grids;
		if (int e = build_tu_tables(rq, n_part, [=](const hevcasm_hip_tu_run &r, uint32_t &variant, uint32_t &bpwg) {
			if (int e = tu_variant(r, variant, n_planes)) return e;
			const hevcasm_hip_tu_run &d = rdq[&r - rq];
			if (d.first != r.first || d.count != r.count || d.coeff_first != r.coeff_first || d.log2TrafoSize != r.log2TrafoSize || d.trType != r.trType || d.plane != r.plane)
				return set_error(HEVCASM_HIP_EINVAL, "run: the dequantiser's run must describe the same units as the quantiser's");
			// the domains of hevcasm_hip_quantize_batch (quantize.c:162-168) and of hevcasm_hip_dequant_inverse_transform_add_batch
			if (r.scale < 0 || r.scale >= 0x8000 || r.offset < 0 || r.offset >= 0x8000 || r.shift < 16 || r.shift > 27)
				return set_error(HEVCASM_HIP_EINVAL, "run: quantiser scale/offset/shift outside the reference's domain (0 <= scale, offset < 0x8000; 16 <= shift <= 27)");
			if (d.shift < 1 || d.shift > 31 || d.scale < -32768 || d.scale > 32767) return set_error(HEVCASM_HIP_EINVAL, "run: dequantiser shift out of range or scale beyond int16");
			bpwg = kFusedRunBlocksPerWg[variant];
			return 0; }, tables, grids, rdq)) return e;
		if (part < 0) continue;
		for (size_t i = 0; i < tables.size(); ++i)
		{
			FusedRunsArgs a = {};
			for (int k = 0; k < n_planes; ++k) a.planes[k] = planes[k];
			a.dst_off = rec_off; a.cur_off = src_off; a.pred_off = pred_off; a.levels = levels; a.cbf = cbf; a.table = tables[i];
			if (any_small && any_big) hipLaunchKernelGGL(fused_tu_runs_kernel<3>, dim3(grids[i]), dim3(kThreads), 0, as_stream(stream), a);
			else if (any_big) hipLaunchKernelGGL(fused_tu_runs_kernel<2>, dim3(grids[i]), dim3(kThreads), 0, as_stream(stream), a);
			else hipLaunchKernelGGL(fused_tu_runs_kernel<1>, dim3(grids[i]), dim3(kThreads), 0, as_stream(stream), a);
			if (int e = launch_status("fused_tu_runs_kernel")) return e;
		}
	}
	return 0;
}

extern "C" int hevcasm_hip_transform_quantize_reconstruct_runs(
	uint8_t *rec, ptrdiff_t stride_rec, const uint8_t *src, ptrdiff_t stride_src, const uint8_t *pred, ptrdiff_t stride_pred,
	const uint32_t *rec_off, const uint32_t *src_off, const uint32_t *pred_off, int16_t *levels, int *cbf,
	const hevcasm_hip_tu_run *runs_quantize, const hevcasm_hip_tu_run *runs_dequantize, int n_runs, void *stream)
{
	if (n_runs > 0 && (!rec || !src || !pred)) { if (int e = check_device()) return e; return set_error(HEVCASM_HIP_EINVAL, "null pointer"); }
	const hevcasm_hip_tu_plane plane = { rec, src, pred, stride_rec, stride_src, stride_pred };
	return hevcasm_hip_transform_quantize_reconstruct_planes(&plane, 1, rec_off, src_off, pred_off, levels, cbf, runs_quantize, runs_dequantize, n_runs, stream);
}

extern "C" int hevcasm_hip_inverse_transform_add_runs(
	uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *pred, ptrdiff_t stride_pred,
	const uint32_t *dst_off, const uint32_t *pred_off, const int16_t *coeffs,
	const hevcasm_hip_tu_run *runs, int n_runs, void *stream)
{
	return inverse_runs_entry<false>(dst, stride_dst, pred, stride_pred, dst_off, pred_off, coeffs, runs, n_runs, stream);
}

extern "C" int hevcasm_hip_dequant_inverse_transform_add_runs(
	uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *pred, ptrdiff_t stride_pred,
	const uint32_t *dst_off, const uint32_t *pred_off, const int16_t *levels,
	const hevcasm_hip_tu_run *runs, int n_runs, void *stream)
{
	return inverse_runs_entry<true>(dst, stride_dst, pred, stride_pred, dst_off, pred_off, levels, runs, n_runs, stream);
}

extern "C" int hevcasm_hip_transform_runs(
	int16_t *coeffs, const int16_t *src, ptrdiff_t src_stride, const uint32_t *src_off,
	const hevcasm_hip_tu_run *runs, int n_runs, void *stream)
{
	if (int e = check_device()) return e;
	if (n_runs < 0) return set_error(HEVCASM_HIP_EINVAL, "negative run count");
	if (n_runs == 0) return 0;
	if (!runs || !coeffs || !src || !src_off) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	std::vector<hevcasm_hip_tu_run> part[2];
	uint64_t wgs_all = 0;
	for (int i = 0; i < n_runs; ++i)
	{
		part[runs[i].log2TrafoSize == 5].push_back(runs[i]);
		if (runs[i].log2TrafoSize >= 2 && runs[i].log2TrafoSize <= 5) wgs_all += ceil_div(runs[i].count, kForwardRunBlocksPerWg[runs[i].log2TrafoSize == 2 ? 0 : runs[i].log2TrafoSize - 1]);
	}
	const bool split = !part[0].empty() && !part[1].empty() && wgs_all > (uint64_t)device_cus() * 8;
	if (!split) { part[0].insert(part[0].end(), part[1].begin(), part[1].end()); part[1].clear(); }
	for (int p = -1; p < 2; ++p)      // p = -1: every run is checked before anything is launched
	{
		if (p >= 0 && part[p].empty()) continue;
		const hevcasm_hip_tu_run *rp = p < 0 ? runs : part[p].data();
		const int n_part = p < 0 ? n_runs : (int)part[p].size();
		bool any_small = false, any_big = false;
		for (int i = 0; i < n_part; ++i) { if (rp[i].count == 0) continue; (rp[i].log2TrafoSize == 5 ? any_big : any_small) = true; }
		std::vector<TuRunTable> tables;
		std::vector<uint32_t> grids;
		if (int e = build_tu_tables(rp, n_part, [](const hevcasm_hip_tu_run &r, uint32_t &variant, uint32_t &bpwg) {
				if (int e = tu_variant(r, variant)) return e;
				bpwg = kForwardRunBlocksPerWg[variant];
				return 0; }, tables, grids)) return e;
		if (p < 0) continue;
		for (size_t i = 0; i < tables.size(); ++i)
		{
			ForwardRunsArgs a = { coeffs, src, src_stride, src_off, tables[i] };
			if (any_small && any_big) hipLaunchKernelGGL(forward_runs_kernel<3>, dim3(grids[i]), dim3(kThreads), 0, as_stream(stream), a);
			else if (any_big) hipLaunchKernelGGL(forward_runs_kernel<2>, dim3(grids[i]), dim3(kThreads), 0, as_stream(stream), a);
			else hipLaunchKernelGGL(forward_runs_kernel<1>, dim3(grids[i]), dim3(kThreads), 0, as_stream(stream), a);
			if (int e = launch_status("forward_runs_kernel")) return e;
		}
	}
	return 0;
}

extern "C" int hevcasm_hip_inverse_transform_add_batch(
	uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *pred, ptrdiff_t stride_pred,
	const uint32_t *dst_off, const uint32_t *pred_off, const int16_t *coeffs,
	int trType, int log2TrafoSize, size_t n_blocks, void *stream)
{
	return inverse_entry<false>(dst, stride_dst, pred, stride_pred, dst_off, pred_off, coeffs, 0, 0, trType, log2TrafoSize, n_blocks, stream);
}

extern "C" int hevcasm_hip_dequant_inverse_transform_add_batch(
	uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *pred, ptrdiff_t stride_pred,
	const uint32_t *dst_off, const uint32_t *pred_off, const int16_t *levels,
	int scale, int shift,
	int trType, int log2TrafoSize, size_t n_blocks, void *stream)
{
	return inverse_entry<true>(dst, stride_dst, pred, stride_pred, dst_off, pred_off, levels, scale, shift, trType, log2TrafoSize, n_blocks, stream);
}

extern "C" int hevcasm_hip_transform_batch(
	int16_t *coeffs, const int16_t *src, ptrdiff_t src_stride, const uint32_t *src_off,
	int trType, int log2TrafoSize, size_t n_blocks, void *stream)
{
	if (int e = check_device()) return e;
	if (log2TrafoSize < 2 || log2TrafoSize > 5 || (trType && log2TrafoSize != 2)) return set_error(HEVCASM_HIP_EINVAL, "transform size/type not supported by the reference");
	if (n_blocks == 0) return 0;
	if (n_blocks > 0xffffffffull) return set_error(HEVCASM_HIP_EINVAL, "n_blocks exceeds 2^32-1");
	if (!coeffs || !src || !src_off) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	hipStream_t st = as_stream(stream);
	if (log2TrafoSize == 2)
	{
		const uint32_t grid = ceil_div(n_blocks, kThreads);
		if (trType) hipLaunchKernelGGL((forward4_lane_kernel<true>), dim3(grid), dim3(kThreads), 0, st, coeffs, src, src_stride, src_off, (uint32_t)n_blocks);
		else hipLaunchKernelGGL((forward4_lane_kernel<false>), dim3(grid), dim3(kThreads), 0, st, coeffs, src, src_stride, src_off, (uint32_t)n_blocks);
		return launch_status("forward4_lane_kernel");
	}
	switch (log2TrafoSize)
	{
	case 3: return launch_forward<8, false>(coeffs, src, src_stride, src_off, n_blocks, st);
	case 4: return launch_forward<16, false>(coeffs, src, src_stride, src_off, n_blocks, st);
	default:
		hipLaunchKernelGGL(forward32_mfma_kernel, dim3(ceil_div(n_blocks, (size_t)(kThreads / kWave) * kFwd32BlocksPerWave)), dim3(kThreads), 0, st, coeffs, src, src_stride, src_off, (uint32_t)n_blocks);
		return launch_status("forward32_mfma_kernel");
	}
}
