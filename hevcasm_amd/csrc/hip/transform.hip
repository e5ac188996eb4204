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
	typedef InverseTiles<N> T;
	constexpr int BPWG = kThreads / N;
	constexpr int LOG2 = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
	__shared__ __attribute__((aligned(16))) uint32_t tiles[BPWG * T::DWORDS];

	const uint32_t wg = xcd_contiguous(blockIdx.x, gridDim.x);
	const uint32_t local = threadIdx.x / N, t = threadIdx.x % N;
	uint32_t blk = wg * BPWG + local;
	const bool live = blk < n_blocks;
	if (!live) blk = n_blocks - 1;
	uint32_t *p1 = tiles + local * T::DWORDS;
	uint32_t *p2 = p1;
	const uint32_t my_pair_bytes = kPairBytes.at[LOG2 - 2][t];   // where row / column t sits in a pair-ordered row

	int v[N], w[N];
	uint32_t p[N / 2];

	// The add stage's lane deal (below: consecutive lanes hold the SAME row of consecutive blocks) is known now, so its
	// offsets and prediction bytes are fetched now, behind the coefficients: where the add stage uses them they would be
	// two more memory round trips at the very end of every workgroup.
	const uint32_t local2 = threadIdx.x % BPWG, t2 = threadIdx.x / BPWG;
	uint32_t blk2 = wg * BPWG + local2;
	const bool live2 = blk2 < n_blocks;
	if (!live2) blk2 = n_blocks - 1;
	// (round 4: the workgroup's offsets by scalar loads + v_cndmask, as the SAD kernels fetch their descriptors, measured -5 ... +6 % by kernel and
	// size, nothing on balance: profiles/r04_scalar_offsets_elsewhere_ab.txt -- these kernels run at the rate of their access mix)
	const uint32_t po = pred_off[blk2], dof = dst_off[blk2];
	const uint8_t *pp = pred + po + t2 * stride_pred;
	uint8_t *d = dst + dof + t2 * stride_dst;
	constexpr int CB = N >= 16 ? 16 : N;
	Chunk<CB> pr[N / CB];

	// row t of the coefficient block -> column-major pairs in P1 (dequantised on the way when fused)
	// coefficients are read once: as streaming loads at 16x16 (-4 %); at 32x32 two lanes share each line and it costs 8 %
	// (the op_sel form of the fused dequantisation only at 32x32: at 16x16 it measured +2 % / +6 % at 2^26 / 2^28 coefficients, twice, same box:
	// profiles/r04_transform_valu_trim_ab.txt -- that kernel is not VALU-bound and liked the older instruction mix better)
	constexpr bool DEQ_OPSEL = DEQUANT && N == 32;
	uint32_t raw[N / 2];
	if (DEQ_OPSEL) global_row_load_raw<N, N == 16>(coeffs + (size_t)blk * (N * N) + t * N, raw);
	else global_row_load<N, N == 16>(coeffs + (size_t)blk * (N * N) + t * N, v);
#pragma unroll
	for (int q = 0; q < N / CB; ++q) pr[q].load(pp + q * CB);
	__builtin_amdgcn_sched_barrier(0);
	{
		// element (row t, column x) -> P1[pair_of(t)][x], half_of(t)
		int16_t *row = reinterpret_cast<int16_t *>(reinterpret_cast<uint8_t *>(p1) + (my_pair_bytes >> 2) * (T::S1 * 4) + (my_pair_bytes & 2));
		if (DEQ_OPSEL)
		{
			// quantize.c:53-62: clip16((level * scale + round) >> shift).  scale < 2^15 (checked by the entry point) and the levels are int16, so the
			// product is v_mad_i32_i16, which picks either half of the loaded dword by op_sel (no unpacking of the pair) and adds the rounding
			// constant; the clip to int16 is the saturating pack.  2.5 vector instructions per level (was 3.5: r04_fused_dct32_order.txt has the
			// before / after; the fused kernel is VALU-bound at large launches, its extra instructions are what it costs over the plain one)
			typedef short s16x2v __attribute__((ext_vector_type(2)));
			const int round = 1 << (shift - 1);
#pragma unroll
			for (int k = 0; k < N / 2; ++k)
			{
				int lo, hi;
				asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(lo) : "v"(raw[k]), "s"(scale), "v"(round));
				asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(hi) : "v"(raw[k]), "s"(scale), "v"(round));
				const uint32_t two = __builtin_bit_cast(uint32_t, (s16x2v)__builtin_amdgcn_cvt_pk_i16(lo >> shift, hi >> shift));   // v_cvt_pk_i16_i32 saturates
				row[4 * k] = (int16_t)two;                 // ds_write_b16
				row[4 * k + 2] = (int16_t)(two >> 16);     // ds_write_b16_d16_hi
			}
		}
		else if (DEQUANT)
		{
			// telling the compiler that scale fits 24 bits turns a quarter-rate v_mul_lo_u32 per coefficient into a full-rate v_mad_i32_i24
			typedef short s16x2v __attribute__((ext_vector_type(2)));
			const int round = 1 << (shift - 1), scale24 = (scale << 8) >> 8;
#pragma unroll
			for (int x = 0; x < N; x += 2)
			{
				const int a = (v[x] * scale24 + round) >> shift, b = (v[x + 1] * scale24 + round) >> shift;
				const uint32_t two = __builtin_bit_cast(uint32_t, (s16x2v)__builtin_amdgcn_cvt_pk_i16(a, b));   // v_cvt_pk_i16_i32 saturates: the clip to int16
				row[2 * x] = (int16_t)two;                 // ds_write_b16
				row[2 * x + 2] = (int16_t)(two >> 16);     // ds_write_b16_d16_hi
			}
		}
		else
		{
#pragma unroll
			for (int x = 0; x < N; ++x) row[2 * x] = (int16_t)v[x];
		}
	}
	__syncthreads();

	// first stage: column t, shift 7, clip to int16; result row y, column t -> P2[y][pair_of(t)], half_of(t)
#pragma unroll
	for (int q = 0; q < N / 2; ++q) p[q] = p1[q * T::S1 + t];
	__syncthreads();   // P1 fully consumed: its storage becomes P2
	inverse_packed<N, DST>(p, 64, w);
	{
		int16_t *col = reinterpret_cast<int16_t *>(reinterpret_cast<uint8_t *>(p2) + my_pair_bytes);
		// the clip to int16 is the saturating pack of two rows' values; its halves go out as ds_write_b16 / ds_write_b16_d16_hi
		typedef short s16x2w __attribute__((ext_vector_type(2)));
#pragma unroll
		for (int y = 0; y < N; y += 2)
		{
			const uint32_t two = __builtin_bit_cast(uint32_t, (s16x2w)__builtin_amdgcn_cvt_pk_i16(w[y] >> 7, w[y + 1] >> 7));
			col[y * (T::S2 * 2)] = (int16_t)two;
			col[(y + 1) * (T::S2 * 2)] = (int16_t)(two >> 16);
		}
	}
	__syncthreads();

	// Second stage + add.  The lanes are re-dealt so that consecutive lanes hold the SAME row of
	// consecutive blocks: with blocks in raster order their prediction / destination rows are then
	// contiguous in memory (one 64..256-byte run per row instead of N-byte pieces a stride apart).
	{
		const uint32_t *row = tiles + local2 * T::DWORDS + t2 * T::S2;
		if constexpr (N >= 8)
		{
#pragma unroll
			for (int q = 0; q < N / 8; ++q)
			{
				const u32x4 d = *reinterpret_cast<const u32x4 *>(row + 4 * q);
				p[4 * q + 0] = d.x; p[4 * q + 1] = d.y; p[4 * q + 2] = d.z; p[4 * q + 3] = d.w;
			}
		}
		else
		{
			const u32x2 d = *reinterpret_cast<const u32x2 *>(row);
			p[0] = d.x; p[1] = d.y;
		}
	}
	inverse_packed<N, DST>(p, 2048, w);

	// Add to the prediction row and saturate to 8 bits.  The reference clips the second-stage result
	// to int16 and narrows pred + residual to int16 before the 8-bit clip (residual_decode.c:350-368).
	// Neither can ever act here: the second stage's inputs are int16 and every column of the
	// transform matrix has sum |M| <= 64 + 31 * 90 = 2854, so |(sum + 2048) >> 12| <= 22834 and
	// pred + residual stays inside int16.  Hence clip8(pred + ((w) >> 12)) == sat_u8((w + (pred << 12)) >> 12)
	// (w already carries the rounding constant; adding a multiple of 4096 commutes with the shift),
	// which is one v_lshl_add per pixel and one v_ashr_pk_u8_i32 per two pixels.
	if (live2)
	{
#pragma unroll
		for (int q = 0; q < N / CB; ++q)
		{
			Chunk<CB> c = pr[q];
#pragma unroll
			for (int k = 0; k < CB / 4; ++k)
			{
				int tt[4];
#pragma unroll
				for (int b = 0; b < 4; ++b) tt[b] = w[q * CB + 4 * k + b] + (int)(((c.d[k] >> (8 * b)) & 0xff) << 12);
				const uint32_t lo = __builtin_amdgcn_ashr_pk_u8_i32(tt[0], tt[1], 12);
				const uint32_t hi = __builtin_amdgcn_ashr_pk_u8_i32(tt[2], tt[3], 12);
				c.d[k] = __builtin_amdgcn_perm(hi, lo, 0x05040100);
			}
			c.store(d + q * CB);
		}
	}
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
	typedef short s16x2v __attribute__((ext_vector_type(2)));
	constexpr PairOrder<N> order = make_pair_order<N>();
	constexpr int RD = N / 2;   // dwords per coefficient row
	const uint32_t blk = xcd_contiguous(blockIdx.x, gridDim.x) * kThreads + threadIdx.x;
	if (blk >= n_blocks) return;

	// r[k * RD + j] = (c[k][2j], c[k][2j + 1])
	uint32_t r[N * RD];
	{
		const uint8_t *p = reinterpret_cast<const uint8_t *>(coeffs + (size_t)blk * (N * N));
#pragma unroll
		for (int i = 0; i < N * RD / 4; ++i)
		{
			Chunk<16> c;
			c.load(p + 16 * i);
#pragma unroll
			for (int j = 0; j < 4; ++j) r[4 * i + j] = c.d[j];
		}
	}
	// The prediction rows (and the offsets they hang off) are fetched now, behind the coefficients, not where the row pass
	// needs them: left to the compiler they were issued after the column pass -- three memory round trips in a row per wave.
	const uint8_t *pp = pred + pred_off[blk];
	uint8_t *d = dst + dst_off[blk];
	Chunk<N> pr[N];
#pragma unroll
	for (int y = 0; y < N; ++y) pr[y].load(pp + y * stride_pred);
	__builtin_amdgcn_sched_barrier(0);
	if (DEQUANT)
	{
		const int round = 1 << (shift - 1), scale24 = (scale << 8) >> 8;   // 24-bit operands: v_mad_i32_i24, not v_mul_lo_u32
#pragma unroll
		for (int i = 0; i < N * RD; ++i)
		{
			const int lo = ((int)(short)(r[i] & 0xffff) * scale24 + round) >> shift;   // quantize.c:53-62;
			const int hi = (((int)r[i] >> 16) * scale24 + round) >> shift;             // v_cvt_pk_i16_i32 is the clip
			r[i] = __builtin_bit_cast(uint32_t, (s16x2v)__builtin_amdgcn_cvt_pk_i16(lo, hi));
		}
	}

	// column pass.  q[p * N + y] = (e[y][first[p]], e[y][second[p]]), e = clip16(sum >> 7): row y of the
	// intermediate block, already in the pair order the row pass consumes.
	uint32_t q[RD * N];
#pragma unroll
	for (int cp = 0; cp < RD; ++cp)
	{
		int w[2][N];
#pragma unroll
		for (int side = 0; side < 2; ++side)
		{
			const int x = side ? order.second[cp] : order.first[cp];
			const uint32_t sel = (x & 1) ? 0x07060302u : 0x05040100u;    // high / low halves of (second, first)
			uint32_t p[RD];
#pragma unroll
			for (int k = 0; k < RD; ++k)
				p[k] = __builtin_amdgcn_perm(r[order.second[k] * RD + (x >> 1)], r[order.first[k] * RD + (x >> 1)], sel);
			inverse_packed<N, DST>(p, 64, w[side]);
		}
#pragma unroll
		for (int y = 0; y < N; ++y)
			q[cp * N + y] = __builtin_bit_cast(uint32_t, (s16x2v)__builtin_amdgcn_cvt_pk_i16(w[0][y] >> 7, w[1][y] >> 7));
	}

	// row pass + prediction + saturate
#pragma unroll
	for (int y = 0; y < N; ++y)
	{
		uint32_t p[RD];
#pragma unroll
		for (int k = 0; k < RD; ++k) p[k] = q[k * N + y];
		int w2[N];
		inverse_packed<N, DST>(p, 2048, w2);
		Chunk<N> c = pr[y];
#pragma unroll
		for (int k = 0; k < N / 4; ++k)
		{
			int tt[4];
#pragma unroll
			for (int b = 0; b < 4; ++b) tt[b] = w2[4 * k + b] + (int)(((c.d[k] >> (8 * b)) & 0xff) << 12);   // exact: see the N x N kernel
			const uint32_t lo = __builtin_amdgcn_ashr_pk_u8_i32(tt[0], tt[1], 12);
			const uint32_t hi = __builtin_amdgcn_ashr_pk_u8_i32(tt[2], tt[3], 12);
			c.d[k] = __builtin_amdgcn_perm(hi, lo, 0x05040100);
		}
		c.store(d + y * stride_dst);
	}
}

// ---- forward transform ----------------------------------------------------------------------

template <int N, bool DST>
__global__ __launch_bounds__(kThreads) void forward_transform_kernel(
	int16_t *__restrict__ coeffs, const int16_t *__restrict__ src, ptrdiff_t src_stride,
	const uint32_t *__restrict__ src_off, uint32_t n_blocks)
{
	constexpr int BPWG = kThreads / N;
	constexpr int S = Tile<N>::S;
	constexpr int LOG2 = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
	constexpr int S1 = LOG2 - 1, S2 = LOG2 + 6;  // residual_decode.c:855-892
	__shared__ __attribute__((aligned(16))) int16_t tiles[BPWG * Tile<N>::HALFWORDS];

	const uint32_t wg = xcd_contiguous(blockIdx.x, gridDim.x);
	const uint32_t local = threadIdx.x / N, t = threadIdx.x % N;
	uint32_t blk = wg * BPWG + local;
	const bool live = blk < n_blocks;
	if (!live) blk = n_blocks - 1;
	int16_t *tile = tiles + local * Tile<N>::HALFWORDS;

	int v[N], w[N];

	// first stage: one residual row per lane; result (truncated to int16) stored transposed.  Lanes
	// are dealt row-major across blocks here (same row of consecutive blocks in consecutive lanes)
	// so that raster-ordered blocks read contiguous runs of the residual plane.
	{
		// (measured: the re-deal pays at N = 4 and 32; at 8 and 16 the transposed LDS store it implies conflicts more than the loads gain)
		constexpr bool REDEAL = N == 4 || N == 32;
		const uint32_t local1 = REDEAL ? threadIdx.x % BPWG : local, t1 = REDEAL ? threadIdx.x / BPWG : t;
		uint32_t blk1 = wg * BPWG + local1;
		if (blk1 >= n_blocks) blk1 = n_blocks - 1;
		int16_t *tile1 = tiles + local1 * Tile<N>::HALFWORDS;
		const uint32_t off1 = src_off[blk1];
		global_row_load<N>(src + off1 + (ptrdiff_t)t1 * src_stride, v);
		forward_1d<N, DST>(v, w);
#pragma unroll
		for (int k = 0; k < N; ++k) tile1[k * S + t1] = (int16_t)((w[k] + (1 << (S1 - 1))) >> S1);
	}
	__syncthreads();

	// second stage: lane t holds horizontal frequency t across all rows; output column t
	row_to_regs<N>(tile + t * S, v);
	forward_1d<N, DST>(v, w);
	__syncthreads();
#pragma unroll
	for (int k = 0; k < N; ++k) tile[k * S + t] = (int16_t)((w[k] + (1 << (S2 - 1))) >> S2);
	__syncthreads();

	// coalesced write-out: lane t stores output row t (2N contiguous bytes)
	row_to_regs<N>(tile + t * S, v);
	// written once, a lane a whole row: streaming stores (8x8 -8 %, 16x16 -9 %; the 4x4 lane kernel, whose four 8-byte
	// stores per lane share lines across instructions, loses 17 % with them)
	if (live) global_row_store<N, true>(coeffs + (size_t)blk * (N * N) + t * N, v);
}

// 4x4 forward transform, one lane per block: as inverse_lane_kernel, everything in registers.
// The first stage's truncation to int16 (residual_decode.c: the intermediate array is int16) is a
// sign-extension of the low half; 32-bit sums are exact for any int16 input.
template <bool DST>
__global__ __launch_bounds__(kThreads) void forward4_lane_kernel(
	int16_t *__restrict__ coeffs, const int16_t *__restrict__ src, ptrdiff_t src_stride,
	const uint32_t *__restrict__ src_off, uint32_t n_blocks)
{
	constexpr int N = 4, S1 = 1, S2 = 8;   // residual_decode.c:855-892
	const uint32_t blk = xcd_contiguous(blockIdx.x, gridDim.x) * kThreads + threadIdx.x;
	if (blk >= n_blocks) return;
	const int16_t *p = src + src_off[blk];

	int a[N][N];   // a[k][y]: horizontal frequency k of residual row y
#pragma unroll
	for (int y = 0; y < N; ++y)
	{
		int v[N], w[N];
		global_row_load<N>(p + (ptrdiff_t)y * src_stride, v);
		forward_1d<N, DST>(v, w);
#pragma unroll
		for (int k = 0; k < N; ++k) a[k][y] = (int)(int16_t)((w[k] + (1 << (S1 - 1))) >> S1);
	}
	int o[N][N];   // o[k][t]: vertical frequency k, horizontal frequency t
#pragma unroll
	for (int t = 0; t < N; ++t)
	{
		int w[N];
		forward_1d<N, DST>(a[t], w);
#pragma unroll
		for (int k = 0; k < N; ++k) o[k][t] = (w[k] + (1 << (S2 - 1))) >> S2;
	}
#pragma unroll
	for (int k = 0; k < N; ++k) global_row_store<N>(coeffs + (size_t)blk * (N * N) + k * N, o[k]);
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
	constexpr int S1 = 4, S2 = 11;   // residual_decode.c:855-892 for N = 32, 8-bit video
	constexpr int BPW = kFwd32BlocksPerWave;
	// a wave walks BPW consecutive blocks; the next block's residuals are in flight while this one is
	// transformed, and the constant operands are fetched once
	const uint32_t wave = __builtin_amdgcn_readfirstlane(xcd_contiguous(blockIdx.x, gridDim.x) * (kThreads / kWave) + (threadIdx.x >> 6));
	const uint32_t first = wave * BPW;
	if (first >= n_blocks) return;
	const int count = (int)min((uint32_t)BPW, n_blocks - first);
	const int l = threadIdx.x & 63, r = l & 31, h = l >> 5;

	const v4i32 b1 = { (int)kFwd32.b1[l][0], (int)kFwd32.b1[l][1], (int)kFwd32.b1[l][2], (int)kFwd32.b1[l][3] };
	const v4i32 a2 = { (int)kFwd32.a2[l][0], (int)kFwd32.a2[l][1], (int)kFwd32.a2[l][2], (int)kFwd32.a2[l][3] };
	const int bias1 = 128 * kFwd32.rowsum[r] + (1 << (S1 - 1));
	int bias2[16];
#pragma unroll
	for (int i = 0; i < 16; ++i) bias2[i] = 128 * kFwd32.rowsum[(i & 3) + 8 * (i >> 2) + 4 * h] + (1 << (S2 - 1));

	// this lane's 16 residuals of a block: row r, columns 16 h .. 16 h + 15
	const ptrdiff_t lane_off = (ptrdiff_t)r * src_stride + 16 * h;
	auto load_block = [&](uint32_t blk, uint32_t (&x)[8])
	{
		const uint8_t *p = reinterpret_cast<const uint8_t *>(src + src_off[blk] + lane_off);
		Chunk<16> c0, c1;
		c0.load(p);
		c1.load(p + 16);
#pragma unroll
		for (int i = 0; i < 4; ++i) { x[i] = c0.d[i]; x[4 + i] = c1.d[i]; }
	};
	uint32_t x[8], xn[8];
	load_block(first, x);

	for (int b = 0; b < count; ++b)
	{
		const uint32_t blk = first + b;
		if (b + 1 < count) load_block(blk + 1, xn);

		v4i32 a_lo, a_hi;
		{
			uint32_t lo[2], hi[2];
			split_bytes({ x[0], x[1], x[2], x[3] }, lo, hi);
			a_lo.x = lo[0]; a_lo.y = lo[1]; a_hi.x = hi[0]; a_hi.y = hi[1];
			split_bytes({ x[4], x[5], x[6], x[7] }, lo, hi);
			a_lo.z = lo[0]; a_lo.w = lo[1]; a_hi.z = hi[0]; a_hi.w = hi[1];
		}

		// stage 1: this lane ends up with T[row][k = r] for its 16 rows
		v16i32 t_lo, t_hi = { 0 };
#pragma unroll
		for (int i = 0; i < 16; ++i) t_lo[i] = bias1;
		t_lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_lo, b1, t_lo, 0, 0, 0);
		t_hi = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_hi, b1, t_hi, 0, 0, 0);

		// intermediate, truncated to int16 as the reference's int16 array does; paired for the byte split
		uint32_t pair[8];
#pragma unroll
		for (int i = 0; i < 16; i += 2)
		{
			const int v0 = ((t_hi[i] << 8) + t_lo[i]) >> S1, v1 = ((t_hi[i + 1] << 8) + t_lo[i + 1]) >> S1;
			pair[i / 2] = __builtin_amdgcn_perm((uint32_t)v1, (uint32_t)v0, 0x05040100);
		}
		v4i32 b_lo, b_hi;
		{
			uint32_t lo[2], hi[2];
			split_bytes({ pair[0], pair[1], pair[2], pair[3] }, lo, hi);
			b_lo.x = lo[0]; b_lo.y = lo[1]; b_hi.x = hi[0]; b_hi.y = hi[1];
			split_bytes({ pair[4], pair[5], pair[6], pair[7] }, lo, hi);
			b_lo.z = lo[0]; b_lo.w = lo[1]; b_hi.z = hi[0]; b_hi.w = hi[1];
		}

		// stage 2: Out[m][k = r] for this lane's 16 output rows m = (i & 3) + 8 (i >> 2) + 4 h
		v16i32 o_lo, o_hi = { 0 };
#pragma unroll
		for (int i = 0; i < 16; ++i) o_lo[i] = bias2[i];
		o_lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2, b_lo, o_lo, 0, 0, 0);
		o_hi = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2, b_hi, o_hi, 0, 0, 0);

		// Two neighbouring columns per dword store, and every lane stores: for a pair of output rows (i, i + 1) the even lanes
		// write columns (r, r + 1) of row i (the odd lane's value comes by DPP) and the odd lanes columns (r - 1, r) of row
		// i + 1 -- 8 store instructions of 64 dwords instead of 16 of 32 (-2.7 %).
		int16_t *out = coeffs + (size_t)blk * 1024 + (r & ~1);
#pragma unroll
		for (int i = 0; i < 16; i += 2)
		{
			const int v0 = ((o_hi[i] << 8) + o_lo[i]) >> S2, v1 = ((o_hi[i + 1] << 8) + o_lo[i + 1]) >> S2;
			const int from_next = __builtin_amdgcn_mov_dpp(v0, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);   // even lanes: row i, column r + 1
			const int from_prev = __builtin_amdgcn_mov_dpp(v1, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);   // odd lanes: row i + 1, column r - 1
			const uint32_t even_pair = __builtin_amdgcn_perm((uint32_t)from_next, (uint32_t)v0, 0x05040100);
			const uint32_t odd_pair = __builtin_amdgcn_perm((uint32_t)v1, (uint32_t)from_prev, 0x05040100);
			const int m0 = (i & 3) + 8 * (i >> 2) + 4 * h, m1 = ((i + 1) & 3) + 8 * ((i + 1) >> 2) + 4 * h;
			*reinterpret_cast<uint32_t *>(out + 32 * ((l & 1) ? m1 : m0)) = (l & 1) ? odd_pair : even_pair;
		}
#pragma unroll
		for (int i = 0; i < 8; ++i) x[i] = xn[i];
	}
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

}  // namespace hevc

using namespace hevc;

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
