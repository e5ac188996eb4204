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

namespace hevc {

// ---- coefficients ------------------------------------------------------------------------

struct Matrix32 { signed char m[32][32]; };

constexpr Matrix32 make_matrix32()
{
	// 64 * sqrt(2) * cos(a * pi / 64) as HEVC rounds it; row 0 of the matrix is the constant 64
	constexpr int mag[33] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
		61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0 };
	Matrix32 t = {};
	for (int k = 0; k < 32; ++k)
		for (int i = 0; i < 32; ++i)
		{
			int a = (k * (2 * i + 1)) & 127;
			if (a > 64) a = 128 - a;
			t.m[k][i] = (signed char)(k == 0 ? 64 : (a > 32 ? -mag[64 - a] : mag[a]));
		}
	return t;
}

static constexpr Matrix32 kDct = make_matrix32();

// entry (k, i) of the N-point matrix = row k * 32 / N of the 32-point one
template <int N> __device__ __forceinline__ constexpr int dct(int k, int i) { return kDct.m[k * (32 / N)][i]; }

// ---- 1-D transforms in registers ----------------------------------------------------------

// out[n] = sum_k dct<N>(k, n) * in[k]     (inverse: coefficient index k -> sample index n)
template <int N> struct Inverse1D
{
	static __device__ __forceinline__ void run(const int (&in)[N], int (&out)[N])
	{
		int even[N / 2], e[N / 2];
#pragma unroll
		for (int k = 0; k < N / 2; ++k) even[k] = in[2 * k];
		Inverse1D<N / 2>::run(even, e);
#pragma unroll
		for (int n = 0; n < N / 2; ++n)
		{
			int o = 0;
#pragma unroll
			for (int k = 1; k < N; k += 2) o += dct<N>(k, n) * in[k];
			out[n] = e[n] + o;
			out[N - 1 - n] = e[n] - o;
		}
	}
};
template <> struct Inverse1D<2>
{
	static __device__ __forceinline__ void run(const int (&in)[2], int (&out)[2])
	{
		out[0] = 64 * (in[0] + in[1]);
		out[1] = 64 * (in[0] - in[1]);
	}
};

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
__device__ __forceinline__ void dst4_inverse(const int (&s)[4], int (&o)[4])
{
	o[0] = 29 * s[0] + 74 * s[1] + 84 * s[2] + 55 * s[3];
	o[1] = 55 * s[0] + 74 * s[1] - 29 * s[2] - 84 * s[3];
	o[2] = 74 * (s[0] - s[2] + s[3]);
	o[3] = 84 * s[0] - 74 * s[1] + 55 * s[2] - 29 * s[3];
}
__device__ __forceinline__ void dst4_forward(const int (&s)[4], int (&o)[4])
{
	o[0] = 29 * s[0] + 55 * s[1] + 74 * s[2] + 84 * s[3];
	o[1] = 74 * (s[0] + s[1] - s[3]);
	o[2] = 84 * s[0] - 29 * s[1] - 74 * s[2] + 55 * s[3];
	o[3] = 55 * s[0] - 84 * s[1] + 74 * s[2] - 29 * s[3];
}

template <int N, bool DST> __device__ __forceinline__ void inverse_1d(const int (&in)[N], int (&out)[N])
{
	if constexpr (DST) dst4_inverse(in, out); else Inverse1D<N>::run(in, out);
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
template <int N> __device__ __forceinline__ void global_row_load(const int16_t *p, int (&v)[N])
{
	constexpr int CB = N >= 8 ? 16 : 8;
#pragma unroll
	for (int q = 0; q < 2 * N / CB; ++q)
	{
		Chunk<CB> c;
		c.load(reinterpret_cast<const uint8_t *>(p) + q * CB);
#pragma unroll
		for (int k = 0; k < CB / 4; ++k)
		{
			v[q * (CB / 2) + 2 * k] = (int)(short)(c.d[k] & 0xffff);
			v[q * (CB / 2) + 2 * k + 1] = (int)c.d[k] >> 16;
		}
	}
}
template <int N> __device__ __forceinline__ void global_row_store(int16_t *p, const int (&v)[N])
{
	constexpr int CB = N >= 8 ? 16 : 8;
#pragma unroll
	for (int q = 0; q < 2 * N / CB; ++q)
	{
		Chunk<CB> c;
#pragma unroll
		for (int k = 0; k < CB / 4; ++k)
			c.d[k] = ((uint32_t)v[q * (CB / 2) + 2 * k] & 0xffffu) | ((uint32_t)v[q * (CB / 2) + 2 * k + 1] << 16);
		c.store(reinterpret_cast<uint8_t *>(p) + q * CB);
	}
}

// ---- inverse transform + add ---------------------------------------------------------------

template <int N, bool DST, bool DEQUANT>
__global__ __launch_bounds__(kThreads) void inverse_transform_add_kernel(
	uint8_t *__restrict__ dst, ptrdiff_t stride_dst, const uint8_t *__restrict__ pred, ptrdiff_t stride_pred,
	const uint32_t *__restrict__ dst_off, const uint32_t *__restrict__ pred_off,
	const int16_t *__restrict__ coeffs, int scale, int shift, uint32_t n_blocks)
{
	constexpr int BPWG = kThreads / N;
	constexpr int S = Tile<N>::S;
	__shared__ __attribute__((aligned(16))) int16_t tiles[BPWG * Tile<N>::HALFWORDS];

	const uint32_t wg = xcd_contiguous(blockIdx.x, gridDim.x);
	const uint32_t local = threadIdx.x / N, t = threadIdx.x % N;
	uint32_t blk = wg * BPWG + local;
	const bool live = blk < n_blocks;
	if (!live) blk = n_blocks - 1;
	int16_t *tile = tiles + local * Tile<N>::HALFWORDS;

	int v[N], w[N];

	// row t of the coefficient block -> LDS row t (dequantised on the way when fused)
	global_row_load<N>(coeffs + (size_t)blk * (N * N) + t * N, v);
	if (DEQUANT)
	{
		const int round = 1 << (shift - 1);
#pragma unroll
		for (int k = 0; k < N; ++k) v[k] = clip3(-32768, 32767, (v[k] * scale + round) >> shift);  // quantize.c:53-62
	}
	regs_to_row<N>(tile + t * S, v);
	__syncthreads();

	// first stage: column t, shift 7, clip to int16, written back in place (column t is this lane's)
#pragma unroll
	for (int k = 0; k < N; ++k) v[k] = tile[k * S + t];
	inverse_1d<N, DST>(v, w);
#pragma unroll
	for (int y = 0; y < N; ++y) tile[y * S + t] = (int16_t)clip3(-32768, 32767, (w[y] + 64) >> 7);
	__syncthreads();

	// second stage: row t, shift 12, clip to int16
	row_to_regs<N>(tile + t * S, v);
	inverse_1d<N, DST>(v, w);

	// add to prediction row t.  The reference narrows pred + residual to int16 before clipping
	// to 8 bits (residual_decode.c:350-368); the sum of a byte and a clipped int16 wraps only
	// above 32767, which the narrowing cast reproduces.
	const uint8_t *p = pred + pred_off[blk] + t * stride_pred;
	uint8_t *d = dst + dst_off[blk] + t * stride_dst;
	constexpr int CB = N >= 16 ? 16 : N;
	if (live)
	{
#pragma unroll
		for (int q = 0; q < N / CB; ++q)
		{
			Chunk<CB> c;
			c.load(p + q * CB);
#pragma unroll
			for (int k = 0; k < CB / 4; ++k)
			{
				uint32_t o = 0;
#pragma unroll
				for (int b = 0; b < 4; ++b)
				{
					const int r = clip3(-32768, 32767, (w[q * CB + 4 * k + b] + 2048) >> 12);
					const int s = (int)(short)((int)((c.d[k] >> (8 * b)) & 0xff) + r);
					o |= (uint32_t)clip3(0, 255, s) << (8 * b);
				}
				c.d[k] = o;
			}
			c.store(d + q * CB);
		}
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

	// first stage: row t of the residual; result (truncated to int16) stored transposed
	global_row_load<N>(src + src_off[blk] + (ptrdiff_t)t * src_stride, v);
	forward_1d<N, DST>(v, w);
#pragma unroll
	for (int k = 0; k < N; ++k) tile[k * S + t] = (int16_t)((w[k] + (1 << (S1 - 1))) >> S1);
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
	if (live) global_row_store<N>(coeffs + (size_t)blk * (N * N) + t * N, v);
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
	if (n == 0) return 0;
	if (n > 0xffffffffull) return set_error(HEVCASM_HIP_EINVAL, "n_blocks exceeds 2^32-1");
	if (!dst || !pred || !dof || !pof || !c) return set_error(HEVCASM_HIP_EINVAL, "null pointer");
	hipStream_t st = as_stream(stream);
	if (trType) return launch_inverse<4, true, DEQUANT>(dst, sd, pred, sp, dof, pof, c, scale, shift, n, st);
	switch (log2n)
	{
	case 2: return launch_inverse<4, false, DEQUANT>(dst, sd, pred, sp, dof, pof, c, scale, shift, n, st);
	case 3: return launch_inverse<8, false, DEQUANT>(dst, sd, pred, sp, dof, pof, c, scale, shift, n, st);
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
	if (trType) return launch_forward<4, true>(coeffs, src, src_stride, src_off, n_blocks, st);
	switch (log2TrafoSize)
	{
	case 2: return launch_forward<4, false>(coeffs, src, src_stride, src_off, n_blocks, st);
	case 3: return launch_forward<8, false>(coeffs, src, src_stride, src_off, n_blocks, st);
	case 4: return launch_forward<16, false>(coeffs, src, src_stride, src_off, n_blocks, st);
	default: return launch_forward<32, false>(coeffs, src, src_stride, src_off, n_blocks, st);
	}
}
