// intra_predict.hpp -- the prediction of ONE intra block from its neighbour record staged in LDS, shared by the flat batch kernel (pred_intra.hip) and the
// frame pipeline (intra_frame.hip).  INTRA_DC is the reference's C (pred_intra.c:61-101); planar and the 33 angular modes follow H.265 8.4.4.2.4 / .6
// (the reference has them only as un-wired f265 assembly, f265/intra.asm:216-1394): PARITY UNPINNED beyond DC, see pred_intra.hip.
#pragma once

#include "common.hpp"

namespace hevc {

// intraPredAngle and invAngle of H.265 8.4.4.2.6 (tables 8-4, 8-5) from packed literals, no memory: with d = the mode's distance from its pure direction
// (10 - mode for the horizontal family 2..17, mode - 26 for the vertical one 18..34; -8 .. 8) the angle is sign(d) * { 0, 2, 5, 9, 13, 17, 21, 26, 32 }[|d|]
// and the inverse angle of a negative one -{ 4096, 1638, 910, 630, 482, 390, 315, 256 }[|d| - 1].  (Round 6: these were two __constant__ arrays -- in the
// frame pipeline, where the mode is wave-uniform, two dependent scalar-memory loads on every TU's critical path.)
__device__ __forceinline__ int intra_angle(int mode)
{
	const int d = mode < 18 ? 10 - mode : mode - 26, a = min(d < 0 ? -d : d, 8);      // (modes 0 and 1 have no angle: whatever comes out is not used)
	const int mag = a == 8 ? 32 : (int)((0x1a15110d09050200ull >> (8 * a)) & 0xffu);
	return d < 0 ? -mag : mag;
}
__device__ __forceinline__ int intra_inv_angle(int mode)      // modes 11 .. 25 (negative angles)
{
	const int d = mode < 18 ? 10 - mode : mode - 26, a = (d < 0 ? -d : d) - 1;      // 0 .. 7
	const unsigned long long packed = a < 4 ? 0x0276038e06661000ull : 0x0100013b018601e2ull;      // 4096 1638 910 630 | 482 390 315 256, 16 bits each
	return -(int)((packed >> (16 * (a & 3))) & 0xffffu);
}

typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_mad_u16(uint32_t a, uint32_t b, uint32_t c)
{
	uint32_t r;
	asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
}
__device__ __forceinline__ uint32_t pk_lshr_u16(uint32_t a, uint32_t sh)
{
	uint32_t r;
	asm("v_pk_lshrrev_b16 %0, %1, %2" : "=v"(r) : "v"(sh | (sh << 16)), "v"(a));     // the shift count is packed too: one per half
	return r;
}

template <int N> struct IntraGeom
{
	static constexpr int PX = N < 16 ? N : 16;        // output bytes per lane = one family row (4x4, 8x8: a whole row; 16: a row; 32: half a row)
	static constexpr int LPB = N * N / PX;            // lanes per block: 4, 8, 16, 64
	static constexpr int LOG2 = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
	static constexpr int REF = 3 * N + 12;            // ref[] entries i = -N .. 2N + 1 at [N + i]; the dword reads of a row may run 8 bytes past the last entry
	static constexpr int RAW = 4 * N + 4;             // the compact record: raw[2N - 1 - y] = p[-1][y], raw[2N + x] = p[x][-1], raw[4N] = p[-1][-1]
};

// raw: the block's 4N + 1 neighbour samples (compact record, above), already in LDS and visible to the block's lanes.  ref (REF bytes) and tile (N * N bytes)
// are scratch of the block.  Lane j of the block (j < LPB computes; with STRIDE > LPB the lanes beyond only help staging) ends up with the PX output bytes at
// row-major position PX * j of the N x N block.  block_sync: makes the block's LDS writes visible to its lanes (a wave-level fence when they share a wave).
// Second form (round 5).  Everything angular is computed in FAMILY space: a horizontal mode m is the transpose of the vertical mode 36 - m on swapped
// edges, so the staging step puts the edge the mode runs along ("main": top for modes 18..34, left for 2..17) into ref[] and the kernel always filters
// along it; a lane owns 16 family-space bytes = R rows of L = min(N, 16) pixels, each row one (iIdx, iFact): L + 1 consecutive ref bytes come out of LDS
// as aligned dwords + v_alignbyte, and the blend ((32 - f) a + f b + 16) >> 5 runs on even / odd bytes with v_pk_mad_u16 (11 instructions per 4 pixels;
// f = 0 gives a exactly, so no branch).  Vertical modes store the 16 bytes as they are; horizontal modes scatter them transposed into an LDS tile and
// read their row-major 16 bytes back.  Planar and DC stay per pixel (few operations each).
// PXO: output bytes per lane (a multiple of 4 that divides N; N * N / PXO lanes compute).  The flat kernel uses IntraGeom's deal -- many blocks per workgroup, few
// lanes each; the frame pipeline, whose wave works on ONE block, deals a 16x16 block over all 64 lanes (PXO = 4).
template <int N, int STRIDE, int PXO = IntraGeom<N>::PX, typename Sync>
__device__ __forceinline__ void intra_predict_block(const uint8_t *raw, uint8_t *ref, uint8_t *tile, const int j, const int mode, const int edge, Sync &&block_sync,
	Chunk<PXO> &out)
{
	constexpr int PX = PXO, LPB = N * N / PXO, LOG2 = IntraGeom<N>::LOG2, L = PX;
	static_assert(PXO % 4 == 0 && N % PXO == 0 && LPB <= STRIDE, "a lane's bytes are whole dwords of one row");
	const bool computes = STRIDE == LPB || j < LPB;
	auto top_at = [&](int x) { return (int)raw[2 * N + x]; };          // p[x][-1]
	auto left_at = [&](int y) { return (int)raw[2 * N - 1 - y]; };     // p[-1][y]
	const int angle = intra_angle(mode);
	const bool vertical = mode >= 18;
	if (mode >= 2)
	{
		// ref[N + i], i >= 0: the corner, then the edge the mode runs along -- whole dwords at a time: the top row lies in raw[] in this order one byte
		// off a dword boundary (v_alignbyte), the left column in the opposite order (v_alignbyte + a byte reversal); entries past 2N are never used by a
		// sample that matters (they meet a zero weight) and take whatever the guard leaves
		const uint32_t *raw32 = reinterpret_cast<const uint32_t *>(raw);
		uint32_t *ref32w = reinterpret_cast<uint32_t *>(ref);
		for (int q = j; q < (2 * N + 12) / 4; q += STRIDE)
		{
			uint32_t w = 0;
			if (q <= 2 * N / 4)
			{
				if (vertical)
				{
					const uint32_t hi = q < 2 * N / 4 ? raw32[2 * N / 4 + q] : 0u;
					w = __builtin_amdgcn_alignbyte(hi, raw32[2 * N / 4 + q - 1], 3);      // bytes raw[2N - 1 + 4q ...]
				}
				else
				{
					const uint32_t lo = q < 2 * N / 4 ? raw32[2 * N / 4 - q - 1] : 0u;
					w = __builtin_amdgcn_perm(0u, __builtin_amdgcn_alignbyte(raw32[2 * N / 4 - q], lo, 1), 0x00010203u);   // bytes raw[2N - 4q], raw[2N - 4q - 1], ...
				}
				if (q == 0) w = (w & 0xffffff00u) | raw[4 * N];
			}
			ref32w[N / 4 + q] = w;
		}
		if (angle < 0)
		{
			// i < 0: the other edge projected by the inverse angle (8.4.4.2.6); entries below (N * angle) >> 5 are never read
			const int inv = intra_inv_angle(mode);
			for (int e = j; e < N; e += STRIDE)
			{
				const int i = e - N;
				const int k = min((i * inv + 128) >> 8, 2 * N);
				ref[e] = (uint8_t)(vertical ? left_at(k - 1) : top_at(k - 1));
			}
		}
	}
	else if (mode == 0)
	{
		for (int e = j; e < 2 * N + 2; e += STRIDE) ref[e] = (uint8_t)(e <= N ? top_at(e) : left_at(e - N - 1));      // top 0..N, then left 0..N
	}
	else
	{
		for (int e = j; e < 2 * N; e += STRIDE) ref[e] = (uint8_t)(e < N ? top_at(e) : left_at(e - N));               // top 0..N-1, then left 0..N-1
	}
	block_sync();

	const int jc = computes ? j : 0;      // (lanes that only helped staging repeat lane 0's rows: same values to the same places, no branch)
	if (mode >= 2)
	{
		const uint32_t *ref32 = reinterpret_cast<const uint32_t *>(ref);
		const int corner = ref[N], first = ref[N + 1];
		{
			const int v = PX * jc / N, u0 = PX * jc % N;
			const int tt = (v + 1) * angle, idx = tt >> 5;
			const uint32_t f = (uint32_t)tt & 31u;
			const int s = N + u0 + idx + 1;
			const uint32_t sh = (uint32_t)s & 3u;
			uint32_t d[L / 4 + 2], a[L / 4 + 1];
#pragma unroll
			for (int k = 0; k < L / 4 + 2; ++k) d[k] = ref32[(s >> 2) + k];
#pragma unroll
			for (int k = 0; k < L / 4 + 1; ++k) a[k] = __builtin_amdgcn_alignbyte(d[k + 1], d[k], sh);
			const uint32_t w1 = f | (f << 16), w0 = 0x00200020u - w1;
#pragma unroll
			for (int k = 0; k < L / 4; ++k)
			{
				const uint32_t b = __builtin_amdgcn_alignbyte(a[k + 1], a[k], 1);
				const uint32_t ae = a[k] & 0x00ff00ffu, ao = (a[k] >> 8) & 0x00ff00ffu, be = b & 0x00ff00ffu, bo = (b >> 8) & 0x00ff00ffu;
				const uint32_t re = pk_lshr_u16(pk_mad_u16(ae, w0, pk_mad_u16(be, w1, 0x00100010u)), 5);
				const uint32_t ro = pk_lshr_u16(pk_mad_u16(ao, w0, pk_mad_u16(bo, w1, 0x00100010u)), 5);
				out.d[k] = re | (ro << 8);
			}
			if (edge && angle == 0 && u0 == 0)
			{
				// modes 26 / 10 with the boundary smoothing: the pixel next to the main edge takes half the gradient along the other edge
				const int other = vertical ? left_at(v) : top_at(v);
				out.d[0] = (out.d[0] & 0xffffff00u) | (uint32_t)clip3(0, 255, first + ((other - corner) >> 1));
			}
		}
	}
	else if (mode == 0)
	{
		const int tn = ref[N], ln = ref[2 * N + 1];
		const uint32_t *top32 = reinterpret_cast<const uint32_t *>(ref);
		{
			const int y = PX * jc / N, x0 = PX * jc % N;
			const int ly = ref[N + 1 + y];
			const int row = (y + 1) * ln + N, step = tn - ly;
			int across = (N - 1 - x0) * ly + (x0 + 1) * tn + row;
#pragma unroll
			for (int k = 0; k < L / 4; ++k)
			{
				const uint32_t tw = top32[x0 / 4 + k];
				uint32_t px = 0;
#pragma unroll
				for (int b = 0; b < 4; ++b)
				{
					px |= (uint32_t)(((int)((tw >> (8 * b)) & 0xff) * (N - 1 - y) + across) >> (LOG2 + 1)) << (8 * b);
					across += step;
				}
				out.d[k] = px;
			}
		}
	}
	else
	{
		uint32_t sum = 0;
#pragma unroll
		for (int q = 0; q < 2 * N / 4; ++q) sum = __builtin_amdgcn_sad_u8(reinterpret_cast<const uint32_t *>(ref)[q], 0, sum);
		const int dc = (int)(sum + N) >> (LOG2 + 1);
		uint8_t px[PX];
#pragma unroll
		for (int b = 0; b < PX; ++b)
		{
			const int o = PX * jc + b, y = o / N, x = o % N;
			int val = dc;
			if (edge)
			{
				if (x == 0 && y == 0) val = (ref[N] + 2 * dc + ref[0] + 2) >> 2;
				else if (y == 0) val = (ref[x] + 3 * dc + 2) >> 2;
				else if (x == 0) val = (ref[N + y] + 3 * dc + 2) >> 2;
			}
			px[b] = (uint8_t)val;
		}
#pragma unroll
		for (int k = 0; k < PX / 4; ++k)
			out.d[k] = px[4 * k] | ((uint32_t)px[4 * k + 1] << 8) | ((uint32_t)px[4 * k + 2] << 16) | ((uint32_t)px[4 * k + 3] << 24);
	}
	// horizontal modes: the lane holds family row v (output COLUMN v), positions u0 .. u0 + L - 1 (output rows): through the tile, transposed
	const bool transposed = mode >= 2 && !vertical;
	if (transposed)
	{
		const int v = PX * jc / N, u0 = PX * jc % N;
#pragma unroll
		for (int b = 0; b < L; ++b) tile[(u0 + b) * N + v] = (uint8_t)(out.d[b / 4] >> (8 * (b % 4)));
	}
	block_sync();
	if (transposed)
	{
#pragma unroll
		for (int k = 0; k < PX / 4; ++k) out.d[k] = reinterpret_cast<const uint32_t *>(tile + PX * jc)[k];
	}
}

}  // namespace hevc
