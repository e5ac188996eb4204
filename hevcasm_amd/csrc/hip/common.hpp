// common.hpp -- shared device helpers for the gfx950 kernels (wave64, CDNA4 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "hevcasm_hip.h"

namespace hevc {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kXcds = 8;           // MI355X: 8 XCDs, each with a private L2
constexpr int kThreads = 256;      // 4 waves per workgroup, one per SIMD

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Loads with no alignment promise.  gfx950 runs with unaligned access enabled, so these
// compile to one global_load_dword{,x2,x4} whatever the address.
struct __attribute__((packed, aligned(1))) U1 { uint32_t v; };
struct __attribute__((packed, aligned(1))) U2 { u32x2 v; };
struct __attribute__((packed, aligned(1))) U3 { uint32_t a, b, c; };
struct __attribute__((packed, aligned(1))) U4 { u32x4 v; };
struct __attribute__((packed, aligned(1))) H1 { uint16_t v; };

// the same, as streaming (non-temporal) loads: for data a kernel touches exactly once, so that it does not push the
// lines other waves are about to reuse out of the vector L1
typedef uint32_t u32x1_u __attribute__((aligned(1)));
typedef u32x2 u32x2_u __attribute__((aligned(1)));
typedef u32x3 u32x3_u __attribute__((aligned(1)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));

template <int BYTES> struct Chunk;
template <> struct Chunk<4>
{
	uint32_t d[1];
	__device__ __forceinline__ void load(const uint8_t *p) { d[0] = reinterpret_cast<const U1 *>(p)->v; }
	__device__ __forceinline__ void load_stream(const uint8_t *p) { d[0] = __builtin_nontemporal_load(reinterpret_cast<const u32x1_u *>(p)); }
	__device__ __forceinline__ void store(uint8_t *p) const { reinterpret_cast<U1 *>(p)->v = d[0]; }
	__device__ __forceinline__ void store_stream(uint8_t *p) const { __builtin_nontemporal_store(d[0], reinterpret_cast<u32x1_u *>(p)); }
};
template <> struct Chunk<8>
{
	uint32_t d[2];
	__device__ __forceinline__ void load(const uint8_t *p)
	{
		const u32x2 v = reinterpret_cast<const U2 *>(p)->v;
		d[0] = v.x; d[1] = v.y;
	}
	__device__ __forceinline__ void load_stream(const uint8_t *p)
	{
		const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2_u *>(p));
		d[0] = v.x; d[1] = v.y;
	}
	__device__ __forceinline__ void store(uint8_t *p) const
	{
		u32x2 v; v.x = d[0]; v.y = d[1];
		reinterpret_cast<U2 *>(p)->v = v;
	}
	__device__ __forceinline__ void store_stream(uint8_t *p) const
	{
		u32x2 v; v.x = d[0]; v.y = d[1];
		__builtin_nontemporal_store(v, reinterpret_cast<u32x2_u *>(p));
	}
};
template <> struct Chunk<12>
{
	uint32_t d[3];
	__device__ __forceinline__ void load(const uint8_t *p)
	{
		const U3 v = *reinterpret_cast<const U3 *>(p);   // one global_load_dwordx3
		d[0] = v.a; d[1] = v.b; d[2] = v.c;
	}
	__device__ __forceinline__ void load_stream(const uint8_t *p)
	{
		const u32x3 v = __builtin_nontemporal_load(reinterpret_cast<const u32x3_u *>(p));
		d[0] = v.x; d[1] = v.y; d[2] = v.z;
	}
	__device__ __forceinline__ void store(uint8_t *p) const
	{
		U3 v; v.a = d[0]; v.b = d[1]; v.c = d[2];
		*reinterpret_cast<U3 *>(p) = v;
	}
};
template <> struct Chunk<16>
{
	uint32_t d[4];
	__device__ __forceinline__ void load(const uint8_t *p)
	{
		const u32x4 v = reinterpret_cast<const U4 *>(p)->v;
		d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
	}
	__device__ __forceinline__ void load_stream(const uint8_t *p)
	{
		const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_u *>(p));
		d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
	}
	__device__ __forceinline__ void store(uint8_t *p) const
	{
		u32x4 v; v.x = d[0]; v.y = d[1]; v.z = d[2]; v.w = d[3];
		reinterpret_cast<U4 *>(p)->v = v;
	}
	__device__ __forceinline__ void store_stream(uint8_t *p) const
	{
		u32x4 v; v.x = d[0]; v.y = d[1]; v.z = d[2]; v.w = d[3];
		__builtin_nontemporal_store(v, reinterpret_cast<u32x4_u *>(p));
	}
};

// Workgroups are dealt round-robin over the 8 XCDs.  Give each XCD one contiguous run of the
// block list so that neighbouring blocks (which share cache lines of the frame rows they
// straddle) meet in the same L2.  Bijective for any grid size; speed only, never correctness.
__device__ __forceinline__ uint32_t xcd_contiguous(uint32_t wg, uint32_t n_wg)
{
	const uint32_t q = n_wg / kXcds, r = n_wg % kXcds;
	const uint32_t xcd = wg % kXcds, k = wg / kXcds;
	// XCD x owns q (+1 if x < r) consecutive slots
	const uint32_t start = xcd * q + (xcd < r ? xcd : r);
	return start + k;
}

// wave-wide reductions of a 32-bit value (every lane ends up with the result)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, d, kWave));
	return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, kWave));
	return v;
}

// v[idx] for a per-lane idx < N out of N wave-uniform values (SGPRs): a binary tree of v_cndmask on the bits of idx when N is a power of two
// (N - 1 selects; the bit tests are shared by every array picked with the same idx), a chain of compares otherwise
template <int N> __device__ __forceinline__ uint32_t pick_by_index(const uint32_t (&v)[N], uint32_t idx)
{
	if constexpr ((N & (N - 1)) == 0)
	{
		uint32_t t[N];
#pragma unroll
		for (int i = 0; i < N; ++i) t[i] = v[i];
#pragma unroll
		for (int bit = 0; (1 << bit) < N; ++bit)
		{
			const bool odd = (idx >> bit) & 1u;
#pragma unroll
			for (int i = 0; i < (N >> (bit + 1)); ++i) t[i] = odd ? t[2 * i + 1] : t[2 * i];
		}
		return t[0];
	}
	else
	{
		uint32_t r = v[0];
#pragma unroll
		for (int i = 1; i < N; ++i) r = idx == (uint32_t)i ? v[i] : r;
		return r;
	}
}

__device__ __forceinline__ int clip3(int lo, int hi, int v) { return min(max(v, lo), hi); }

// host side -------------------------------------------------------------------------------

int set_error(int code, const char *what);     // records thread-local text, returns code
int check_device();                            // 0 or HEVCASM_HIP_ENODEV; the current device must be a gfx950; clears the thread's stale HIP error
int device_cus();                              // compute units of the current device (cached; 256 on an unpartitioned MI355X)
inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

#define HEVC_HIP_TRY(expr) \
	do { hipError_t e_ = (expr); if (e_ != hipSuccess) return hevc::set_error((int)e_, hipGetErrorString(e_)); } while (0)

// check_device() at the top of the entry point has cleared any earlier error of this thread: what is found here is this launch's
inline int launch_status(const char *kernel)
{
	hipError_t e = hipGetLastError();
	return e == hipSuccess ? 0 : set_error((int)e, kernel);
}

inline uint32_t ceil_div(size_t a, size_t b) { return (uint32_t)((a + b - 1) / b); }

}  // namespace hevc
