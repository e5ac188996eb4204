// How fast can 16x16 tiles of an 8-bit frame be streamed, by lane mapping?  Pure loads (xor-reduced), no SAD arithmetic:
// the ceiling of the narrow-block kernels' access pattern, separated from everything else they do.
//   X  block-major (what block_cost_kernel<16,...> does): lane = tile * 8 + row * 2 + chunk, 8-byte lanes, 4 rows in flight
//   Y  row-major across the wave's 8 tiles: lane = row * 16 + tile * 2 + chunk  (16 lanes = 128 contiguous bytes of a frame row)
//   Z  16-byte lanes, row-major: lane = row * 8 + tile  (8 lanes = 128 contiguous bytes), 8 rows in flight
// each with the tile origin displaced per tile by (dx, dy) in [-7, 7]^2 ("ref") or not at all ("src").
// hipcc --offload-arch=gfx950 -O3 tools/ubench/tile_stream.hip -o /tmp/tile_stream && /tmp/tile_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) U2 { u32x2 v; };
typedef u32x2 u32x2_u __attribute__((aligned(1)));
struct __attribute__((packed, aligned(1))) U4 { u32x4 v; };
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// workgroups are dealt round-robin over the 8 XCDs: give each XCD one contiguous run of the tile list (as the library's
// kernels do), so that tiles which share frame lines meet in the same L2 -- without it the displaced pattern fetches
// 1.6x its bytes from HBM (vertical neighbours on another XCD) and the figure says nothing about the kernels
__device__ __forceinline__ uint32_t xcd_contiguous(uint32_t wg, uint32_t n_wg)
{
	const uint32_t q = n_wg / 8, r = n_wg % 8, xcd = wg % 8, k = wg / 8;
	return xcd * q + (xcd < r ? xcd : r) + k;
}

template <int MAP>
__global__ __launch_bounds__(256) void stream(const uint8_t *__restrict__ frame, uint32_t stride, const uint32_t *__restrict__ off, uint32_t *out, uint32_t n)
{
	const uint32_t wave = xcd_contiguous(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	uint32_t acc = 0;
	if (MAP == 0)
	{
		const uint32_t t = lane >> 3, r = (lane >> 1) & 3, c = lane & 1;
		const uint32_t o = off[wave * 8 + t] + r * stride + c * 8;
		u32x2 v[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) v[u] = reinterpret_cast<const U2 *>(frame + o + u * 4 * stride)->v;
#pragma unroll
		for (int u = 0; u < 4; ++u) acc ^= v[u].x ^ v[u].y;
	}
	else if (MAP == 1)
	{
		const uint32_t r = lane >> 4, t = (lane >> 1) & 7, c = lane & 1;
		const uint32_t o = off[wave * 8 + t] + r * stride + c * 8;
		u32x2 v[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) v[u] = reinterpret_cast<const U2 *>(frame + o + u * 4 * stride)->v;
#pragma unroll
		for (int u = 0; u < 4; ++u) acc ^= v[u].x ^ v[u].y;
	}
	else
	{
		const uint32_t r = lane >> 3, t = lane & 7;
		const uint32_t o = off[wave * 8 + t] + r * stride;
		u32x4 v[2];
#pragma unroll
		for (int u = 0; u < 2; ++u) v[u] = reinterpret_cast<const U4 *>(frame + o + u * 8 * stride)->v;
#pragma unroll
		for (int u = 0; u < 2; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
	}
	if (acc == 0x12345678u) out[0] = acc;
}

// both streams in one wave, mapping X: what block_cost_kernel<16, 8, 4, 4, 1, SAD, false> loads, without its arithmetic
template <bool NT_SRC>
__global__ __launch_bounds__(256) void stream_both(const uint8_t *__restrict__ src, const uint8_t *__restrict__ ref, uint32_t stride,
		const uint32_t *__restrict__ soff, const uint32_t *__restrict__ roff, uint32_t *out, uint32_t n)
{
	const uint32_t wave = xcd_contiguous(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	const uint32_t t = lane >> 3, r = (lane >> 1) & 3, c = lane & 1;
	const uint32_t so = soff[wave * 8 + t] + r * stride + c * 8, ro = roff[wave * 8 + t] + r * stride + c * 8;
	u32x2 a[4], b[4];
#pragma unroll
	for (int u = 0; u < 4; ++u)
	{
		if (NT_SRC) a[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x2_u *>(src + so + u * 4 * stride));   // streaming: read once
		else a[u] = reinterpret_cast<const U2 *>(src + so + u * 4 * stride)->v;
		b[u] = reinterpret_cast<const U2 *>(ref + ro + u * 4 * stride)->v;
	}
	uint32_t acc = 0;
#pragma unroll
	for (int u = 0; u < 4; ++u) acc ^= a[u].x ^ a[u].y ^ b[u].x ^ b[u].y;
	if (acc == 0x12345678u) out[0] = acc;
}

int main()
{
	const uint32_t W = 1920, H = 1152, n = 1u << 20, per_frame = (W / 16) * (H / 16), frames = (n + per_frame - 1) / per_frame;
	const uint32_t stride = W + 32, rows = H + 16;
	const size_t bytes = (size_t)stride * rows * frames + 4096;
	uint8_t *frame; uint32_t *off, *out;
	CK(hipMalloc(&frame, bytes)); CK(hipMalloc(&off, 4 * n)); CK(hipMalloc(&out, 4));
	CK(hipMemset(frame, 3, bytes));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	for (int displaced = 0; displaced < 3; ++displaced)
	{
		std::vector<uint32_t> h(n);
		srand(1);
		for (uint32_t i = 0; i < n; ++i)
		{
			const uint32_t f = i / per_frame, k = i % per_frame, by = k / (W / 16), bx = k % (W / 16);
			int dx = 0, dy = 0;
			if (displaced == 1) { dx = rand() % 15 - 7; dy = rand() % 15 - 7; }
			if (displaced == 2) { dx = (rand() % 4 - 2) * 4; dy = rand() % 15 - 7; }
			h[i] = (uint32_t)(((size_t)f * rows + 8 + by * 16 + dy) * stride + 16 + bx * 16 + dx);
		}
		CK(hipMemcpy(off, h.data(), 4 * n, hipMemcpyHostToDevice));
		for (int map = 0; map < 3; ++map)
		{
			float best = 1e30f;
			for (int rep = 0; rep < 8; ++rep)
			{
				// another 256 MiB pass in between so that the Infinity Cache holds none of the frame
				CK(hipMemsetAsync(frame + bytes - 4096, rep, 4096, 0));
				CK(hipEventRecord(e0));
				if (map == 0) hipLaunchKernelGGL(stream<0>, dim3(n / 32), dim3(256), 0, 0, frame, stride, off, out, n);
				else if (map == 1) hipLaunchKernelGGL(stream<1>, dim3(n / 32), dim3(256), 0, 0, frame, stride, off, out, n);
				else hipLaunchKernelGGL(stream<2>, dim3(n / 32), dim3(256), 0, 0, frame, stride, off, out, n);
				CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
				float ms; CK(hipEventElapsedTime(&ms, e0, e1));
				if (rep && ms < best) best = ms;
			}
			printf("%-28s map %c  %7.4f ms  %7.1f GB/s (256 B per tile + 4 B descriptor)\n",
				displaced == 0 ? "tiles in place (src)" : displaced == 1 ? "displaced +-7 (ref)" : "displaced, dx multiple of 4", "XYZ"[map], best, n * 260.0 / best * 1e-6);
		}
	}
	// the SAD 16x16 pattern: one frame stack read in place, a second one displaced
	{
		uint8_t *frame2; uint32_t *off2;
		CK(hipMalloc(&frame2, bytes)); CK(hipMalloc(&off2, 4 * n));
		CK(hipMemset(frame2, 5, bytes));
		std::vector<uint32_t> h0(n), h1(n);
		srand(2);
		for (uint32_t i = 0; i < n; ++i)
		{
			const uint32_t f = i / per_frame, k = i % per_frame, by = k / (W / 16), bx = k % (W / 16);
			const int dx = rand() % 15 - 7, dy = rand() % 15 - 7;
			h0[i] = (uint32_t)(((size_t)f * rows + 8 + by * 16) * stride + 16 + bx * 16);
			h1[i] = (uint32_t)(((size_t)f * rows + 8 + by * 16 + dy) * stride + 16 + bx * 16 + dx);
		}
		CK(hipMemcpy(off, h0.data(), 4 * n, hipMemcpyHostToDevice));
		CK(hipMemcpy(off2, h1.data(), 4 * n, hipMemcpyHostToDevice));
		for (int nt = 0; nt < 2; ++nt)
		{
			float best = 1e30f;
			for (int rep = 0; rep < 8; ++rep)
			{
				CK(hipEventRecord(e0));
				if (nt) hipLaunchKernelGGL(stream_both<true>, dim3(n / 32), dim3(256), 0, 0, frame, frame2, stride, off, off2, out, n);
				else hipLaunchKernelGGL(stream_both<false>, dim3(n / 32), dim3(256), 0, 0, frame, frame2, stride, off, off2, out, n);
				CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
				float ms; CK(hipEventElapsedTime(&ms, e0, e1));
				if (rep && ms < best) best = ms;
			}
			printf("%-28s map X  %7.4f ms  %7.1f GB/s (512 B per tile pair + 8 B descriptors)\n", nt ? "the same, source loads nt" : "in place + displaced (SAD)", best, n * 520.0 / best * 1e-6);
		}
	}
	return 0;
}
