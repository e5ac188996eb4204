// Does a load that finds its line already on the way (requested by an EARLIER instruction of the same wave, fill still pending)
// cost the vector L1 less than a miss, or more?  Each wave streams down its own cold tile of the frame the way the prediction kernel
// does: 21 "PUs" of 3 lanes x 8 bytes side by side (16 bytes apart), 8 row loads in flight, PU p starting dy[p] rows down:
//   coherent  dy = 0          -- the PUs of one instruction share lines, no line is touched by two instructions
//   scattered dy = (5 p) % 8  -- neighbouring PUs reach the same line 1..7 instructions apart (random motion vectors do this)
//   disjoint  dy = 32 p       -- no two PUs ever share a line (most misses)
// Prints ns per wave-instruction per CU and the distinct 128-byte lines a wave touches per instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/tcp_pending.hip -o hevcasm_amd/lib_ab/tcp_pending
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(1))) U2 { u32x2 v; };
constexpr int PITCH = 1952, ROWS = 64, UNROLL = 8;

template <int MODE, int MIS>
__global__ __launch_bounds__(256) void probe(const uint8_t *buf, uint32_t *out, size_t span)
{
	const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	const int lane = threadIdx.x & 63, p = lane / 3, c = lane % 3;
	const int dy = MODE == 0 ? 0 : MODE == 1 ? (5 * p) % 8 : 32 * p;
	// tiles: 336 bytes wide; a wave's tile starts 512 bytes after its neighbour's, ROWS + 8 (+ 21 * 32) rows tall
	const size_t tile_rows = MODE == 2 ? ROWS + 21 * 32 : ROWS + 8;
	const uint8_t *q = buf + ((wave * tile_rows * PITCH) % span) + (size_t)dy * PITCH + p * 16 + c * 8 + MIS;
	uint32_t acc = 0;
	for (int r = 0; r < ROWS; r += UNROLL)
	{
		u32x2 v[UNROLL];
#pragma unroll
		for (int u = 0; u < UNROLL; ++u) v[u] = reinterpret_cast<const U2 *>(q + (size_t)(r + u) * PITCH)->v;
#pragma unroll
		for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y;
	}
	if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE, int MIS> void run(const uint8_t *buf, uint32_t *out, size_t span, const char *what)
{
	const int wgs = 256 * 24;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	float best = 1e9f;
	for (int rep = 0; rep < 3; ++rep)
	{
		hipEventRecord(e0);
		hipLaunchKernelGGL((probe<MODE, MIS>), dim3(wgs), dim3(256), 0, 0, buf, out, span);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		if (ms < best) best = ms;
	}
	const double instrs_per_cu = (double)wgs * 4 * ROWS / 256.0;
	printf("%-10s start %% 4 = %d : %7.2f ns per wave-instruction per CU, %.3f ms\n", what, MIS & 3, best * 1e6 / instrs_per_cu, best);
}

int main()
{
	const size_t span = (size_t)6 << 30;
	uint8_t *buf; uint32_t *out;
	if (hipMalloc(&buf, span + ((size_t)64 << 20)) != hipSuccess) { printf("alloc failed\n"); return 1; }
	hipMalloc(&out, 4);
	hipMemset(buf, 1, span + ((size_t)64 << 20));
	printf("cold tiles (HBM):\n");
	run<0, 0>(buf, out, span, "coherent"); run<1, 0>(buf, out, span, "scattered"); run<2, 0>(buf, out, span, "disjoint");
	run<0, 1>(buf, out, span, "coherent"); run<1, 1>(buf, out, span, "scattered"); run<2, 1>(buf, out, span, "disjoint");
	// 48 tiles (16 for disjoint) shared by all waves: L2-resident after the first launch, far more than one CU's L1 holds
	printf("L2-resident tiles:\n");
	const size_t small = (size_t)48 * (ROWS + 8) * PITCH, small2 = (size_t)16 * (ROWS + 21 * 32) * PITCH;
	run<0, 0>(buf, out, small, "coherent"); run<1, 0>(buf, out, small, "scattered"); run<2, 0>(buf, out, small2, "disjoint");
	run<0, 1>(buf, out, small, "coherent"); run<1, 1>(buf, out, small, "scattered"); run<2, 1>(buf, out, small2, "disjoint");
	return 0;
}
