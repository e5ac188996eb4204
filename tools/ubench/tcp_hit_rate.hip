// How fast does one CU's vector memory pipeline (TA + TCP) serve loads that HIT the L1, per load shape?
// Every wave of the chip loops over the same few KB (rows `ROWPITCH` bytes apart, L1-resident after the first trip) with one
// load shape: BYTES per lane, LPR lanes per row at PITCH bytes, first byte of a row at MIS.  8 independent loads per trip.
// Prints ns per wave-instruction per CU (x clock = cycles): the service time the prediction / SAD kernels' queues run against.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/tcp_hit_rate.hip -o hevcasm_amd/lib_ab/tcp_hit_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) U1 { uint32_t v; };
struct __attribute__((packed, aligned(1))) U2 { u32x2 v; };
struct __attribute__((packed, aligned(1))) U4 { u32x4 v; };

constexpr int ROWPITCH = 1952, TRIPS = 128, UNROLL = 8;

template <int BYTES> __device__ __forceinline__ uint32_t load(const uint8_t *p)
{
	if (BYTES == 4) return reinterpret_cast<const U1 *>(p)->v;
	if (BYTES == 8) { const u32x2 v = reinterpret_cast<const U2 *>(p)->v; return v.x ^ v.y; }
	const u32x4 v = reinterpret_cast<const U4 *>(p)->v; return v.x ^ v.y ^ v.z ^ v.w;
}

template <int BYTES, int LPR, int MIS, int PITCH, int FOOT>
__global__ __launch_bounds__(256) void probe(const uint8_t *buf, uint32_t *out)
{
	const int lane = threadIdx.x & 63, row = lane / LPR, c = lane % LPR;
	// the 8 loads of a trip read 8 consecutive rows' worth (as a kernel streaming down a block would); all waves the same addresses
	const uint8_t *p = buf + 4096 + (size_t)row * ROWPITCH + MIS + c * PITCH;
	uint32_t acc = 0;
	for (int t = 0; t < TRIPS; ++t)
	{
		uint32_t v[UNROLL];
#pragma unroll
		for (int u = 0; u < UNROLL; ++u)
		{
			const uint8_t *q = p + FOOT * u;   // FOOT = 0: the same lines again (L1 hits); 192: 8 lines per row in play (22 KB for 22 rows: L1 / L2 mix)
			asm volatile("" : "+v"(q));        // (or the compiler folds the identical loads into one)
			v[u] = load<BYTES>(q);
		}
#pragma unroll
		for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
		asm volatile("" : "+v"(acc));
	}
	if (acc == 0x12345678u) out[0] = acc;
}

// stores: ACTIVE of every LPR lanes write BYTES (prediction's output rows: the margin lanes do not store); every wave its own rows
template <int BYTES, int LPR, int ACTIVE, bool SIDE>
__global__ __launch_bounds__(256) void store_probe(uint8_t *buf)
{
	const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	const int lane = threadIdx.x & 63, pu = lane / LPR, c = lane % LPR;
	// SIDE: the PUs of a wave lie side by side on one row (prediction's output: ACTIVE * BYTES contiguous bytes per PU); else one row per PU
	uint8_t *p = buf + (wave % 2048) * (1 << 18) + (SIDE ? (size_t)pu * ACTIVE * BYTES : (size_t)pu * 1920) + c * BYTES;
	for (int t = 0; t < TRIPS; ++t)
	{
#pragma unroll
		for (int u = 0; u < UNROLL; ++u)
			if (c < ACTIVE)
			{
				if (BYTES == 8) { u32x2 v; v.x = t; v.y = u; __builtin_nontemporal_store(v, reinterpret_cast<u32x2 *>(p + (size_t)((t * UNROLL + u) % 128) * 1920)); }
				else *reinterpret_cast<uint32_t *>(p + (size_t)((t * UNROLL + u) % 128) * 1920) = t;
			}
	}
}

template <int BYTES, int LPR, int ACTIVE, bool SIDE> void run_store(uint8_t *buf, const char *what)
{
	const int wgs = 256 * 8;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	float best = 1e9f;
	for (int rep = 0; rep < 4; ++rep)
	{
		hipEventRecord(e0);
		hipLaunchKernelGGL((store_probe<BYTES, LPR, ACTIVE, SIDE>), dim3(wgs), dim3(256), 0, 0, buf);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		if (rep && ms < best) best = ms;
	}
	const double instrs_per_cu = (double)wgs * 4 * TRIPS * UNROLL / 256.0;
	printf("STORE %2d bytes/lane, %d of %d lanes per PU, %s : %7.2f ns per wave-instruction per CU  (%s)\n", BYTES, ACTIVE, LPR, SIDE ? "PUs side by side on one row" : "one row per PU", best * 1e6 / instrs_per_cu, what);
}

template <int BYTES, int LPR, int MIS, int PITCH = BYTES, int FOOT = 0> void run(const uint8_t *buf, uint32_t *out, const char *what)
{
	const int wgs = 256 * 8;        // 8 workgroups = 32 waves per CU
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	float best = 1e9f;
	for (int rep = 0; rep < 4; ++rep)
	{
		hipEventRecord(e0);
		hipLaunchKernelGGL((probe<BYTES, LPR, MIS, PITCH, FOOT>), dim3(wgs), dim3(256), 0, 0, buf, out);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		if (rep && ms < best) best = ms;
	}
	const double instrs_per_cu = (double)wgs * 4 * TRIPS * UNROLL / 256.0;
	printf("%s %2d bytes/lane, %2d lanes/row at pitch %2d (%2d rows/wave), row start %% 4 = %d : %7.2f ns per wave-instruction per CU  (%s)\n",
		FOOT ? "L2" : "L1", BYTES, LPR, PITCH, (64 + LPR - 1) / LPR, MIS & 3, best * 1e6 / instrs_per_cu, what);
}

int main()
{
	uint8_t *buf; uint32_t *out;
	hipMalloc(&buf, 1 << 20); hipMalloc(&out, 4);
	hipMemset(buf, 1, 1 << 20);
	run<4, 64, 0>(buf, out, "one 256-byte row: the floor");
	run<16, 64, 0>(buf, out, "one 1 KiB row");
	run<8, 3, 1>(buf, out, "prediction, 16-wide PU: 3 lanes x 8 bytes, byte alignment");
	run<8, 3, 0>(buf, out, "the same, dword-aligned");
	run<8, 9, 1>(buf, out, "prediction, 64-wide PU: 9 lanes x 8 bytes");
	run<8, 9, 0>(buf, out, "the same, dword-aligned");
	run<4, 3, 1>(buf, out, "prediction, 4-wide PU: 3 lanes x 4 bytes");
	run<4, 3, 0>(buf, out, "the same, dword-aligned");
	run<16, 2, 0, 8>(buf, out, "16-wide PU as 2 overlapping 16-byte lanes (the P16 experiment)");
	run<16, 2, 1, 8>(buf, out, "the same, byte alignment");
	run<16, 2, 0>(buf, out, "32-byte rows as 2 aligned 16-byte lanes (LDS-window staging)");
	run<16, 4, 0>(buf, out, "64-byte rows as 4 aligned 16-byte lanes");
	run<16, 5, 0>(buf, out, "80-byte rows as 5 aligned 16-byte lanes");
	run<8, 2, 0>(buf, out, "16-byte rows as 2 aligned 8-byte lanes (SAD 16-wide)");
	run<8, 2, 1>(buf, out, "the same, byte alignment");
	run<16, 1, 0>(buf, out, "16-byte rows, one lane each");
	run<16, 1, 1>(buf, out, "the same, byte alignment");
	run<8, 1, 0>(buf, out, "8-byte rows, one lane each");
	run<4, 1, 0>(buf, out, "4-byte rows, one lane each");
	run<8, 3, 1, 8, 192>(buf, out, "prediction 16-wide, 8 lines per row in play");
	run<8, 3, 0, 8, 192>(buf, out, "the same, dword-aligned");
	run<16, 2, 0, 16, 192>(buf, out, "32-byte rows as 2 aligned 16-byte lanes, 8 lines per row in play");
	run<16, 4, 0, 16, 192>(buf, out, "64-byte rows as 4 aligned 16-byte lanes, 8 lines per row in play");
	uint8_t *big; hipMalloc(&big, ((size_t)2048 << 18) + (1 << 20));
	run_store<8, 3, 2, true>(big, "prediction 16-wide output rows");
	run_store<8, 3, 2, false>(big, "the same bytes, every PU on a row of its own");
	run_store<8, 9, 8, true>(big, "prediction 64-wide output rows");
	run_store<4, 3, 1, true>(big, "prediction 4-wide output rows");
	run_store<8, 64, 64, true>(big, "one 512-byte row");
	return 0;
}
