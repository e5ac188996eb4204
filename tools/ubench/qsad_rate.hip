// Issue rate of the byte-SAD instructions a full search can be built from, gfx950 -- chains of INDEPENDENT
// accumulators and nothing else in the loop (tools/ubench/valu_rates.hip's qsad line carries a v_xor and 64-bit
// register pairing; this one pins the instruction itself).
//   v_sad_u8            4 byte differences per lane-op
//   v_qsad_pk_u16_u8   16 (four 4-byte SADs at byte offsets 0..3 of an 8-byte window, packed u16 accumulate)
//   v_mqsad_pk_u16_u8  16 (masked variant)
//   v_mqsad_u32_u8     16 (masked, four u32 accumulators)
// 8 waves per SIMD resident.  hipcc --offload-arch=gfx950 -O3 qsad_rate.hip -o qsad_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#define ITER 2048
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
enum { SAD, QSAD, MQSAD, MQSAD32, SADPAIR };

template <int OP> __global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
	const uint32_t x = seed ^ threadIdx.x, y = seed * 3 + 1;
	const unsigned long long win = ((unsigned long long)y << 32) | x;
	uint32_t a[16];
	unsigned long long q[8];
	u32x4 m[4];
#pragma unroll
	for (int i = 0; i < 16; ++i) a[i] = i;
#pragma unroll
	for (int i = 0; i < 8; ++i) q[i] = i;
#pragma unroll
	for (int i = 0; i < 4; ++i) m[i] = u32x4{ 0, 1, 2, 3 };
	for (int it = 0; it < ITER; ++it)
	{
		if (OP == SAD)
#pragma unroll
			for (int i = 0; i < 16; ++i) a[i] = __builtin_amdgcn_sad_u8(x, y + i, a[i]);
		else if (OP == QSAD)
#pragma unroll
			for (int r = 0; r < 2; ++r)
#pragma unroll
				for (int i = 0; i < 8; ++i) q[i] = __builtin_amdgcn_qsad_pk_u16_u8(win, x + i, q[i]);
		else if (OP == MQSAD)
#pragma unroll
			for (int r = 0; r < 2; ++r)
#pragma unroll
				for (int i = 0; i < 8; ++i) q[i] = __builtin_amdgcn_mqsad_pk_u16_u8(win, x + i, q[i]);
		else if (OP == MQSAD32)
#pragma unroll
			for (int r = 0; r < 4; ++r)
#pragma unroll
				for (int i = 0; i < 4; ++i) m[i] = __builtin_amdgcn_mqsad_u32_u8(win, x + i, m[i]);
		else if (OP == SADPAIR)   // what a qsad-free search issues per 16 byte differences: 4 x v_sad_u8 + 1 realignment
#pragma unroll
			for (int i = 0; i < 16; i += 4)
			{
				const uint32_t w = __builtin_amdgcn_alignbyte(y + i, a[(i + 4) & 15], 1);   // depends on the loop, or the compiler hoists it
				a[i] = __builtin_amdgcn_sad_u8(w, y, a[i]); a[i + 1] = __builtin_amdgcn_sad_u8(w, y + 1, a[i + 1]);
				a[i + 2] = __builtin_amdgcn_sad_u8(w, y + 2, a[i + 2]); a[i + 3] = __builtin_amdgcn_sad_u8(w, y + 3, a[i + 3]);
			}
	}
	uint32_t s = 0;
#pragma unroll
	for (int i = 0; i < 16; ++i) s ^= a[i];
#pragma unroll
	for (int i = 0; i < 8; ++i) s ^= (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32);
#pragma unroll
	for (int i = 0; i < 4; ++i) s ^= m[i].x ^ m[i].y ^ m[i].z ^ m[i].w;
	out[blockIdx.x * 256 + threadIdx.x] = s;
}

static int g_cus = 256;
static bool g_quiet = false;
static double g_last = 0;      // T byte differences/s of the last run

template <int OP> void run(const char *name, double diffs_per_inst, double insts_per_iter)
{
	const int blocks = g_cus * 8;   // 8 workgroups of 4 waves per CU -> 8 waves per SIMD
	uint32_t *out;
	hipMalloc(&out, blocks * 256 * 4);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	k<OP><<<blocks, 256>>>(out, 12345);
	float best = 1e9f;
	for (int rep = 0; rep < 3; ++rep)
	{
		hipEventRecord(e0); k<OP><<<blocks, 256>>>(out, 12345); hipEventRecord(e1); hipDeviceSynchronize();
		float ms; hipEventElapsedTime(&ms, e0, e1);
		best = ms < best ? ms : best;
	}
	const double insts_per_simd = 8.0 * ITER * insts_per_iter;
	const double wall_cycles = best * 1e-3 * 2.4e9;
	const double chip = (double)g_cus * 4 * 64 * diffs_per_inst * insts_per_simd / (best * 1e-3);    // byte differences per second, whole chip
	g_last = chip / 1e12;
	if (!g_quiet) printf("%-18s %.3f ms  %.2f cycles per wave-instruction @2.4 GHz  ->  %.1f T byte differences/s per chip\n", name, best, wall_cycles / insts_per_simd, chip / 1e12);
	hipFree(out);
}

// `qsad_rate live`: only the v_qsad_pk_u16_u8 rate, as one number (T byte differences/s) -- what bench.py prices the full search against,
// measured in the same run on the same box (clocks differ from box to box by more than the kernel's distance from this roof)
int main(int argc, char **argv)
{
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, 0) == hipSuccess && prop.multiProcessorCount > 0) g_cus = prop.multiProcessorCount;
	if (argc > 1 && !strcmp(argv[1], "live"))
	{
		g_quiet = true;
		run<QSAD>("v_qsad_pk_u16_u8", 16, 16);
		printf("%.3f\n", g_last);
		return 0;
	}
	run<SAD>("v_sad_u8", 4, 16); run<QSAD>("v_qsad_pk_u16_u8", 16, 16); run<MQSAD>("v_mqsad_pk_u16_u8", 16, 16); run<MQSAD32>("v_mqsad_u32_u8", 16, 16);
	run<SADPAIR>("4 sad_u8 + alignbyte", 16.0 / 5, 20);
	return 0;
}
