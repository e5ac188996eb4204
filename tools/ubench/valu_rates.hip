// Issue rate of the integer VALU instructions the kernels lean on, gfx950.
// Each kernel runs ITER iterations of 16 independent instructions per lane; 8 waves per SIMD resident.
// Prints wave-instructions per cycle per SIMD (cycles from s_memtime).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s16x2 __attribute__((ext_vector_type(2)));
#define ITER 2048
enum { MAD24, MUL24, ADD3, DOT2, PKMAD, PERM, DPP, SAD, DOT4, MULLO, ALIGNBIT, MED3, QSAD, SWIZZLE };

template <int OP> __global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed, unsigned long long *cycles)
{
	uint32_t a[16];
	const uint32_t x = seed ^ threadIdx.x, y = seed * 3 + 1;
#pragma unroll
	for (int i = 0; i < 16; ++i) a[i] = x + i;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < ITER; ++it)
	{
#pragma unroll
		for (int i = 0; i < 16; ++i)
		{
			if (OP == MAD24) a[i] = (uint32_t)(((int)(a[i] << 8) >> 8) * ((int)(y << 8) >> 8)) + x;   // v_mad_i32_i24
			else if (OP == MUL24) a[i] = (uint32_t)(((int)(a[i] << 8) >> 8) * ((int)(y << 8) >> 8));  // v_mul_i32_i24
			else if (OP == ADD3) a[i] = a[i] + x + y;
			else if (OP == DOT2) a[i] = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, x), __builtin_bit_cast(s16x2, y), (int)a[i], false);
			else if (OP == PKMAD) { s16x2 r = __builtin_bit_cast(s16x2, a[i]) * __builtin_bit_cast(s16x2, y) + __builtin_bit_cast(s16x2, x); a[i] = __builtin_bit_cast(uint32_t, r); }
			else if (OP == PERM) a[i] = __builtin_amdgcn_perm(a[i], x, 0x0c020c00 + y);
			else if (OP == DPP) a[i] = __builtin_amdgcn_mov_dpp((int)a[i], 0x130, 0xf, 0xf, true);
			else if (OP == SAD) a[i] = __builtin_amdgcn_sad_u8(x, y, a[i]);
			else if (OP == DOT4) a[i] = __builtin_amdgcn_udot4(x, y, a[i], false);
			else if (OP == MULLO) a[i] = a[i] * y;
			else if (OP == ALIGNBIT) a[i] = __builtin_amdgcn_alignbit(a[i], x, 16);
			else if (OP == MED3) a[i] = (uint32_t)min(max((int)a[i], (int)x), (int)y);
			else if (OP == QSAD) { unsigned long long r = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)y << 32) | x, x, ((unsigned long long)a[(i + 1) & 15] << 32) | a[i]); a[i] = (uint32_t)r ^ (uint32_t)(r >> 32); }
			else if (OP == SWIZZLE) a[i] = (uint32_t)__builtin_amdgcn_ds_swizzle((int)a[i], 0x101f);
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	uint32_t s = 0;
#pragma unroll
	for (int i = 0; i < 16; ++i) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
	if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int OP> void run(const char *name)
{
	const int blocks = 256 * 8;   // 8 workgroups of 4 waves per CU -> 8 waves per SIMD
	uint32_t *out; unsigned long long *cyc;
	hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	k<OP><<<blocks, 256>>>(out, 12345, cyc);
	hipEventRecord(e0); k<OP><<<blocks, 256>>>(out, 12345, cyc); hipEventRecord(e1); hipDeviceSynchronize();
	float ms; hipEventElapsedTime(&ms, e0, e1);
	unsigned long long h[256 * 8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
	double avg = 0; for (int i = 0; i < blocks; ++i) avg += (double)h[i]; avg /= blocks;
	// per SIMD: 8 waves x ITER x 16 instructions in `avg` cycles (s_memtime ticks at 100 MHz on some parts: report wall too)
	const double insts_per_simd = 8.0 * ITER * 16;
	const double wall_cycles = ms * 1e-3 * 2.4e9;
	printf("%-10s  %.3f ms  wave-instr/cycle/SIMD (wall @2.4GHz) = %.3f   -> %.2f cycles per wave-instruction;  s_memtime delta %.0f\n",
		name, ms, insts_per_simd / wall_cycles, wall_cycles / insts_per_simd, avg);
	hipFree(out); hipFree(cyc);
}

int main()
{
	run<MAD24>("mad_i24"); run<MUL24>("mul_i24"); run<ADD3>("add3"); run<DOT2>("dot2_i16"); run<PKMAD>("pk_mad_u16");
	run<PERM>("perm"); run<DPP>("dpp wave_shl"); run<SAD>("sad_u8"); run<DOT4>("dot4_u8"); run<MULLO>("mul_lo_u32"); run<ALIGNBIT>("alignbit"); run<MED3>("min/max"); run<QSAD>("qsad_pk_u16_u8"); run<SWIZZLE>("ds_swizzle");
	return 0;
}
