// What does a grid of workgroups that exit at once cost?  (A second kernel behind a first one that has already done the
// work, checking a flag or a per-block sentinel and leaving: the price of deciding on the device which of two kernels runs.)
// Variants: exit on a scalar flag (uniform, one s_load); exit after one vector load per lane-group (a per-block sentinel);
// 64- and 256-thread workgroups; VGPR-heavy kernel descriptor (launch_bounds-independent: registers are part of dispatch).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/empty_grid.hip -o /tmp/empty_grid && /tmp/empty_grid
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int T> __global__ __launch_bounds__(T) void exit_on_flag(const int *flag, int *out)
{
	if (*flag == 0) return;
	out[blockIdx.x * T + threadIdx.x] = 1;
}
template <int T> __global__ __launch_bounds__(T) void exit_on_sentinel(const int *sent, int *out)
{
	// one sentinel per 32 lanes (a 64-wide block pair per wave): leave unless one of the wave's two is -1
	const int v = sent[(blockIdx.x * T + threadIdx.x) >> 5 << 2];
	if (__builtin_amdgcn_ballot_w64(v == -1) == 0) return;
	out[blockIdx.x * T + threadIdx.x] = v;
}
// the same with a large register footprint (the generic 4-way kernel has 236 VGPRs)
template <int T> __global__ __launch_bounds__(T) void exit_on_flag_fat(const int *flag, int *out, const int *in)
{
	if (*flag == 0) return;
	int a[200];
#pragma unroll
	for (int i = 0; i < 200; ++i) a[i] = in[threadIdx.x + 64 * i];
	int s = 0;
#pragma unroll
	for (int i = 0; i < 200; ++i) s += a[i] * (i + 1);
	out[blockIdx.x * T + threadIdx.x] = s;
}

template <typename F> static float time_ms(F f, int reps = 20)
{
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	f(); CK(hipDeviceSynchronize());
	CK(hipEventRecord(e0));
	for (int i = 0; i < reps; ++i) f();
	CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1));
	return ms / reps;
}

int main()
{
	int *flag, *out, *sent;
	const size_t n = 1 << 26;
	CK(hipMalloc(&flag, 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&sent, n)); 
	CK(hipMemset(flag, 0, 4)); CK(hipMemset(sent, 0, n));
	for (int wgs : { 4096, 32768, 131072, 262144 })
	{
		printf("%7d workgroups: exit on scalar flag 256 thr %7.2f us | 64 thr x4 grid %7.2f us | fat (VGPR-heavy) 256 thr %7.2f us | exit on per-pair sentinel 256 thr %7.2f us\n", wgs,
			1e3f * time_ms([&] { hipLaunchKernelGGL(exit_on_flag<256>, dim3(wgs), dim3(256), 0, 0, flag, out); }),
			1e3f * time_ms([&] { hipLaunchKernelGGL(exit_on_flag<64>, dim3(wgs * 4), dim3(64), 0, 0, flag, out); }),
			1e3f * time_ms([&] { hipLaunchKernelGGL(exit_on_flag_fat<256>, dim3(wgs), dim3(256), 0, 0, flag, out, sent); }),
			1e3f * time_ms([&] { hipLaunchKernelGGL(exit_on_sentinel<256>, dim3(wgs), dim3(256), 0, 0, sent, out); }));
	}
	return 0;
}
