// What a plain streaming kernel reaches on this MI355X: read-only sum, copy, and write-only fill over
// buffers far larger than the 256 MB Infinity Cache.  Gives the "achievable" line the block kernels'
// algorithmic GB/s can be read against (the roofline peak stays the 8 TB/s datasheet figure).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/hbm_stream.hip -o /tmp/hbm_stream && /tmp/hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void read_sum(const u32x4 *__restrict__ in, size_t n, uint32_t *out)
{
	u32x4 acc = {0, 0, 0, 0};
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += in[i];
	const uint32_t s = acc.x ^ acc.y ^ acc.z ^ acc.w;
	if (s == 0x12345678u) out[0] = s;   // practically never: keeps the loads alive
}
__global__ __launch_bounds__(256) void read_sum_nt(const u32x4 *__restrict__ in, size_t n, uint32_t *out)
{
	u32x4 acc = {0, 0, 0, 0};
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += __builtin_nontemporal_load(in + i);
	const uint32_t s = acc.x ^ acc.y ^ acc.z ^ acc.w;
	if (s == 0x12345678u) out[0] = s;
}
__global__ __launch_bounds__(256) void copy_nt(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}
__global__ __launch_bounds__(256) void copy(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
__global__ __launch_bounds__(256) void fill(u32x4 *__restrict__ out, size_t n, uint32_t v)
{
	const u32x4 x = {v, v, v, v};
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = x;
}

int main()
{
	const size_t bytes = (size_t)4 << 30, n = bytes / 16;
	u32x4 *a, *b; uint32_t *o;
	CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 4));
	CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const int grids[] = {256 * 8, 256 * 16, 256 * 32, 256 * 64, 1 << 20};
	for (int g : grids)
		for (int which = 0; which < 3; ++which)
		{
			float best = 1e30f;
			for (int rep = 0; rep < 6; ++rep)
			{
				CK(hipEventRecord(e0));
				if (which == 0) hipLaunchKernelGGL(read_sum, dim3(g), dim3(256), 0, 0, a, n, o);
				else if (which == 1) hipLaunchKernelGGL(copy, dim3(g), dim3(256), 0, 0, a, b, n);
				else hipLaunchKernelGGL(fill, dim3(g), dim3(256), 0, 0, b, n, 7u);
				CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
				float ms; CK(hipEventElapsedTime(&ms, e0, e1));
				if (rep && ms < best) best = ms;
			}
			const double moved = which == 1 ? 2.0 * bytes : (double)bytes;
			printf("%-8s grid %8d  %7.3f ms  %7.1f GB/s\n", which == 0 ? "read" : which == 1 ? "copy" : "fill", g, best, moved / best * 1e-6);
		}
	// streaming (non-temporal) forms, one-shot workgroups
	for (int which = 0; which < 2; ++which)
	{
		float best = 1e30f;
		for (int rep = 0; rep < 6; ++rep)
		{
			CK(hipEventRecord(e0));
			if (which == 0) hipLaunchKernelGGL(read_sum_nt, dim3(1 << 20), dim3(256), 0, 0, a, n, o);
			else hipLaunchKernelGGL(copy_nt, dim3(1 << 20), dim3(256), 0, 0, a, b, n);
			CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
			float ms; CK(hipEventElapsedTime(&ms, e0, e1));
			if (rep && ms < best) best = ms;
		}
		printf("%-8s grid %8d  %7.3f ms  %7.1f GB/s\n", which == 0 ? "read nt" : "copy nt", 1 << 20, best, (which ? 2.0 : 1.0) * bytes / best * 1e-6);
	}
	// Short launches: one pass over S bytes with one-shot workgroups, every repetition on a different slice of the
	// 4 GiB buffer (so the 256 MB Infinity Cache never holds it).  This is the line the bench's 0.1-0.4 ms kernels
	// (1 Mi small blocks per launch) are to be read against: launch ramp-up and the tail are a fixed cost.
	for (size_t mb : {64, 128, 256, 512, 1024, 2048})
	{
		const size_t sn = (mb << 20) / 16;
		const int slices = (int)(n / sn);
		float best = 1e30f, sum = 0; int cnt = 0;
		for (int rep = 0; rep < 12; ++rep)
		{
			const u32x4 *p = a + (size_t)(rep % slices) * sn;
			CK(hipEventRecord(e0));
			hipLaunchKernelGGL(read_sum, dim3((unsigned)(sn / 256)), dim3(256), 0, 0, p, sn, o);
			CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
			float ms; CK(hipEventElapsedTime(&ms, e0, e1));
			if (rep >= 2) { if (ms < best) best = ms; sum += ms; ++cnt; }
		}
		printf("read one-shot %5zu MB  best %7.4f ms %7.1f GB/s   mean %7.4f ms %7.1f GB/s\n", mb, best, (mb << 20) / best * 1e-6, sum / cnt, (mb << 20) / (sum / cnt) * 1e-6);
	}
	return 0;
}
