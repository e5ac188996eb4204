// The memory side of the inverse transform + add kernels with the arithmetic taken out: what rate does THIS access mix reach on gfx950?
//   per N x N block: 2 N^2 bytes of coefficients read as one contiguous run (16-byte lane loads), N prediction rows of N bytes read and N
//   destination rows of N bytes written, rows a frame stride apart, blocks tiling 1920 x 1152 frames in raster order (tools/bench_families.py's
//   layout: prediction stride 1920, destination stride 1952), lanes dealt as transform.hip deals them (add stage: consecutive lanes = the same row
//   of consecutive blocks).  Reads : writes = 3 : 1, 4 N^2 algorithmic bytes per block.
// Every transform kernel of every size takes 0.21-0.23 ms for 2^28 coefficients whatever its arithmetic (profiles/r04_transform_valu_trim_ab.txt);
// this probe says how much of that is the access mix itself.   hipcc --offload-arch=gfx950 -O3 transform_mix.hip -o transform_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int N> __global__ __launch_bounds__(256) void mix(uint8_t *dst, const uint8_t *pred, const int16_t *coeffs, uint32_t n_blocks, int per_row, int per_frame)
{
	constexpr int BPWG = 256 / N;
	// XCD-contiguous workgroup order, as the library's kernels use
	const uint32_t nwg = gridDim.x, per_xcd = (nwg + 7) / 8, wg = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
	if (wg >= nwg) return;
	const uint32_t local = threadIdx.x / N, t = threadIdx.x % N;
	uint32_t blk = wg * BPWG + local;
	if (blk >= n_blocks) blk = n_blocks - 1;
	// coefficients: row t of block blk, 2 N bytes
	const u32x4 *c = reinterpret_cast<const u32x4 *>(coeffs + (size_t)blk * N * N + t * N);
	u32x4 acc = c[0];
#pragma unroll
	for (int q = 1; q < 2 * N / 16; ++q) { const u32x4 v = c[q]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
	// add stage deal: lane = (row t2, block local2)
	const uint32_t local2 = threadIdx.x % BPWG, t2 = threadIdx.x / BPWG;
	uint32_t blk2 = wg * BPWG + local2;
	if (blk2 >= n_blocks) blk2 = n_blocks - 1;
	const uint32_t f = blk2 / per_frame, r = blk2 % per_frame, by = r / per_row, bx = r % per_row;
	const size_t po = ((size_t)f * 1152 + by * N + t2) * 1920 + bx * N, dof = ((size_t)f * 1152 + by * N + t2) * 1952 + 16 + bx * N;
	const u32x4 *p = reinterpret_cast<const u32x4 *>(pred + po);
	u32x4 *d = reinterpret_cast<u32x4 *>(dst + dof);
	// the exchange through LDS that makes the stores depend on the coefficient loads of OTHER lanes (as the transposes do)
	__shared__ uint32_t x[256];
	x[threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
	__syncthreads();
	const uint32_t mixin = x[(threadIdx.x * 7 + 3) & 255];
#pragma unroll
	for (int q = 0; q < N / 16; ++q)
	{
		u32x4 v = p[q];
		v.x ^= mixin; v.y += mixin; v.z ^= mixin; v.w += mixin;
		d[q] = v;
	}
}


// Round 5: the same bytes with COALESCED deals.  Coefficients: the workgroup's blocks are one contiguous 16 KiB run; load instruction q of thread tid takes
// bytes [(q * 256 + tid) * 16, + 16) of it -- every wave instruction reads 1 KiB of consecutive bytes (the shipped deal gives a lane its row: 2 N bytes a lane
// apart, so a 16-byte instruction touches a quarter (N = 32) or half (N = 16) of every line and the line is visited again by the next instruction).
// Prediction / destination at N = 32 (WIDE): 16 lanes cover the 256 contiguous bytes of one row of the workgroup's 8 blocks, a thread takes rows r and r + 16
// (shipped: 8 lanes x 32 bytes, two 16-byte instructions each touching every other 16 bytes).  What a kernel pays for it: LDS scatter addresses that depend on
// (q, tid) instead of on the lane's row, and for WIDE an exchange of half rows between lane pairs before the stores.
template <int N, bool WIDE> __global__ __launch_bounds__(256) void mix_coalesced(uint8_t *dst, const uint8_t *pred, const int16_t *coeffs, uint32_t n_blocks, int per_row, int per_frame)
{
	constexpr int BPWG = 256 / N;
	const uint32_t nwg = gridDim.x, per_xcd = (nwg + 7) / 8, wg = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
	if (wg >= nwg) return;
	const size_t run = (size_t)wg * BPWG * N * N;                       // first coefficient of the workgroup's run
	const size_t last = (size_t)n_blocks * N * N - 8;
	u32x4 acc = { 0, 0, 0, 0 };
#pragma unroll
	for (int q = 0; q < BPWG * N * N * 2 / (256 * 16); ++q)
	{
		size_t e = run + ((size_t)q * 256 + threadIdx.x) * 8;
		if (e > last) e = last;
		const u32x4 v = *reinterpret_cast<const u32x4 *>(coeffs + e);
		acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
	}
	__shared__ uint32_t x[256];
	x[threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
	__syncthreads();
	const uint32_t mixin = x[(threadIdx.x * 7 + 3) & 255];
	if (WIDE)
	{
		// lane = (row pair t2 = tid / 16, 16-byte piece = tid % 16): block = piece / 2, half = piece % 2
		const uint32_t piece = threadIdx.x % 16, t2 = threadIdx.x / 16;
		uint32_t blk2 = wg * BPWG + piece / (N / 16);
		if (blk2 >= n_blocks) blk2 = n_blocks - 1;
		const uint32_t f = blk2 / per_frame, r = blk2 % per_frame, by = r / per_row, bx = r % per_row;
#pragma unroll
		for (int h = 0; h < N / 16; ++h)
		{
			const uint32_t row = t2 + 16 * h;
			const size_t po = ((size_t)f * 1152 + by * N + row) * 1920 + bx * N + 16 * (piece % (N / 16));
			const size_t dof = ((size_t)f * 1152 + by * N + row) * 1952 + 16 + bx * N + 16 * (piece % (N / 16));
			u32x4 v = *reinterpret_cast<const u32x4 *>(pred + po);
			v.x ^= mixin; v.y += mixin; v.z ^= mixin; v.w += mixin;
			*reinterpret_cast<u32x4 *>(dst + dof) = v;
		}
	}
	else
	{
		const uint32_t local2 = threadIdx.x % BPWG, t2 = threadIdx.x / BPWG;
		uint32_t blk2 = wg * BPWG + local2;
		if (blk2 >= n_blocks) blk2 = n_blocks - 1;
		const uint32_t f = blk2 / per_frame, r = blk2 % per_frame, by = r / per_row, bx = r % per_row;
		const size_t po = ((size_t)f * 1152 + by * N + t2) * 1920 + bx * N, dof = ((size_t)f * 1152 + by * N + t2) * 1952 + 16 + bx * N;
		const u32x4 *p = reinterpret_cast<const u32x4 *>(pred + po);
		u32x4 *d = reinterpret_cast<u32x4 *>(dst + dof);
#pragma unroll
		for (int q = 0; q < N / 16; ++q)
		{
			u32x4 v = p[q];
			v.x ^= mixin; v.y += mixin; v.z ^= mixin; v.w += mixin;
			d[q] = v;
		}
	}
}

// Deal 3: prediction and destination BLOCK-LINEAR (stride N: block b's rows are the N x N contiguous bytes at b * N * N -- what hevcasm_pred_intra writes,
// and what a codec keeps its prediction scratch in); every lane takes row t of ITS block (no re-deal): all three streams are contiguous runs.
template <int N> __global__ __launch_bounds__(256) void mix_linear(uint8_t *dst, const uint8_t *pred, const int16_t *coeffs, uint32_t n_blocks)
{
	constexpr int BPWG = 256 / N;
	const uint32_t nwg = gridDim.x, per_xcd = (nwg + 7) / 8, wg = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
	if (wg >= nwg) return;
	const uint32_t local = threadIdx.x / N, t = threadIdx.x % N;
	uint32_t blk = wg * BPWG + local;
	if (blk >= n_blocks) blk = n_blocks - 1;
	const u32x4 *c = reinterpret_cast<const u32x4 *>(coeffs + (size_t)blk * N * N + t * N);
	u32x4 acc = c[0];
#pragma unroll
	for (int q = 1; q < 2 * N / 16; ++q) { const u32x4 v = c[q]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
	const u32x4 *p = reinterpret_cast<const u32x4 *>(pred + (size_t)blk * N * N + t * N);
	u32x4 *d = reinterpret_cast<u32x4 *>(dst + (size_t)blk * N * N + t * N);
	__shared__ uint32_t x[256];
	x[threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
	__syncthreads();
	const uint32_t mixin = x[(threadIdx.x * 7 + 3) & 255];
#pragma unroll
	for (int q = 0; q < N / 16; ++q)
	{
		u32x4 v = p[q];
		v.x ^= mixin; v.y += mixin; v.z ^= mixin; v.w += mixin;
		d[q] = v;
	}
}

static bool g_live = false;
static int g_deal = 0;   // 0 = the shipped deals, 1 = coalesced coefficients, 2 = + wide prediction / destination rows (N = 32)
template <int N> void run(size_t coeffs_total)
{
	const size_t n = coeffs_total / (N * N);
	const int per_row = 1920 / N, per_frame = per_row * (1152 / N);
	const size_t frames = (n + per_frame - 1) / per_frame;
	uint8_t *dst, *pred; int16_t *co;
	if (hipMalloc(&dst, frames * 1152 * 1952 + 64) || hipMalloc(&pred, frames * 1152 * 1920 + 64) || hipMalloc(&co, coeffs_total * 2)) { printf("alloc failed\n"); return; }
	hipMemset(pred, 1, frames * 1152 * 1920); hipMemset(co, 2, coeffs_total * 2);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const uint32_t grid = (uint32_t)((n + 256 / N - 1) / (256 / N));
	std::vector<float> ms;
	for (int rep = 0; rep < 8; ++rep)
	{
		hipEventRecord(e0);
		if (g_deal == 0) mix<N><<<grid, 256>>>(dst, pred, co, (uint32_t)n, per_row, per_frame);
		else if (g_deal == 1) mix_coalesced<N, false><<<grid, 256>>>(dst, pred, co, (uint32_t)n, per_row, per_frame);
		else if (g_deal == 2) mix_coalesced<N, true><<<grid, 256>>>(dst, pred, co, (uint32_t)n, per_row, per_frame);
		else mix_linear<N><<<grid, 256>>>(dst, pred, co, (uint32_t)n);
		hipEventRecord(e1); hipDeviceSynchronize();
		float t; hipEventElapsedTime(&t, e0, e1);
		if (rep) ms.push_back(t);
	}
	float best = 1e9f, sum = 0;
	for (float t : ms) { best = t < best ? t : best; sum += t; }
	const double bytes = 4.0 * coeffs_total;
	if (g_live) { printf("idct%d_2p28 %.4f\n", N, sum / ms.size()); hipFree(dst); hipFree(pred); hipFree(co); return; }
	printf("deal %d  %2dx%-2d  2^%d coefficients  mean %.4f ms  min %.4f ms  ->  %.0f GB/s algorithmic (%.3f of 8 TB/s)\n", g_deal, N, N, (int)(63 - __builtin_clzll(coeffs_total)), sum / ms.size(), best,
			bytes / (sum / ms.size() * 1e-3) / 1e9, bytes / (sum / ms.size() * 1e-3) / 8e12);
	hipFree(dst); hipFree(pred); hipFree(co);
}

int main(int argc, char **)
{
	if (argc > 1)
	{
		// bench.py's figures: the access mix of its two 2^28-coefficient inverse-transform rows, on the box and in the run that measures the kernels
		g_live = true;
		run<16>((size_t)1 << 28);
		run<32>((size_t)1 << 28);
		return 0;
	}
	for (int deal = 0; deal < 4; ++deal)
		for (size_t total : { (size_t)1 << 26, (size_t)1 << 28 })
		{
			g_deal = deal;
			run<16>(total);
			run<32>(total);
		}
	return 0;
}
