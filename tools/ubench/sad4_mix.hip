// The memory side of the four-reference SAD of narrow blocks (config #2, W <= 32) with the arithmetic taken out: how much would a UNION-WINDOW deal save?
//   geometry: blocks tile 1920 x 1152 source frames in raster order (stride 1920); the reference stack has a 16 x 8 border (stride 1952); the four candidates
//   of a block are displaced by (dx, dy) uniform in [-7, 7]^2 (tools/bench.py's case), 1 Mi blocks per launch, buffers rotated so that nothing is cached.
//   deal A (what block_cost_kernel<..., 4, ...> does): per block row one piece of the source and one piece of EACH candidate (W bytes, 8- or 16-byte lanes,
//          many rows to an instruction): 5 H pieces per block, 5 W H lane-bytes;
//   deal B (union window): per block the source rows and ONE window of (H + 14) rows x (W + 14) bytes that holds all four candidates, 16-byte lanes,
//          ceil((W + 14) / 16) lanes to a row: 2 H + 14 pieces per block, W H + (H + 14) * 16 * lanes bytes.
// Every lane folds what it loads into a checksum and one lane per block stores 16 bytes (the four SADs' place): loads dominate, as in the kernel.
// The prediction probes (pred_mix.hip) showed that such loads cost per scattered piece AND instruction, not per byte: this probe says what deal B is worth
// before anybody writes its arithmetic (candidate extraction across lanes).   hipcc --offload-arch=gfx950 -O3 sad4_mix.hip -o sad4_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) U2 { u32x2 v; };
struct __attribute__((packed, aligned(1))) U4 { u32x4 v; };

constexpr int FW = 1920, FH = 1152, RS = FW + 32, RH = FH + 16;

__device__ inline uint32_t wg_contiguous()
{
	const uint32_t nwg = gridDim.x, q = nwg / 8, r8 = nwg % 8, xcd = blockIdx.x % 8, k = blockIdx.x / 8;
	return xcd * q + (xcd < r8 ? xcd : r8) + k;
}
__device__ inline void displacement(uint32_t blk, int c, int &dx, int &dy)
{
	const uint32_t h = (blk * 4u + (uint32_t)c) * 2654435761u;
	dx = (int)((h >> 7) % 15u) - 7;
	dy = (int)((h >> 19) % 15u) - 7;
}
// the descriptors the library's entry point receives (source offset, four reference offsets per block) + the union window's corner for deal B
__global__ void setup(uint32_t *src_off, uint32_t *ref_off, uint32_t *win_off, int W, int H, uint32_t n, int per_row, int per_frame)
{
	const uint32_t blk = blockIdx.x * blockDim.x + threadIdx.x;
	if (blk >= n) return;
	const uint32_t f = blk / per_frame, rr = blk % per_frame, by = rr / per_row, bx = rr % per_row;
	src_off[blk] = (uint32_t)(((size_t)f * FH + by * H) * FW + bx * W);
	int mdx = 7, mdy = 7;
	for (int c = 0; c < 4; ++c)
	{
		int dx, dy; displacement(blk, c, dx, dy);
		mdx = min(mdx, dx); mdy = min(mdy, dy);
		ref_off[4 * (size_t)blk + c] = (uint32_t)(((size_t)f * RH + 8 + by * H + dy) * RS + 16 + bx * W + dx);
	}
	win_off[blk] = (uint32_t)(((size_t)f * RH + 8 + by * H + mdy) * RS + 16 + bx * W + mdx);
}

// deal A: LB bytes per lane, W / LB lanes to a row, 64 / (W / LB) rows to an instruction; a wave takes whole blocks (H rows each), several if they fit
template <int LB> __global__ __launch_bounds__(256) void deal_a(uint32_t *out, const uint8_t *src, const uint8_t *ref, int W, int H, uint32_t n, const uint32_t *src_off, const uint32_t *ref_off)
{
	const uint32_t wave = wg_contiguous() * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	const int lpr = W / LB, rows_per_instr = 64 / lpr;
	const int bpw = rows_per_instr >= H ? rows_per_instr / H : 1;          // blocks per wave
	const int trips = rows_per_instr >= H ? 1 : H / rows_per_instr;
	const int row_in_wave = lane / lpr, j = lane % lpr;
	const int local = rows_per_instr >= H ? row_in_wave / H : 0, row0 = rows_per_instr >= H ? row_in_wave % H : row_in_wave;
	uint32_t blk = wave * bpw + local;
	const bool live = local < bpw && blk < n;
	if (blk >= n) blk = n - 1;
	const uint32_t so0 = src_off[blk];
	const u32x4 ro4 = *reinterpret_cast<const u32x4 *>(ref_off + 4 * (size_t)blk);
	const uint32_t ro0[4] = { ro4.x, ro4.y, ro4.z, ro4.w };
	uint32_t acc = lane;
	for (int t = 0; t < trips; ++t)
	{
		const int row = row0 + t * rows_per_instr;
		const size_t so = (size_t)so0 + (size_t)row * FW + j * LB;
		if (LB == 8) { const u32x2 v = reinterpret_cast<const U2 *>(src + so)->v; acc ^= v.x + v.y; }
		else { const u32x4 v = reinterpret_cast<const U4 *>(src + so)->v; acc ^= (v.x + v.y) ^ (v.z + v.w); }
#pragma unroll
		for (int c = 0; c < 4; ++c)
		{
			const size_t ro = (size_t)ro0[c] + (size_t)row * RS + j * LB;
			if (LB == 8) { const u32x2 v = reinterpret_cast<const U2 *>(ref + ro)->v; acc += (v.x ^ v.y) + c; }
			else { const u32x4 v = reinterpret_cast<const U4 *>(ref + ro)->v; acc += ((v.x ^ v.y) + (v.z ^ v.w)) + c; }
		}
	}
	// one lane per block stores the four sums' place (the others' checksums reach it as the kernel's reduction would)
	acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2);
	if (live && row0 == 0 && j == 0) { u32x4 o; o.x = acc; o.y = acc + 1; o.z = acc + 2; o.w = acc + 3; *reinterpret_cast<u32x4 *>(out + 4 * (size_t)blk) = o; }
}

// deal B: the source as in deal A (16-byte lanes), then the union window: ceil((W + 14) / 16) lanes of 16 bytes to a row, all (H + 14) rows; the window's
// corner is the block's position displaced by the smallest dx and dy of its candidates (spans up to 14 more in each direction: loaded whole here)
__global__ __launch_bounds__(256) void deal_b(uint32_t *out, const uint8_t *src, const uint8_t *ref, int W, int H, uint32_t n, const uint32_t *src_off, const uint32_t *win_off, int bpw)
{
	const uint32_t wave = wg_contiguous() * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	const int lanes_per_block = 64 / bpw, local = lane / lanes_per_block, l = lane % lanes_per_block;
	uint32_t blk = wave * bpw + local;
	const bool live = blk < n;
	if (blk >= n) blk = n - 1;
	const uint32_t so0 = src_off[blk], wo0 = win_off[blk];
	uint32_t acc = lane;
	// source: W / 16 lanes to a row (W = 8: 8-byte lanes)
	{
		const int lb = W >= 16 ? 16 : 8, lpr = W / lb, rpi = lanes_per_block / lpr;
		for (int r0 = 0; r0 < H; r0 += rpi)
		{
			const int row = r0 + l / lpr, j = l % lpr;
			if (row < H && l < rpi * lpr)
			{
				const size_t so = (size_t)so0 + (size_t)row * FW + j * lb;
				if (lb == 8) { const u32x2 v = reinterpret_cast<const U2 *>(src + so)->v; acc ^= v.x + v.y; }
				else { const u32x4 v = reinterpret_cast<const U4 *>(src + so)->v; acc ^= (v.x + v.y) ^ (v.z + v.w); }
			}
		}
	}
	// union window
	{
		const int lpr = (W + 14 + 15) / 16, rpi = lanes_per_block / lpr, rows = H + 14;
		for (int r0 = 0; r0 < rows; r0 += rpi)
		{
			const int row = r0 + l / lpr, j = l % lpr;
			if (row < rows && l < rpi * lpr)
			{
				const size_t ro = (size_t)wo0 + (size_t)row * RS + j * 16;
				const u32x4 v = reinterpret_cast<const U4 *>(ref + ro)->v;
				acc += (v.x ^ v.y) + (v.z ^ v.w);
			}
		}
	}
	acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2);
	if (live && l == 0) { u32x4 o; o.x = acc; o.y = acc + 1; o.z = acc + 2; o.w = acc + 3; *reinterpret_cast<u32x4 *>(out + 4 * (size_t)blk) = o; }
}

static void report(const char *name, int W, int H, uint32_t n, const std::vector<float> &ms)
{
	float best = 1e9f, sum = 0;
	for (float t : ms) { best = t < best ? t : best; sum += t; }
	const double compulsory = (double)n * (W * H * 2.031 + 16 + 20);
	printf("%2dx%-2d  %-34s mean %.4f ms  min %.4f ms  ->  %.0f GB/s of the compulsory bytes (%.3f of 8 TB/s)\n", W, H, name, sum / ms.size(), best,
			compulsory / (sum / ms.size() * 1e-3) / 1e9, compulsory / (sum / ms.size() * 1e-3) / 8e12);
}

int main()
{
	const uint32_t n = 1u << 20;
	const int sizes[][2] = { { 16, 16 }, { 8, 8 }, { 16, 8 }, { 8, 16 }, { 16, 32 }, { 32, 16 }, { 32, 32 }, { 16, 64 }, { 32, 64 }, { 8, 32 }, { 32, 8 } };
	// buffers for the largest case (32 x 64: 1 Mi blocks = 1080 frames), three rotating sets
	const size_t max_frames = (size_t)n / ((FW / 32) * (FH / 64)) + 2;
	uint8_t *src, *ref; uint32_t *out, *so_d, *ro_d, *wo_d;
	const size_t ssz = max_frames * FH * FW, rsz = max_frames * RH * RS;
	if (hipMalloc(&src, 3 * ssz + 65536) || hipMalloc(&ref, 3 * rsz + 65536) || hipMalloc(&out, (size_t)n * 16) || hipMalloc(&so_d, (size_t)n * 4) || hipMalloc(&ro_d, (size_t)n * 16) || hipMalloc(&wo_d, (size_t)n * 4)) { printf("alloc failed\n"); return 1; }
	hipMemset(src, 1, 3 * ssz); hipMemset(ref, 2, 3 * rsz);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (auto &s : sizes)
	{
		const int W = s[0], H = s[1], per_row = FW / W, per_frame = per_row * (FH / H);
		setup<<<(n + 255) / 256, 256>>>(so_d, ro_d, wo_d, W, H, n, per_row, per_frame);
		for (int form = 0; form < 3; ++form)
		{
			if (form == 1 && W < 16) continue;
			std::vector<float> ms;
			char name[64];
			for (int rep = 0; rep < 6; ++rep)
			{
				const uint8_t *s_ = src + (size_t)(rep % 3) * ssz, *r_ = ref + (size_t)(rep % 3) * rsz;
				hipEventRecord(e0);
				if (form < 2)
				{
					const int LB = form ? 16 : 8, lpr = W / LB, rpi = 64 / lpr, bpw = rpi >= H ? rpi / H : 1;
					const uint32_t waves = (n + bpw - 1) / bpw, grid = (waves + 3) / 4;
					if (form) deal_a<16><<<grid, 256>>>(out, s_, r_, W, H, n, so_d, ro_d); else deal_a<8><<<grid, 256>>>(out, s_, r_, W, H, n, so_d, ro_d);
					snprintf(name, sizeof name, "deal A, %2d-byte lanes", LB);
				}
				else
				{
					// blocks per wave: as many as leave a block's window rows about one trip
					const int lpr = (W + 14 + 15) / 16, need = (H + 14) * lpr;
					const int bpw = need <= 64 ? 4 : need <= 128 ? 2 : 1;      // 16 / 32 / 64 lanes to a block, a few trips over its rows
					const uint32_t waves = (n + bpw - 1) / bpw, grid = (waves + 3) / 4;
					deal_b<<<grid, 256>>>(out, s_, r_, W, H, n, so_d, wo_d, bpw);
					snprintf(name, sizeof name, "deal B (union window), %d block(s)/wave", bpw);
				}
				hipEventRecord(e1); hipDeviceSynchronize();
				float t; hipEventElapsedTime(&t, e0, e1);
				if (rep) ms.push_back(t);
			}
			report(name, W, H, n, ms);
		}
	}
	return 0;
}
