// The memory side of the luma uni-prediction kernel with the arithmetic taken out: what rate does THIS access mix reach on gfx950?
//   per W x H PU: H + 7 window rows of W + 7 bytes read (8-byte lane loads at byte alignment, ceil((W + 7) / 8) lanes per PU, 64 / that many PUs per
//   wave, D rows in flight per lane), H destination rows of W bytes written (8 bytes per producing lane), rows a frame stride apart, PUs tiling
//   1920 x 1080 frames in raster order, the window displaced by an integer motion vector in [-8, 8]^2 (tools/bench_families.py's layout: destination
//   stride 1920, reference frames with a 16-pixel border).  Workgroups in XCD-contiguous order, as the library's kernels take them.
// Two kernels with very different instruction counts (pred_stream_kernel and the H-first experiment, profiles/r04_pred_hfirst_ab.txt) take the same time;
// this probe says how much of that time is the access mix itself.  Modes: 0 = loads and stores, 1 = loads only (one store per lane at the end),
// 2 = stores only (one load per lane at the start).   hipcc --offload-arch=gfx950 -O3 pred_mix.hip -o pred_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int LB> struct Vec;
template <> struct Vec<8> { typedef u32x2 T; };
template <> struct Vec<16> { typedef u32x4 T; };
template <int LB> struct __attribute__((packed, aligned(1))) UV { typename Vec<LB>::T v; };
__device__ inline void put(uint8_t *p, u32x2 v, int) { *reinterpret_cast<u32x2 *>(p) = v; }
__device__ inline void put(uint8_t *p, u32x4 v, int sb) { if (sb == 16) *reinterpret_cast<u32x4 *>(p) = v; else if (sb == 4) *reinterpret_cast<uint32_t *>(p) = v.x; else { u32x2 h; h.x = v.x; h.y = v.y; *reinterpret_cast<u32x2 *>(p) = h; } }
__device__ inline uint32_t fold(u32x2 v) { return v.x + v.y; }
__device__ inline uint32_t fold(u32x4 v) { return (v.x + v.y) ^ (v.z + v.w); }

struct Geo { int FW, FH, RS, RH, margin, before, pitch, store_bytes, DS, split; long long plane2_ref, plane2_dst; /* round 5: a lane does the SAME PU of a second plane too (Cb + Cr in one lane): byte distances, 0 = off */ };
constexpr Geo kLuma = { 1920, 1080, 1952, 1112, 7, 3, 0, 0, 0, 0, 0, 0 };
constexpr Geo kChroma = { 960, 540, 992, 572, 3, 1, 0, 0, 0, 0, 0, 0 };

template <int D, int LB, bool ALIGNED, int REFS> __global__ __launch_bounds__(256) void mix(uint8_t *dst, const uint8_t *ref, const uint8_t *ref1, int W, int H, int L, int ppw, uint32_t n, int per_row, int per_frame, int mode, int coherent, Geo g)
{
	const uint32_t nwg = gridDim.x, q = nwg / 8, r8 = nwg % 8, xcd = blockIdx.x % 8, k = blockIdx.x / 8;
	const uint32_t wg = xcd * q + (xcd < r8 ? xcd : r8) + k;
	const uint32_t wave = wg * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	const uint32_t local = lane / L;
	uint32_t j = lane % L;
	// split deal (pred_hfirst_bi8_kernel's): a PU's lanes are its 8-column strips of reference 0, then those of reference 1; each loads ITS reference's
	// 16 window bytes per row and stores 4 of the strip's 8 output bytes
	const uint32_t which = g.split ? j / (L / 2) : 0;
	if (g.split) j %= L / 2;
	uint32_t pu = wave * ppw + local;
	const bool active = local < (uint32_t)ppw && pu < n;
	if (pu >= n) pu = n - 1;
	const uint32_t f = pu / per_frame, rr = pu % per_frame, by = rr / per_row, bx = rr % per_row;
	const uint32_t h = pu * 2654435761u;
	const int mvx = coherent ? 3 : (int)((h >> 8) % 17u) - 8, mvy = coherent ? -2 : (int)((h >> 17) % 17u) - 8;
	const int mvx1 = coherent ? -5 : (int)((h >> 3) % 17u) - 8, mvy1 = coherent ? 4 : (int)((h >> 12) % 17u) - 8;
	typedef typename Vec<LB>::T V;
	const int FW = g.FW, FH = g.FH, RS = g.RS, RH = g.RH, pitch = g.pitch ? g.pitch : LB, SB = g.store_bytes ? g.store_bytes : LB;
	size_t ro = ((size_t)f * RH + 16 + by * H - g.before + (which ? mvy1 : mvy)) * RS + 16 + bx * W - g.before + (which ? mvx1 : mvx) + pitch * j;
	if (which) ref = ref1;
	if (ALIGNED) ro &= ~(size_t)3;              // the aligned words around the lane's bytes (a real kernel then loads one dword more and realigns in registers)
	const int DS = g.DS ? g.DS : FW;
	const size_t dof = ((size_t)f * FH + by * H) * DS + bx * W + (g.split ? 8 * j + 4 * which : SB * j);
	const bool stores = active && (g.split || (int)j < W / SB);
	const int rows = H + g.margin, first_out = g.margin;
	size_t ro1 = ((size_t)f * RH + 16 + by * H - g.before + mvy1) * RS + 16 + bx * W - g.before + mvx1 + pitch * j;
	if (ALIGNED) ro1 &= ~(size_t)3;
	V ring[D], ring1[REFS == 2 ? D : 1];
	uint32_t acc = lane;
	if (mode == 2)
	{
		const V v = reinterpret_cast<const UV<LB> *>(ref + ro)->v;
		acc ^= fold(v);
		for (int y = 0; y < H; ++y)
			if (stores) { V o = v; o.x = acc + y; o.y = acc ^ y; put(dst + dof + (size_t)y * DS, o, SB); }
		return;
	}
#pragma unroll
	for (int i = 0; i < D; ++i)
	{
		ring[i] = reinterpret_cast<const UV<LB> *>(ref + ro + (size_t)(i < rows ? i : rows - 1) * RS)->v;
		if (REFS == 2) ring1[i] = reinterpret_cast<const UV<LB> *>(ref1 + ro1 + (size_t)(i < rows ? i : rows - 1) * RS)->v;
	}
	for (int r0 = 0; r0 < rows; r0 += D)
	{
#pragma unroll
		for (int i = 0; i < D; ++i)
		{
			const int r = r0 + i;
			const V v = ring[i];
			const int nx = r + D < rows ? r + D : rows - 1;
			ring[i] = reinterpret_cast<const UV<LB> *>(ref + ro + (size_t)nx * RS)->v;
			acc = (acc << 1 | acc >> 31) ^ fold(v);
			V v2 = v;
			if (g.plane2_ref)      // the second plane's row of the same window (not ringed: issued and consumed within the row, as a first-order model)
			{
				v2 = reinterpret_cast<const UV<LB> *>(ref + g.plane2_ref + ro + (size_t)(r < rows ? r : rows - 1) * RS)->v;
				if (REFS == 2) acc += fold(reinterpret_cast<const UV<LB> *>(ref1 + g.plane2_ref + ro1 + (size_t)(r < rows ? r : rows - 1) * RS)->v);
			}
			if (REFS == 2)
			{
				acc += fold(ring1[i]);
				ring1[i] = reinterpret_cast<const UV<LB> *>(ref1 + ro1 + (size_t)nx * RS)->v;
			}
			if (mode == 0 && r >= first_out && r - first_out < H && stores)
			{
				V o = v; o.x = acc; o.y = acc ^ v.y; put(dst + dof + (size_t)(r - first_out) * DS, o, SB);
				if (g.plane2_dst) { V o2 = v2; o2.x ^= acc; put(dst + g.plane2_dst + dof + (size_t)(r - first_out) * DS, o2, SB); }
			}
		}
	}
	if (mode == 1 && stores) { V o = ring[0]; o.x = acc; o.y = ~acc; put(dst + dof, o, SB); }
}

static bool g_quiet = false;
static int g_max_frames = 256;
template <int D, int LB, bool ALIGNED, int REFS = 1> float run(int W, int H, int frames, int mode, uint8_t *dst, const uint8_t *ref, int coherent = 0, Geo g = kLuma)
{
	const int FW = g.FW, FH = g.FH, RS = g.RS, RH = g.RH, pitch = g.pitch ? g.pitch : LB, SB = g.store_bytes ? g.store_bytes : LB;
	const int per_row = FW / W, per_frame = per_row * (FH / H);
	const uint32_t n = (uint32_t)per_frame * frames;
	if (W % SB) return 0.f;
	const int L = g.split ? 2 * (W / 8) : g.pitch ? W / pitch : (W + g.margin + LB - 1) / LB, ppw = 64 / L;
	const uint32_t waves = (n + ppw - 1) / ppw, grid = (waves + 3) / 4;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	std::vector<float> ms;
	for (int rep = 0; rep < 7; ++rep)
	{
		// a different third of the buffers each time: nothing comes from the Infinity Cache
		const size_t shift_f = (size_t)(rep % 3) * frames;
		hipEventRecord(e0);
		mix<D, LB, ALIGNED, REFS><<<grid, 256>>>(dst + shift_f * FH * (g.DS ? g.DS : FW), ref + shift_f * RH * RS, ref + (shift_f + 3 * (size_t)g_max_frames) * RH * RS, W, H, L, ppw, n, per_row, per_frame, mode, coherent, g);
		hipEventRecord(e1); hipDeviceSynchronize();
		float t; hipEventElapsedTime(&t, e0, e1);
		if (rep) ms.push_back(t);
	}
	float best = 1e9f, sum = 0;
	for (float t : ms) { best = t < best ? t : best; sum += t; }
	const double ld = (double)(W + g.margin) * (H + g.margin), st = (double)W * H;
	const int refs = g.split ? 2 : REFS;
	const double bytes = (mode == 0 ? refs * ld + st : mode == 1 ? refs * ld : st) * n;
	if (!g_quiet)
	if (!g_quiet && g.plane2_ref) printf("[two planes per lane] ");
	printf("%s %2dx%-2d %3d frames  %d ref  %-11s D=%-2d  %2d-byte lanes %-9s %-11s  mean %.4f ms  min %.4f ms  ->  %.0f GB/s of this mode's bytes\n", g.margin == 7 ? "luma  " : g.pitch ? "chroma (strip deal)" : "chroma (stream deal)", W, H, frames, REFS, coherent ? "one vector" : "vector/PU", D, LB, ALIGNED ? "aligned" : "unaligned",
			mode == 0 ? "loads+stores" : mode == 1 ? "loads only" : "stores only", sum / ms.size(), best, bytes / (sum / ms.size() * 1e-3) / 1e9);
	return sum / ms.size();
}

int main(int argc, char **argv)
{
	const int max_frames = (argc > 1 && argv[1][0] == 'l') ? 64 : 256;
	g_max_frames = max_frames;
	uint8_t *dst, *ref;
	if (hipMalloc(&dst, (size_t)3 * max_frames * kLuma.FH * kLuma.FW + 4096) || hipMalloc(&ref, (size_t)6 * max_frames * kLuma.RH * 2304 + 65536)) { printf("alloc failed\n"); return 1; }
	hipMemset(ref, 3, (size_t)6 * max_frames * kLuma.RH * kLuma.RS);
	const int sizes[][2] = { { 16, 16 }, { 32, 32 }, { 64, 64 }, { 16, 64 }, { 64, 16 }, { 8, 8 } };
	if (argc > 1 && argv[1][0] == 'l')
	{
		// bench.py's figures: the access patterns of its four 64-frame prediction rows WITHOUT arithmetic, on the box and in the run that measures the kernels
		// (luma 16x16 uni: the stream kernel's 8-byte byte-aligned lanes; luma 16x16 bi: pred_hfirst_bi8_kernel's deal -- a lane per 8-column strip and reference,
		// one 16-byte load per row at an 8-byte pitch, 16 rows in flight, 4-byte stores (the stream deal it was measured against until then no
		// longer runs at that size; that one stays as luma16_bi_stream); chroma 8x8 Cb + Cr uni / bi: the strip kernel's word-aligned 16-byte lanes)
		g_quiet = true;
		Geo strip = kChroma; strip.pitch = 8; strip.store_bytes = 8;
		Geo hfirst = kLuma; hfirst.pitch = 8; hfirst.store_bytes = 4; hfirst.split = 1;
		printf("luma16_uni %.4f\n", run<8, 8, false>(16, 16, 64, 0, dst, ref));
		printf("luma16_bi %.4f\n", run<16, 16, false>(16, 16, 64, 0, dst, ref, 0, hfirst));
		printf("luma16_bi_stream %.4f\n", run<8, 8, false, 2>(16, 16, 64, 0, dst, ref));
		printf("chroma8_uni %.4f\n", run<4, 16, true>(8, 8, 128, 0, dst, ref, 0, strip));
		printf("chroma8_bi %.4f\n", run<4, 16, true, 2>(8, 8, 128, 0, dst, ref, 0, strip));
		return 0;
	}
	if (argc > 1 && argv[1][0] == 's')
	{
		// does the reference stride matter?  (the caller chooses it: integration advice if it does)
		for (int rs : { 1952, 1984, 2048, 2112, 2176, 2304 })
			for (int fw : { 1920, 2048 })
			{
				Geo g = kLuma; g.RS = rs; g.DS = fw;
				printf("reference stride %d, destination stride %d:\n", rs, fw);
				run<8, 8, false>(16, 16, 64, 0, dst, ref, 0, g);
				run<8, 8, false>(64, 64, 64, 0, dst, ref, 0, g);
				run<8, 8, false>(16, 16, 64, 1, dst, ref, 0, g);
			}
		return 0;
	}
	if (argc > 1 && argv[1][0] == '2')
	{
		// Cb + Cr: two launches' worth of planes in one grid (what ships: blockIdx.y picks the plane; here 128 planes) against BOTH planes in one lane (64 planes,
		// every lane loads and stores the same PU of plane b as well)
		Geo strip = kChroma; strip.pitch = 8; strip.store_bytes = 8;
		Geo both = strip; both.plane2_ref = (long long)64 * kChroma.RH * kChroma.RS; both.plane2_dst = (long long)64 * kChroma.FH * kChroma.FW;
		const int csizes[][2] = { { 8, 8 }, { 16, 16 }, { 8, 16 }, { 4, 8 } };
		for (auto &sz : csizes)
		{
			run<4, 16, true>(sz[0], sz[1], 128, 0, dst, ref, 0, strip);
			run<4, 16, true>(sz[0], sz[1], 64, 0, dst, ref, 0, both);
			run<4, 16, true, 2>(sz[0], sz[1], 128, 0, dst, ref, 0, strip);
			run<4, 16, true, 2>(sz[0], sz[1], 64, 0, dst, ref, 0, both);
		}
		return 0;
	}
	const bool chroma = argc > 1;
	if (chroma)
	{
		// 960 x 540 planes, 4 taps; "frames" = planes per launch (Cb + Cr of 64 frames = 128).  Strip deal = pred_strip4_kernel's: one lane per 8-column strip,
		// one dword-aligned 16-byte load per lane and row at an 8-byte pitch, every lane stores 8 bytes; stream deal = pred_stream_kernel's 8-byte lanes.
		Geo strip = kChroma; strip.pitch = 8; strip.store_bytes = 8;
		const int csizes[][2] = { { 8, 8 }, { 16, 16 }, { 8, 16 }, { 16, 8 }, { 32, 32 }, { 8, 32 } };
		for (int planes : { 64, 128 })
			for (auto &s : csizes)
				for (int mode = 0; mode < 3; ++mode)
				{
					run<8, 16, true>(s[0], s[1], planes, mode, dst, ref, 0, strip);
					run<4, 16, true>(s[0], s[1], planes, mode, dst, ref, 0, strip);
					run<8, 8, false>(s[0], s[1], planes, mode, dst, ref, 0, kChroma);
					if (mode == 0)
					{
						run<8, 16, true, 2>(s[0], s[1], planes, mode, dst, ref, 0, strip);
						run<4, 16, true, 2>(s[0], s[1], planes, mode, dst, ref, 0, strip);
						run<8, 16, true>(s[0], s[1], planes, mode, dst, ref, 1, strip);
					}
				}
		return 0;
	}
	const bool quick = false;
	for (int frames : { 64, 256 })
		for (auto &s : sizes)
		{
			if (frames == 256 && s[0] != s[1]) continue;
			for (int mode = 0; mode < 3; ++mode)
			{
				run<8, 8, false>(s[0], s[1], frames, mode, dst, ref);
				run<8, 8, true>(s[0], s[1], frames, mode, dst, ref);
				if (quick) continue;
				run<16, 8, false>(s[0], s[1], frames, mode, dst, ref);
				run<8, 16, false>(s[0], s[1], frames, mode, dst, ref);
				run<8, 16, true>(s[0], s[1], frames, mode, dst, ref);
			}
			// one motion vector for the whole batch: neighbouring PUs' window rows are contiguous
			run<8, 8, false>(s[0], s[1], frames, 0, dst, ref, 1);
			run<8, 8, false>(s[0], s[1], frames, 1, dst, ref, 1);
			// two references (bi-prediction)
			run<8, 8, false, 2>(s[0], s[1], frames, 0, dst, ref);
			run<8, 8, true, 2>(s[0], s[1], frames, 0, dst, ref);
			run<8, 8, false, 2>(s[0], s[1], frames, 1, dst, ref);
		}
	return 0;
}
