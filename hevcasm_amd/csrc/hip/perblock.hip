// perblock.hip -- the reference's per-block function types, served by the batched kernels.
//
// Behind the HEVCASM_HIP bit a hevcasm_table_* slot must hold a function with the reference's
// per-block signature taking HOST pointers (sad.h:50, pred_inter.h:56, ...).  Each wrapper here
// copies the block (plus filter margins) into pinned host memory that is mapped into the
// device's address space, launches the batched kernel with n_blocks = 1 reading and writing
// that memory directly over the host link (no separate memcpy calls), waits on the calling
// thread's private stream, and copies the result out.  Same kernels, same arithmetic as the
// batch API; one launch per block, so this is the verification path (hevcasm_test's HIP
// column), not the throughput path.
//
// No fallback: the per-block signatures cannot report errors and returning a CPU-computed value would hide a
// broken GPU path.  A HIP failure here latches the device "unavailable" for the process (shim.hip: later
// hevcasm_populate_*() calls fill their slots from the other bits of the mask), hands the error text to the
// application's failure handler and returns with its outputs unspecified -- or, without a handler, aborts
// (include/hevcasm_hip.h: hevcasm_hip_set_failure_handler).

#include "common.hpp"
#include "../host/perblock.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

constexpr size_t kArenaBytes = 256 * 1024;

struct Arena
{
	uint8_t *host = nullptr;
	uint8_t *dev = nullptr;
	hipStream_t stream = nullptr;
	size_t used = 0;
	bool failed = false;      // this call has failed: every later step of it is skipped, its outputs are left as they are

	bool fail(const char *what, const char *detail)
	{
		char text[256];
		std::snprintf(text, sizeof(text), "hevcasm HIP per-block path: %s failed: %s", what, detail);
		failed = true;
		hevcasm_hip1_failed(text);      // latches; calls the application's handler, or aborts
		return false;
	}
	bool check(hipError_t e, const char *what) { return e == hipSuccess ? true : fail(what, hipGetErrorString(e)); }
	bool check_call(int rc, const char *what) { return rc == 0 ? true : fail(what, hevcasm_hip_last_error()); }
	bool init()
	{
		if (host) return true;
		uint8_t *h = nullptr, *d = nullptr;
		hipStream_t st = nullptr;
		if (!check(hipHostMalloc(reinterpret_cast<void **>(&h), kArenaBytes, hipHostMallocMapped), "hipHostMalloc")) return false;
		if (!check(hipHostGetDevicePointer(reinterpret_cast<void **>(&d), h, 0), "hipHostGetDevicePointer") ||
		    !check(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate"))
		{
			(void)hipHostFree(h);
			return false;
		}
		host = h; dev = d; stream = st;
		return true;
	}
	bool reset() { failed = false; used = 0; return init(); }
	// returns the offset of a 64-byte aligned region (offset 0 after a failure: still inside the arena, the call is abandoned anyway)
	size_t take(size_t bytes)
	{
		const size_t at = (used + 63) & ~size_t(63);
		if (at + bytes > kArenaBytes) { if (!failed) fail("staging", "arena exhausted"); return 0; }
		used = at + bytes;
		return at;
	}
	template <typename T> T *h(size_t off) { return reinterpret_cast<T *>(host + off); }
	template <typename T> T *d(size_t off) { return reinterpret_cast<T *>(dev + off); }
	bool wait() { return check(hipStreamSynchronize(stream), "hipStreamSynchronize"); }
};

thread_local Arena g_arena;

// copy a w x h byte rectangle whose top-left is `p` (row stride `stride`), with `before_x` columns / `before_y` rows
// before it and `after_x` / `after_y` after it, into the arena; returns the offset of the staged top-left sample and the
// staged row stride.  Nothing outside that rectangle is read from the caller's memory.
size_t stage_window(Arena &a, const uint8_t *p, ptrdiff_t stride, int w, int h, int before_x, int after_x, int before_y, int after_y, ptrdiff_t *staged_stride)
{
	const int sw = w + before_x + after_x, sh = h + before_y + after_y;
	const ptrdiff_t ss = (sw + 15) & ~15;
	const size_t off = a.take((size_t)ss * sh);
	for (int y = 0; y < sh; ++y)
		std::memcpy(a.h<uint8_t>(off) + y * ss, p + (ptrdiff_t)(y - before_y) * stride - before_x, (size_t)sw);
	*staged_stride = ss;
	return off + (size_t)before_y * ss + before_x;
}

size_t stage_plane(Arena &a, const uint8_t *p, ptrdiff_t stride, int w, int h, ptrdiff_t *staged_stride)
{
	return stage_window(a, p, stride, w, h, 0, 0, 0, 0, staged_stride);
}

// A prediction reference: the filter margins are staged only in a direction whose fraction is non-zero, which is what the
// reference reads (pred_inter.c:141-228) and what the kernel reads (pred_inter.hip, Footprint).  Both references of a
// bi-prediction get the same staged stride (the batch entry point takes one stride_ref): rows are 64 + 7 bytes at most.
size_t stage_reference(Arena &a, const uint8_t *ref, ptrdiff_t stride_ref, int taps, int w, int h, int xFrac, int yFrac, ptrdiff_t *staged_stride)
{
	const int before = taps / 2 - 1, after = taps / 2;
	const ptrdiff_t ss = 80;
	const int bx = xFrac ? before : 0, ax = xFrac ? after : 0, by = yFrac ? before : 0, ay = yFrac ? after : 0;
	const size_t off = a.take((size_t)ss * (h + by + ay));
	for (int y = 0; y < h + by + ay; ++y)
		std::memcpy(a.h<uint8_t>(off) + y * ss, ref + (ptrdiff_t)(y - by) * stride_ref - bx, (size_t)(w + bx + ax));
	*staged_stride = ss;
	return off + (size_t)by * ss + bx;
}

size_t stage_u32(Arena &a, uint32_t v0, uint32_t v1 = 0, uint32_t v2 = 0, uint32_t v3 = 0)
{
	const size_t off = a.take(16);
	uint32_t *p = a.h<uint32_t>(off);
	p[0] = v0; p[1] = v1; p[2] = v2; p[3] = v3;
	return off;
}


void pred_uni(int taps, uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *ref, ptrdiff_t stride_ref, int w, int h, int xFrac, int yFrac)
{
	Arena &a = g_arena;
	if (!a.reset()) return;
	ptrdiff_t sr;
	const size_t r = stage_reference(a, ref, stride_ref, taps, w, h, xFrac, yFrac, &sr);
	const ptrdiff_t sd = (w + 15) & ~15;
	const size_t d = a.take((size_t)sd * h);
	const size_t offs = stage_u32(a, (uint32_t)d, (uint32_t)r);
	const size_t fr = a.take(16);
	a.h<uint8_t>(fr)[0] = (uint8_t)(xFrac | (yFrac << 4));
	if (a.failed || !a.check_call(hevcasm_hip_pred_uni_8to8_batch(a.dev, sd, a.dev, sr, a.d<uint32_t>(offs), a.d<uint32_t>(offs) + 1, a.d<uint8_t>(fr),
			taps, w, h, 1, a.stream), "pred_uni")) return;
	if (!a.wait()) return;
	for (int y = 0; y < h; ++y) std::memcpy(dst + y * stride_dst, a.h<uint8_t>(d) + y * sd, (size_t)w);
}

void pred_bi(int taps, uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *ref0, const uint8_t *ref1, ptrdiff_t stride_ref,
		int w, int h, int xFrac0, int yFrac0, int xFrac1, int yFrac1)
{
	Arena &a = g_arena;
	if (!a.reset()) return;
	ptrdiff_t sr, sr1;
	const size_t r0 = stage_reference(a, ref0, stride_ref, taps, w, h, xFrac0, yFrac0, &sr);
	const size_t r1 = stage_reference(a, ref1, stride_ref, taps, w, h, xFrac1, yFrac1, &sr1);
	const ptrdiff_t sd = (w + 15) & ~15;
	const size_t d = a.take((size_t)sd * h);
	const size_t offs = stage_u32(a, (uint32_t)d, (uint32_t)r0, (uint32_t)r1);
	const size_t fr = a.take(16);
	a.h<uint8_t>(fr)[0] = (uint8_t)(xFrac0 | (yFrac0 << 4));
	a.h<uint8_t>(fr)[1] = (uint8_t)(xFrac1 | (yFrac1 << 4));
	if (a.failed || !a.check_call(hevcasm_hip_pred_bi_8to8_batch(a.dev, sd, a.dev, a.dev, sr, a.d<uint32_t>(offs), a.d<uint32_t>(offs) + 1, a.d<uint32_t>(offs) + 2,
			a.d<uint8_t>(fr), a.d<uint8_t>(fr) + 1, taps, w, h, 1, a.stream), "pred_bi")) return;
	if (!a.wait()) return;
	for (int y = 0; y < h; ++y) std::memcpy(dst + y * stride_dst, a.h<uint8_t>(d) + y * sd, (size_t)w);
}

void inverse(int trType, int log2n, uint8_t *dst, ptrdiff_t stride_dst, const uint8_t *pred, ptrdiff_t stride_pred, const int16_t *coeffs)
{
	const int n = 1 << log2n;
	Arena &a = g_arena;
	if (!a.reset()) return;
	ptrdiff_t sp;
	const size_t p = stage_plane(a, pred, stride_pred, n, n, &sp);
	const ptrdiff_t sd = (n + 15) & ~15;
	const size_t d = a.take((size_t)sd * n);
	const size_t c = a.take(sizeof(int16_t) * n * n);
	std::memcpy(a.h<int16_t>(c), coeffs, sizeof(int16_t) * n * n);
	const size_t offs = stage_u32(a, (uint32_t)d, (uint32_t)p);
	if (a.failed || !a.check_call(hevcasm_hip_inverse_transform_add_batch(a.dev, sd, a.dev, sp, a.d<uint32_t>(offs), a.d<uint32_t>(offs) + 1, a.d<int16_t>(c),
			trType, log2n, 1, a.stream), "inverse_transform_add")) return;
	if (!a.wait()) return;
	for (int y = 0; y < n; ++y) std::memcpy(dst + y * stride_dst, a.h<uint8_t>(d) + y * sd, (size_t)n);
}

void forward(int trType, int log2n, int16_t *coeffs, const int16_t *src, ptrdiff_t src_stride)
{
	const int n = 1 << log2n;
	Arena &a = g_arena;
	if (!a.reset()) return;
	const size_t s = a.take(sizeof(int16_t) * n * n);
	for (int y = 0; y < n; ++y) std::memcpy(a.h<int16_t>(s) + y * n, src + y * src_stride, sizeof(int16_t) * n);
	const size_t c = a.take(sizeof(int16_t) * n * n);
	const size_t offs = stage_u32(a, 0);
	if (a.failed || !a.check_call(hevcasm_hip_transform_batch(a.d<int16_t>(c), a.d<int16_t>(s), n, a.d<uint32_t>(offs), trType, log2n, 1, a.stream), "transform")) return;
	if (!a.wait()) return;
	std::memcpy(coeffs, a.h<int16_t>(c), sizeof(int16_t) * n * n);
}

}  // namespace

extern "C" {

int hevcasm_hip1_sad(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref, ptrdiff_t stride_ref, uint32_t rect)
{
	const int w = (int)(rect >> 8), h = (int)(rect & 0xff);
	Arena &a = g_arena;
	if (!a.reset()) return 0;
	ptrdiff_t ss, sr;
	const size_t s = stage_plane(a, src, stride_src, w, h, &ss);
	const size_t r = stage_plane(a, ref, stride_ref, w, h, &sr);
	const size_t offs = stage_u32(a, (uint32_t)s, (uint32_t)r);
	const size_t out = a.take(16);
	if (a.failed || !a.check_call(hevcasm_hip_sad_batch(a.dev, ss, a.dev, sr, a.d<uint32_t>(offs), a.d<uint32_t>(offs) + 1, a.d<int>(out), rect, 1, a.stream), "sad")) return 0;
	if (!a.wait()) return 0;
	return *a.h<int>(out);
}

void hevcasm_hip1_sad_multiref_4(const uint8_t *src, ptrdiff_t stride_src, const uint8_t *ref[], ptrdiff_t stride_ref, int sad[], uint32_t rect)
{
	const int w = (int)(rect >> 8), h = (int)(rect & 0xff);
	Arena &a = g_arena;
	if (!a.reset()) return;
	ptrdiff_t ss, sr = 0;
	const size_t s = stage_plane(a, src, stride_src, w, h, &ss);
	size_t r[4];
	for (int k = 0; k < 4; ++k) r[k] = stage_plane(a, ref[k], stride_ref, w, h, &sr);
	const size_t so = stage_u32(a, (uint32_t)s);
	const size_t ro = stage_u32(a, (uint32_t)r[0], (uint32_t)r[1], (uint32_t)r[2], (uint32_t)r[3]);
	const size_t out = a.take(16);
	if (a.failed || !a.check_call(hevcasm_hip_sad_multiref_batch(a.dev, ss, a.dev, sr, a.d<uint32_t>(so), a.d<uint32_t>(ro), a.d<int>(out), rect, 1, a.stream), "sad_multiref")) return;
	if (!a.wait()) return;
	std::memcpy(sad, a.h<int>(out), 4 * sizeof(int));
}

int hevcasm_hip1_ssd(const uint8_t *srcA, ptrdiff_t stride_srcA, const uint8_t *srcB, ptrdiff_t stride_srcB, int w, int h)
{
	Arena &a = g_arena;
	if (!a.reset()) return 0;
	ptrdiff_t sa, sb;
	const size_t pa = stage_plane(a, srcA, stride_srcA, w, h, &sa);
	const size_t pb = stage_plane(a, srcB, stride_srcB, w, h, &sb);
	const size_t offs = stage_u32(a, (uint32_t)pa, (uint32_t)pb);
	const size_t out = a.take(16);
	if (a.failed || !a.check_call(hevcasm_hip_ssd_batch(a.dev, sa, a.dev, sb, a.d<uint32_t>(offs), a.d<uint32_t>(offs) + 1, a.d<int>(out), w, h, 1, a.stream), "ssd")) return 0;
	if (!a.wait()) return 0;
	return *a.h<int>(out);
}

static int satd1(int log2n, const uint8_t *srcA, ptrdiff_t stride_srcA, const uint8_t *srcB, ptrdiff_t stride_srcB)
{
	const int n = 1 << log2n;
	Arena &a = g_arena;
	if (!a.reset()) return 0;
	ptrdiff_t sa, sb;
	const size_t pa = stage_plane(a, srcA, stride_srcA, n, n, &sa);
	const size_t pb = stage_plane(a, srcB, stride_srcB, n, n, &sb);
	const size_t offs = stage_u32(a, (uint32_t)pa, (uint32_t)pb);
	const size_t out = a.take(16);
	if (a.failed || !a.check_call(hevcasm_hip_hadamard_satd_batch(a.dev, sa, a.dev, sb, a.d<uint32_t>(offs), a.d<uint32_t>(offs) + 1, a.d<int>(out), log2n, 1, a.stream), "hadamard_satd")) return 0;
	if (!a.wait()) return 0;
	return *a.h<int>(out);
}
int hevcasm_hip1_satd_2(const uint8_t *a, ptrdiff_t sa, const uint8_t *b, ptrdiff_t sb) { return satd1(1, a, sa, b, sb); }
int hevcasm_hip1_satd_4(const uint8_t *a, ptrdiff_t sa, const uint8_t *b, ptrdiff_t sb) { return satd1(2, a, sa, b, sb); }
int hevcasm_hip1_satd_8(const uint8_t *a, ptrdiff_t sa, const uint8_t *b, ptrdiff_t sb) { return satd1(3, a, sa, b, sb); }

void hevcasm_hip1_pred_uni_8tap(uint8_t *dst, ptrdiff_t sd, const uint8_t *ref, ptrdiff_t sr, int w, int h, int xf, int yf) { pred_uni(8, dst, sd, ref, sr, w, h, xf, yf); }
void hevcasm_hip1_pred_uni_4tap(uint8_t *dst, ptrdiff_t sd, const uint8_t *ref, ptrdiff_t sr, int w, int h, int xf, int yf) { pred_uni(4, dst, sd, ref, sr, w, h, xf, yf); }
void hevcasm_hip1_pred_bi_8tap(uint8_t *dst, ptrdiff_t sd, const uint8_t *r0, const uint8_t *r1, ptrdiff_t sr, int w, int h, int x0, int y0, int x1, int y1) { pred_bi(8, dst, sd, r0, r1, sr, w, h, x0, y0, x1, y1); }
void hevcasm_hip1_pred_bi_4tap(uint8_t *dst, ptrdiff_t sd, const uint8_t *r0, const uint8_t *r1, ptrdiff_t sr, int w, int h, int x0, int y0, int x1, int y1) { pred_bi(4, dst, sd, r0, r1, sr, w, h, x0, y0, x1, y1); }

void hevcasm_hip1_idst_4(uint8_t *d, ptrdiff_t sd, const uint8_t *p, ptrdiff_t sp, const int16_t *c) { inverse(1, 2, d, sd, p, sp, c); }
void hevcasm_hip1_idct_4(uint8_t *d, ptrdiff_t sd, const uint8_t *p, ptrdiff_t sp, const int16_t *c) { inverse(0, 2, d, sd, p, sp, c); }
void hevcasm_hip1_idct_8(uint8_t *d, ptrdiff_t sd, const uint8_t *p, ptrdiff_t sp, const int16_t *c) { inverse(0, 3, d, sd, p, sp, c); }
void hevcasm_hip1_idct_16(uint8_t *d, ptrdiff_t sd, const uint8_t *p, ptrdiff_t sp, const int16_t *c) { inverse(0, 4, d, sd, p, sp, c); }
void hevcasm_hip1_idct_32(uint8_t *d, ptrdiff_t sd, const uint8_t *p, ptrdiff_t sp, const int16_t *c) { inverse(0, 5, d, sd, p, sp, c); }

void hevcasm_hip1_dst_4(int16_t *c, const int16_t *s, ptrdiff_t st) { forward(1, 2, c, s, st); }
void hevcasm_hip1_dct_4(int16_t *c, const int16_t *s, ptrdiff_t st) { forward(0, 2, c, s, st); }
void hevcasm_hip1_dct_8(int16_t *c, const int16_t *s, ptrdiff_t st) { forward(0, 3, c, s, st); }
void hevcasm_hip1_dct_16(int16_t *c, const int16_t *s, ptrdiff_t st) { forward(0, 4, c, s, st); }
void hevcasm_hip1_dct_32(int16_t *c, const int16_t *s, ptrdiff_t st) { forward(0, 5, c, s, st); }

void hevcasm_hip1_quantize_inverse(int16_t *dst, const int16_t *src, int scale, int shift, int n)
{
	Arena &a = g_arena;
	for (int done = 0; done < n;)
	{
		const int m = n - done < 32768 ? n - done : 32768;
		if (!a.reset()) return;
		const size_t s = a.take(sizeof(int16_t) * m), d = a.take(sizeof(int16_t) * m);
		std::memcpy(a.h<int16_t>(s), src + done, sizeof(int16_t) * m);
		if (a.failed || !a.check_call(hevcasm_hip_quantize_inverse(a.d<int16_t>(d), a.d<int16_t>(s), scale, shift, (size_t)m, a.stream), "quantize_inverse")) return;
		if (!a.wait()) return;
		std::memcpy(dst + done, a.h<int16_t>(d), sizeof(int16_t) * m);
		done += m;
	}
}

int hevcasm_hip1_quantize(int16_t *dst, const int16_t *src, int scale, int shift, int offset, int n)
{
	// the batch kernel takes whole blocks of 16, 64, 256 or 1024 coefficients: use the largest that divides n
	Arena &a = g_arena;
	int cbf = 0;
	for (int done = 0; done < n;)
	{
		const int left = n - done;
		const int npb = left >= 1024 ? 1024 : left >= 256 ? 256 : left >= 64 ? 64 : 16;
		if (!a.reset()) return 0;
		if (left < 16) { a.fail("quantize", "n must be a multiple of 16"); return 0; }
		const size_t s = a.take(sizeof(int16_t) * npb), d = a.take(sizeof(int16_t) * npb), c = a.take(16);
		std::memcpy(a.h<int16_t>(s), src + done, sizeof(int16_t) * npb);
		if (a.failed || !a.check_call(hevcasm_hip_quantize_batch(a.d<int16_t>(d), a.d<int16_t>(s), scale, shift, offset, npb, 1, a.d<int>(c), a.stream), "quantize")) return 0;
		if (!a.wait()) return 0;
		std::memcpy(dst + done, a.h<int16_t>(d), sizeof(int16_t) * npb);
		cbf |= *a.h<int>(c);
		done += npb;
	}
	return cbf;
}

void hevcasm_hip1_quantize_reconstruct(uint8_t *rec, ptrdiff_t stride_rec, const uint8_t *pred, ptrdiff_t stride_pred, const int16_t *res, int n)
{
	Arena &a = g_arena;
	if (!a.reset()) return;
	ptrdiff_t sp;
	const size_t p = stage_plane(a, pred, stride_pred, n, n, &sp);
	const ptrdiff_t sd = (n + 15) & ~15;
	const size_t d = a.take((size_t)sd * n);
	const size_t r = a.take(sizeof(int16_t) * n * n);
	std::memcpy(a.h<int16_t>(r), res, sizeof(int16_t) * n * n);
	const size_t offs = stage_u32(a, (uint32_t)d, (uint32_t)p);
	if (a.failed || !a.check_call(hevcasm_hip_quantize_reconstruct_batch(a.dev, sd, a.dev, sp, a.d<uint32_t>(offs), a.d<uint32_t>(offs) + 1, a.d<int16_t>(r), n, 1, a.stream), "quantize_reconstruct")) return;
	if (!a.wait()) return;
	for (int y = 0; y < n; ++y) std::memcpy(rec + y * stride_rec, a.h<uint8_t>(d) + y * sd, (size_t)n);
}

void hevcasm_hip1_pred_intra_dc(uint8_t *dst, const uint8_t *neighbours, int intraPredMode, hevcasm_pred_intra_packed packed)
{
	const int n = 1 << ((packed >> 1) & 0x7f);
	Arena &a = g_arena;
	if (!a.reset()) return;
	// stage bytes [64 - n, 64 + n) of the caller's array at the same indices of a 128-byte record
	const size_t nb = a.take(128), d = a.take((size_t)n * n);
	std::memcpy(a.h<uint8_t>(nb) + 64 - n, neighbours + 64 - n, 2 * (size_t)n);
	if (a.failed || !a.check_call(hevcasm_hip_pred_intra_batch(a.d<uint8_t>(d), a.d<uint8_t>(nb), 128, intraPredMode, packed, 1, a.stream), "pred_intra")) return;
	if (!a.wait()) return;
	std::memcpy(dst, a.h<uint8_t>(d), (size_t)n * n);
}

// planar / angular: the record's 4N + 1 samples (bytes [64 - 2N, 64 + 2N) and the corner at [128])
void hevcasm_hip1_pred_intra_any(uint8_t *dst, const uint8_t *neighbours, int intraPredMode, hevcasm_pred_intra_packed packed)
{
	const int n = 1 << ((packed >> 1) & 0x7f);
	Arena &a = g_arena;
	if (!a.reset()) return;
	const size_t nb = a.take(144), d = a.take((size_t)n * n);
	std::memcpy(a.h<uint8_t>(nb) + 64 - 2 * n, neighbours + 64 - 2 * n, 4 * (size_t)n);
	a.h<uint8_t>(nb)[128] = neighbours[128];
	if (a.failed || !a.check_call(hevcasm_hip_pred_intra_batch(a.d<uint8_t>(d), a.d<uint8_t>(nb), 144, intraPredMode, packed, 1, a.stream), "pred_intra")) return;
	if (!a.wait()) return;
	std::memcpy(dst, a.h<uint8_t>(d), (size_t)n * n);
}

int hevcasm_hip1_ssd_linear(const uint8_t *src0, const uint8_t *src1, int size)
{
	Arena &a = g_arena;
	unsigned total = 0;
	// runs longer than the staging arena go through it in pieces; 32-bit sums add up exactly (mod 2^32)
	const int piece_max = 96 * 1024;
	for (int done = 0; done < size || (size == 0 && done == 0); done += piece_max)
	{
		const int piece = size - done < piece_max ? size - done : piece_max;
		if (!a.reset()) return 0;
		const size_t p0 = a.take((size_t)piece + 16), p1 = a.take((size_t)piece + 16);
		std::memcpy(a.h<uint8_t>(p0), src0 + done, (size_t)piece);
		std::memcpy(a.h<uint8_t>(p1), src1 + done, (size_t)piece);
		const size_t offs = stage_u32(a, (uint32_t)p0, (uint32_t)p1);
		const size_t out = a.take(16);
		if (a.failed || !a.check_call(hevcasm_hip_ssd_linear_batch(a.dev, a.dev, a.d<uint32_t>(offs), a.d<uint32_t>(offs) + 1, a.d<int>(out), piece, 1, a.stream), "ssd_linear")) return 0;
		if (!a.wait()) return 0;
		total += (unsigned)*a.h<int>(out);
		if (size == 0) break;
	}
	return (int)total;
}

}  // extern "C"
