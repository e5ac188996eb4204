/*
 * bench_host.c -- times this library's host table entries on the SAD mix of BASELINE.json config #2.
 *
 * The per-block calling pattern of a codec (populate a table once, call the slot per block: SURVEY.md 3.2) over a
 * synthetic frame pair laid out like bench.py's workload: source block i = tile i of a 1920 x 1152 frame stack,
 * candidate = the same tile of a second stack (16 / 8 pixels of border) displaced by up to +-7 pixels.  For each of the
 * 20 block sizes >= 8x8, one reference and four, it loops over `blocks` blocks until `ms` milliseconds have passed and
 * prints "w h ways blocks seconds".  bench.py runs one of these per host core for its host-SIMD baseline column
 * (mask = the x86 bits: the intrinsics kernels of sad_x86.c; mask = 1: the plain C path).
 *
 *     hevcasm_host_bench <instruction-set mask, hex> <ms per combination> <blocks> [seed] [family]
 *
 * family (round 4; default "sad"): "pred" = config #3's PUs through the prediction tables -- luma 16x16 8-tap and chroma 8x8 4-tap,
 * uni and bi, integer vectors in +-8 plus every fractional phase, 1920x1080 / 960x540 frames with 16 pixels of border -- printing
 * "pred_uni|pred_bi taps w h blocks seconds"; "inv" = config #4's blocks through the inverse-transform table -- 8x8 / 16x16 / 32x32,
 * sparse low-magnitude levels, plain and with hevcasm_quantize_inverse (scale 51, shift 14) in front -- printing
 * "inv log2 fused blocks seconds".
 */
#include "hevcasm.h"
#include "sad.h"
#include "pred_inter.h"
#include "residual_decode.h"
#include "quantize.h"
#include <string.h>

#include <stdio.h>
#include <stdlib.h>
#include <time.h>

static double now(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static uint64_t rng_state;
static uint32_t rng(void)     /* splitmix64 */
{
	uint64_t z = (rng_state += 0x9e3779b97f4a7c15ull);
	z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
	z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
	return (uint32_t)((z ^ (z >> 31)) >> 16);
}

static void fill_bytes(uint8_t *p, size_t n)
{
	size_t i = 0;
	for (; i + 4 <= n; i += 4) { const uint32_t v = rng(); memcpy(p + i, &v, 4); }
	for (; i < n; ++i) p[i] = (uint8_t)rng();
}

/* config #3 through the host prediction tables */
static int bench_pred(hevcasm_instruction_set mask, int ms, int blocks)
{
	hevcasm_table_pred_uni_8to8 tu;
	hevcasm_table_pred_bi_8to8 tb;
	hevcasm_populate_pred_uni_8to8(&tu, mask);
	hevcasm_populate_pred_bi_8to8(&tb, mask);
	for (int taps = 8; taps >= 4; taps -= 4)
	{
		const int w = taps == 8 ? 16 : 8, h = w, FW = taps == 8 ? 1920 : 960, FH = taps == 8 ? 1080 : 540, B = 16, phases = taps == 8 ? 4 : 8;
		const int per_row = FW / w, per_frame = per_row * (FH / h);
		const int frames = (blocks + per_frame - 1) / per_frame;
		const ptrdiff_t sd = FW, sr = FW + 2 * B;
		const size_t ref_frame = (size_t)sr * (FH + 2 * B), ref_bytes = frames * ref_frame, dst_bytes = (size_t)frames * FW * FH;
		uint8_t *dst = malloc(dst_bytes), *ref0 = malloc(ref_bytes), *ref1 = malloc(ref_bytes);
		size_t *dof = malloc(sizeof(size_t) * blocks), *rof = malloc(sizeof(size_t) * 2 * blocks);
		uint8_t *fr = malloc(4 * (size_t)blocks);
		if (!dst || !ref0 || !ref1 || !dof || !rof || !fr) { fprintf(stderr, "out of memory\n"); return 1; }
		fill_bytes(ref0, ref_bytes); fill_bytes(ref1, ref_bytes); memset(dst, 0, dst_bytes);
		for (int i = 0; i < blocks; ++i)
		{
			const int f = i / per_frame, k = i % per_frame, bx = (k % per_row) * w, by = (k / per_row) * h;
			dof[i] = (size_t)f * FW * FH + (size_t)by * sd + bx;
			for (int r = 0; r < 2; ++r)
			{
				const int dx = (int)(rng() % 17) - 8, dy = (int)(rng() % 17) - 8;
				rof[2 * i + r] = f * ref_frame + (size_t)(by + B + dy) * sr + (size_t)(bx + B + dx);
				fr[4 * i + 2 * r] = (uint8_t)(rng() % phases); fr[4 * i + 2 * r + 1] = (uint8_t)(rng() % phases);
			}
		}
		for (int bi = 0; bi < 2; ++bi)
		{
			long long done = 0;
			const double t0 = now();
			double t = t0;
			do
			{
				for (int i = 0; i < blocks; ++i)
				{
					const int xa = fr[4 * i], ya = fr[4 * i + 1], xb = fr[4 * i + 2], yb = fr[4 * i + 3];
					if (bi)
					{
						hevcasm_pred_bi_8to8 *f = *hevcasm_get_pred_bi_8to8(&tb, taps, w, h, xa, ya, xb, yb);
						if (!f) { fprintf(stderr, "no pred_bi entry under mask %x\n", (unsigned)mask); return 1; }
						f(dst + dof[i], sd, ref0 + rof[2 * i], ref1 + rof[2 * i + 1], sr, w, h, xa, ya, xb, yb);
					}
					else
					{
						hevcasm_pred_uni_8to8 *f = *hevcasm_get_pred_uni_8to8(&tu, taps, w, h, xa, ya);
						if (!f) { fprintf(stderr, "no pred_uni entry under mask %x\n", (unsigned)mask); return 1; }
						f(dst + dof[i], sd, ref0 + rof[2 * i], sr, w, h, xa, ya);
					}
				}
				done += blocks;
				t = now();
			} while ((t - t0) * 1e3 < ms);
			printf("%s %d %d %d %lld %.6f\n", bi ? "pred_bi" : "pred_uni", taps, w, h, done, t - t0);
		}
		free(dst); free(ref0); free(ref1); free(dof); free(rof); free(fr);
	}
	return 0;
}

/* config #4 through the host inverse-transform / dequantiser tables */
static int bench_inv(hevcasm_instruction_set mask, int ms, int blocks)
{
	hevcasm_table_inverse_transform_add ti;
	hevcasm_table_quantize_inverse tq;
	hevcasm_populate_inverse_transform_add(&ti, mask, 1);
	hevcasm_populate_quantize_inverse(&tq, mask);
	for (int log2n = 3; log2n <= 5; ++log2n)
	{
		const int n = 1 << log2n, FW = 1920, FH = 1152, per_row = FW / n, per_frame = per_row * (FH / n);
		const int nb = ((blocks * 64) >> (2 * log2n)) > 0 ? (blocks * 64) >> (2 * log2n) : 1;     /* `blocks` counts 8x8 blocks: equal coefficient counts per size */
		const int frames = (nb + per_frame - 1) / per_frame;
		const ptrdiff_t sp = FW, sd = FW + 32;
		const size_t pred_bytes = (size_t)frames * FW * FH, dst_bytes = (size_t)frames * sd * FH;
		uint8_t *pred = malloc(pred_bytes), *dst = malloc(dst_bytes);
		int16_t *coeffs = malloc(sizeof(int16_t) * (size_t)nb * n * n);
		if (!pred || !dst || !coeffs) { fprintf(stderr, "out of memory\n"); return 1; }
		fill_bytes(pred, pred_bytes); memset(dst, 0, dst_bytes);
		for (size_t i = 0; i < (size_t)nb * n * n; ++i)
		{
			const uint32_t v = rng();
			int mag = 1;
			for (uint32_t b = v >> 8; (b & 3) != 0 && mag < 40; b >>= 2) ++mag;      /* geometric, p = 1/4 */
			coeffs[i] = (int16_t)((v & 0xff) < 179 ? 0 : (v & 0x80000000u) ? -mag : mag);   /* 70 % zeros */
		}
		hevcasm_inverse_transform_add *f = *hevcasm_get_inverse_transform_add(&ti, 0, log2n);
		hevcasm_quantize_inverse *dq = *hevcasm_get_quantize_inverse(&tq);
		if (!f || !dq) { fprintf(stderr, "no inverse-transform entry under mask %x\n", (unsigned)mask); return 1; }
		for (int fused = 0; fused < 2; ++fused)
		{
			long long done = 0;
			const double t0 = now();
			double t = t0;
			do
			{
				for (int i = 0; i < nb; ++i)
				{
					HEVCASM_ALIGN(32, int16_t, tmp[32 * 32]);
					const int fi = i / per_frame, k = i % per_frame, bx = (k % per_row) * n, by = (k / per_row) * n;
					const int16_t *c = coeffs + (size_t)i * n * n;
					if (fused) { dq(tmp, c, 51, 14, n * n); c = tmp; }
					f(dst + ((size_t)fi * FH + by) * sd + bx, sd, pred + ((size_t)fi * FH + by) * sp + bx, sp, c);
				}
				done += nb;
				t = now();
			} while ((t - t0) * 1e3 < ms);
			printf("inv %d %d %lld %.6f\n", log2n, fused, done, t - t0);
		}
		free(pred); free(dst); free(coeffs);
	}
	return 0;
}

int main(int argc, char **argv)
{
	if (argc < 4) { fprintf(stderr, "usage: %s <mask hex> <ms> <blocks> [seed] [sad|pred|inv]\n", argv[0]); return 2; }
	const hevcasm_instruction_set mask = (hevcasm_instruction_set)strtoul(argv[1], 0, 16);
	const int ms = atoi(argv[2]), blocks = atoi(argv[3]);
	rng_state = argc > 4 ? strtoull(argv[4], 0, 10) : 0x48455643ull;
	if (argc > 5 && !strcmp(argv[5], "pred")) return bench_pred(mask, ms, blocks);
	if (argc > 5 && !strcmp(argv[5], "inv")) return bench_inv(mask, ms, blocks);
	enum { FW = 1920, FH = 1152, BX = 16, BY = 8 };
	const ptrdiff_t ss = FW, sr = FW + 2 * BX;
	static const unsigned char sizes[20][2] = {
		{ 64, 64 }, { 64, 48 }, { 64, 32 }, { 64, 16 }, { 48, 64 }, { 32, 64 }, { 32, 32 }, { 32, 24 }, { 32, 16 }, { 32, 8 },
		{ 24, 32 }, { 16, 64 }, { 16, 32 }, { 16, 16 }, { 16, 12 }, { 16, 8 }, { 12, 16 }, { 8, 32 }, { 8, 16 }, { 8, 8 } };

	hevcasm_table_sad t1;
	hevcasm_table_sad_multiref t4;
	hevcasm_populate_sad(&t1, mask);
	hevcasm_populate_sad_multiref(&t4, mask);

	for (int s = 0; s < 20; ++s)
	{
		const int w = sizes[s][0], h = sizes[s][1];
		const int per_row = FW / w, per_frame = per_row * (FH / h);
		const int frames = (blocks + per_frame - 1) / per_frame;
		const size_t src_bytes = (size_t)frames * FW * FH, ref_frame = (size_t)sr * (FH + 2 * BY), ref_bytes = frames * ref_frame;
		uint8_t *src = malloc(src_bytes), *ref = malloc(ref_bytes);
		size_t *so = malloc(sizeof(size_t) * blocks), *ro = malloc(sizeof(size_t) * 4 * blocks);
		if (!src || !ref || !so || !ro) { fprintf(stderr, "out of memory\n"); return 1; }
		for (size_t i = 0; i < src_bytes; i += 4) { const uint32_t v = rng(); src[i] = (uint8_t)v; src[i + 1] = (uint8_t)(v >> 8); src[i + 2] = (uint8_t)(v >> 16); src[i + 3] = (uint8_t)(v >> 24); }
		for (size_t i = 0; i + 4 <= ref_bytes; i += 4) { const uint32_t v = rng(); ref[i] = (uint8_t)v; ref[i + 1] = (uint8_t)(v >> 8); ref[i + 2] = (uint8_t)(v >> 16); ref[i + 3] = (uint8_t)(v >> 24); }
		for (int i = 0; i < blocks; ++i)
		{
			const int f = i / per_frame, k = i % per_frame, bx = (k % per_row) * w, by = (k / per_row) * h;
			so[i] = (size_t)f * FW * FH + (size_t)by * ss + bx;
			for (int c = 0; c < 4; ++c)
			{
				const int dx = (int)(rng() % 15) - 7, dy = (int)(rng() % 15) - 7;
				ro[4 * i + c] = f * ref_frame + (size_t)(by + BY + dy) * sr + (size_t)(bx + BX + dx);
			}
		}
		hevcasm_sad *f1 = *hevcasm_get_sad(&t1, w, h);
		hevcasm_sad_multiref *f4 = *hevcasm_get_sad_multiref(&t4, 4, w, h);
		if (!f1 || !f4) { fprintf(stderr, "no table entry for %dx%d under mask %x\n", w, h, (unsigned)mask); return 1; }
		const uint32_t rect = HEVCASM_RECT(w, h);
		volatile long long sink = 0;
		for (int ways = 1; ways <= 4; ways += 3)
		{
			long long done = 0;
			const double t0 = now();
			double t = t0;
			do
			{
				long long acc = 0;
				if (ways == 1)
					for (int i = 0; i < blocks; ++i) acc += f1(src + so[i], ss, ref + ro[4 * i], sr, rect);
				else
					for (int i = 0; i < blocks; ++i)
					{
						const uint8_t *r[4] = { ref + ro[4 * i], ref + ro[4 * i + 1], ref + ro[4 * i + 2], ref + ro[4 * i + 3] };
						int sad[4];
						f4(src + so[i], ss, r, sr, sad, rect);
						acc += sad[0] + sad[1] + sad[2] + sad[3];
					}
				sink += acc;
				done += blocks;
				t = now();
			} while ((t - t0) * 1e3 < ms);
			printf("%d %d %d %lld %.6f\n", w, h, ways, done, t - t0);
		}
		free(src); free(ref); free(so); free(ro);
	}
	return 0;
}
