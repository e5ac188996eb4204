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
 *     hevcasm_host_bench <instruction-set mask, hex> <ms per combination> <blocks> [seed]
 */
#include "hevcasm.h"
#include "sad.h"

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

int main(int argc, char **argv)
{
	if (argc < 4) { fprintf(stderr, "usage: %s <mask hex> <ms> <blocks> [seed]\n", argv[0]); return 2; }
	const hevcasm_instruction_set mask = (hevcasm_instruction_set)strtoul(argv[1], 0, 16);
	const int ms = atoi(argv[2]), blocks = atoi(argv[3]);
	rng_state = argc > 4 ? strtoull(argv[4], 0, 10) : 0x48455643ull;
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
