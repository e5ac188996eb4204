/*
 * batch_sad.c -- what INTEGRATION.md section 3 describes, as a program a codec maintainer can compile:
 * the reference-style per-block loop (function-pointer table, C path) next to the batched GPU call, on
 * the same blocks, results compared.
 *
 *   cc -Iinclude examples/batch_sad.c -Lhevcasm_amd/lib -lhevcasm_hip -Wl,-rpath,$PWD/hevcasm_amd/lib -o batch_sad
 *   ./batch_sad            exit status 0 = every SAD identical (77 = no gfx950 device visible)
 */
#include "sad.h"
#include "hevcasm_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { FRAME_W = 640, FRAME_H = 384, BLOCK = 16, RANGE = 7 };

int main(void)
{
	const int bx = FRAME_W / BLOCK, by = FRAME_H / BLOCK, n = (bx - 2) * (by - 2);
	uint8_t *cur = malloc(FRAME_W * FRAME_H), *ref = malloc(FRAME_W * FRAME_H);
	uint32_t *cur_off = malloc(4 * n), *ref_off = malloc(4 * n);
	int *cpu = malloc(4 * n), *gpu = malloc(4 * n);
	srand(7);
	for (int i = 0; i < FRAME_W * FRAME_H; ++i) { cur[i] = (uint8_t)rand(); ref[i] = (uint8_t)rand(); }
	/* interior blocks, each with its own motion vector in [-RANGE, RANGE]^2 */
	int k = 0;
	for (int y = 1; y < by - 1; ++y)
		for (int x = 1; x < bx - 1; ++x, ++k)
		{
			const int mvx = rand() % (2 * RANGE + 1) - RANGE, mvy = rand() % (2 * RANGE + 1) - RANGE;
			cur_off[k] = (uint32_t)(y * BLOCK * FRAME_W + x * BLOCK);
			ref_off[k] = (uint32_t)((y * BLOCK + mvy) * FRAME_W + x * BLOCK + mvx);
		}

	/* 1. the codec's loop as it is today: one table call per block (reference sad.h:89, :50) */
	hevcasm_table_sad table;
	hevcasm_populate_sad(&table, HEVCASM_C_REF | HEVCASM_C_OPT);
	hevcasm_sad *sad16 = *hevcasm_get_sad(&table, BLOCK, BLOCK);
	for (int i = 0; i < n; ++i)
		cpu[i] = sad16(cur + cur_off[i], FRAME_W, ref + ref_off[i], FRAME_W, HEVCASM_RECT(BLOCK, BLOCK));

	/* 2. the same loop as one batched call: frames resident on the device, descriptors in, costs out */
	if (!hevcasm_hip_available())
	{
		fprintf(stderr, "no gfx950 device: %s\n", hevcasm_hip_last_error());
		return 77;
	}
	uint8_t *d_cur, *d_ref; uint32_t *d_cur_off, *d_ref_off; int *d_cost; void *stream;
	int rc = hevcasm_hip_stream_create(&stream);
	rc |= hevcasm_hip_malloc((void **)&d_cur, FRAME_W * FRAME_H);
	rc |= hevcasm_hip_malloc((void **)&d_ref, FRAME_W * FRAME_H);
	rc |= hevcasm_hip_malloc((void **)&d_cur_off, 4 * (size_t)n);
	rc |= hevcasm_hip_malloc((void **)&d_ref_off, 4 * (size_t)n);
	rc |= hevcasm_hip_malloc((void **)&d_cost, 4 * (size_t)n);
	rc |= hevcasm_hip_upload(d_cur, cur, FRAME_W * FRAME_H, stream);
	rc |= hevcasm_hip_upload(d_ref, ref, FRAME_W * FRAME_H, stream);
	rc |= hevcasm_hip_upload(d_cur_off, cur_off, 4 * (size_t)n, stream);
	rc |= hevcasm_hip_upload(d_ref_off, ref_off, 4 * (size_t)n, stream);
	if (!rc) rc = hevcasm_hip_sad_batch(d_cur, FRAME_W, d_ref, FRAME_W, d_cur_off, d_ref_off, d_cost, HEVCASM_RECT(BLOCK, BLOCK), (size_t)n, stream);
	if (!rc) rc = hevcasm_hip_download(gpu, d_cost, 4 * (size_t)n, stream);
	if (!rc) rc = hevcasm_hip_stream_synchronize(stream);
	if (rc)
	{
		fprintf(stderr, "hevcasm_hip call failed (%d): %s\n", rc, hevcasm_hip_last_error());   /* no CPU fallback exists */
		return 1;
	}
	int bad = 0;
	for (int i = 0; i < n; ++i) bad += cpu[i] != gpu[i];
	printf("%d blocks of %dx%d: %d differ between the per-block C table and hevcasm_hip_sad_batch\n", n, BLOCK, BLOCK, bad);
	hevcasm_hip_free(d_cur); hevcasm_hip_free(d_ref); hevcasm_hip_free(d_cur_off); hevcasm_hip_free(d_ref_off); hevcasm_hip_free(d_cost);
	hevcasm_hip_stream_destroy(stream);
	return bad != 0;
}
