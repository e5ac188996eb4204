/*
 * multi_gpu_sad.c -- SURVEY.md 8(e) as a C program: ONE host process, one host thread + one stream per device, a batch of
 * independent blocks cut into contiguous ranges (one per device), no exchange between devices, results gathered into one
 * host array.  With G visible gfx950 devices it uses all of them (or argv[1] of them); on a one-GPU box it runs the same
 * code with G = 1, and with argv[2] = "threads" it drives ONE device from several host threads (each its own stream and
 * its own range) -- the library's entry points are thread-safe and hevcasm_hip_set_device() is per thread.
 *
 *   cc -Iinclude examples/multi_gpu_sad.c -Lhevcasm_amd/lib -lhevcasm_hip -lpthread -Wl,-rpath,$PWD/hevcasm_amd/lib -o multi_gpu_sad
 *   ./multi_gpu_sad [ranks] [threads]      exit status 0 = every SAD equals the per-block C table's (77 = no gfx950 device)
 */
#include "sad.h"
#include "hevcasm_hip.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { FRAME_W = 1920, FRAME_H = 1088, BLOCK = 16, RANGE = 7 };

typedef struct
{
	int device, first, count, rc;            /* blocks [first, first + count) */
	const uint8_t *cur, *ref;
	const uint32_t *cur_off, *ref_off;
	int *cost;                                /* the one host array every rank writes its range of */
	char error[160];
} rank_t;

static void *run_rank(void *arg)
{
	rank_t *r = (rank_t *)arg;
	uint8_t *d_cur = 0, *d_ref = 0; uint32_t *d_co = 0, *d_ro = 0; int *d_cost = 0; void *stream = 0;
	const size_t plane = (size_t)FRAME_W * FRAME_H, n = (size_t)r->count;
	int rc = hevcasm_hip_set_device(r->device);
	/* a real codec would upload only the rows its blocks (and their search range) lie in; the frame is small here */
	if (!rc) rc = hevcasm_hip_stream_create(&stream);
	if (!rc) rc = hevcasm_hip_malloc((void **)&d_cur, plane) | hevcasm_hip_malloc((void **)&d_ref, plane) | hevcasm_hip_malloc((void **)&d_co, 4 * n + 4)
	            | hevcasm_hip_malloc((void **)&d_ro, 4 * n + 4) | hevcasm_hip_malloc((void **)&d_cost, 4 * n + 4);
	if (!rc) rc = hevcasm_hip_upload(d_cur, r->cur, plane, stream) | hevcasm_hip_upload(d_ref, r->ref, plane, stream)
	            | hevcasm_hip_upload(d_co, r->cur_off + r->first, 4 * n, stream) | hevcasm_hip_upload(d_ro, r->ref_off + r->first, 4 * n, stream);
	if (!rc) rc = hevcasm_hip_sad_batch(d_cur, FRAME_W, d_ref, FRAME_W, d_co, d_ro, d_cost, HEVCASM_RECT(BLOCK, BLOCK), n, stream);
	if (!rc) rc = hevcasm_hip_download(r->cost + r->first, d_cost, 4 * n, stream);
	if (!rc) rc = hevcasm_hip_stream_synchronize(stream);
	if (rc) snprintf(r->error, sizeof(r->error), "%s", hevcasm_hip_last_error());   /* the error text is per thread too */
	hevcasm_hip_free(d_cur); hevcasm_hip_free(d_ref); hevcasm_hip_free(d_co); hevcasm_hip_free(d_ro); hevcasm_hip_free(d_cost);
	if (stream) hevcasm_hip_stream_destroy(stream);
	r->rc = rc;
	return 0;
}

int main(int argc, char **argv)
{
	const int devices = hevcasm_hip_device_count();
	if (devices < 1 || !hevcasm_hip_available())
	{
		fprintf(stderr, "no gfx950 device: %s\n", hevcasm_hip_last_error());
		return 77;
	}
	const int share_one_device = argc > 2 && !strcmp(argv[2], "threads");
	int ranks = argc > 1 ? atoi(argv[1]) : devices;
	if (ranks < 1) ranks = 1;
	if (!share_one_device && ranks > devices) ranks = devices;
	if (ranks > 64) ranks = 64;

	const int bx = FRAME_W / BLOCK, by = FRAME_H / BLOCK, n = (bx - 2) * (by - 2);
	uint8_t *cur = malloc((size_t)FRAME_W * FRAME_H), *ref = malloc((size_t)FRAME_W * FRAME_H);
	uint32_t *cur_off = malloc(4 * (size_t)n), *ref_off = malloc(4 * (size_t)n);
	int *cpu = malloc(4 * (size_t)n), *gpu = malloc(4 * (size_t)n);
	srand(11);
	for (int i = 0; i < FRAME_W * FRAME_H; ++i) { cur[i] = (uint8_t)rand(); ref[i] = (uint8_t)rand(); }
	int k = 0;
	for (int y = 1; y < by - 1; ++y)
		for (int x = 1; x < bx - 1; ++x, ++k)
		{
			const int mvx = rand() % (2 * RANGE + 1) - RANGE, mvy = rand() % (2 * RANGE + 1) - RANGE;
			cur_off[k] = (uint32_t)(y * BLOCK * FRAME_W + x * BLOCK);
			ref_off[k] = (uint32_t)((y * BLOCK + mvy) * FRAME_W + x * BLOCK + mvx);
		}
	memset(gpu, 0xff, 4 * (size_t)n);

	/* the reference-style loop, for comparison (sad.h:89, :50) */
	hevcasm_table_sad table;
	hevcasm_populate_sad(&table, HEVCASM_C_REF | HEVCASM_C_OPT);
	hevcasm_sad *sad16 = *hevcasm_get_sad(&table, BLOCK, BLOCK);
	for (int i = 0; i < n; ++i) cpu[i] = sad16(cur + cur_off[i], FRAME_W, ref + ref_off[i], FRAME_W, HEVCASM_RECT(BLOCK, BLOCK));

	/* contiguous block ranges: rank r takes blocks [r n / G, (r + 1) n / G) */
	rank_t rank[64];
	pthread_t thread[64];
	for (int r = 0; r < ranks; ++r)
	{
		const int first = (int)((long long)n * r / ranks), last = (int)((long long)n * (r + 1) / ranks);
		rank[r] = (rank_t){ share_one_device ? 0 : r, first, last - first, 0, cur, ref, cur_off, ref_off, gpu, "" };
		pthread_create(&thread[r], 0, run_rank, &rank[r]);
	}
	int failed = 0;
	for (int r = 0; r < ranks; ++r)
	{
		pthread_join(thread[r], 0);
		if (rank[r].rc) { fprintf(stderr, "rank %d (device %d) failed (%d): %s\n", r, rank[r].device, rank[r].rc, rank[r].error); failed = 1; }
	}
	int bad = 0;
	for (int i = 0; i < n; ++i) bad += cpu[i] != gpu[i];
	printf("%d blocks of %dx%d over %d rank(s) on %d device(s)%s: %d differ from the per-block C table\n", n, BLOCK, BLOCK, ranks,
	       share_one_device ? 1 : ranks, share_one_device ? " (threads sharing device 0)" : "", bad);
	return failed || bad != 0;
}
