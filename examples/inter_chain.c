/*
 * inter_chain.c -- a C codec's inter loop for one frame cut into prediction units of MANY sizes, twice: PU by PU through the reference's
 * function-pointer tables (sad.h:50, pred_inter.h:56 -- the C entries of this library: full search, quarter-pel refinement, prediction, cost), and as
 * FOUR device calls for the whole frame, one per stage, whatever the sizes ("one launch per frame"):
 *
 *     hevcasm_hip_me_full_search_runs  ->  hevcasm_hip_subpel_refine_runs  ->  hevcasm_hip_pred_uni_8to8_runs  ->  hevcasm_hip_sad_runs
 *
 * Between the stages the codec turns vectors into offsets on the host (it owns the mode decision; a few kilobytes travel), everything else stays on the
 * device.  Vectors, refined vectors, the predicted plane and every PU's cost must agree bit for bit.
 *
 *   cc -Iinclude examples/inter_chain.c -Lhevcasm_amd/lib -lhevcasm_hip -Wl,-rpath,$PWD/hevcasm_amd/lib -o inter_chain
 *   ./inter_chain          exit status 0 = identical (77 = no gfx950 device visible)
 */
#include "sad.h"
#include "pred_inter.h"
#include "hevcasm_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { W = 320, H = 192, PAD = 24, RANGE = 6, SR = W + 2 * PAD, RH = H + 2 * PAD, MAX_PUS = (W / 4) * (H / 4) };

typedef struct { int x, y, w, h; } pu_t;
static pu_t pus[MAX_PUS];
static int n_pus;

static unsigned hash3(unsigned a, unsigned b, unsigned c) { unsigned h = a * 2654435761u ^ b * 40503u ^ c * 69069u; h ^= h >> 13; h *= 0x5bd1e995u; return h ^ (h >> 15); }

/* one coding unit -> its prediction units by an HEVC part mode (2Nx2N, 2NxN, Nx2N, and the four asymmetric ones from 16x16 up): pred_inter.c:438-444's sizes */
static void cu(int x, int y, int s)
{
	const unsigned m = hash3((unsigned)x, (unsigned)y, 99u) % (s >= 16 ? 7u : 3u);
	const int q = s / 4;
	const pu_t parts[7][2] = {
		{ { 0, 0, s, s }, { 0, 0, 0, 0 } }, { { 0, 0, s, s / 2 }, { 0, s / 2, s, s / 2 } }, { { 0, 0, s / 2, s }, { s / 2, 0, s / 2, s } },
		{ { 0, 0, s, q }, { 0, q, s, s - q } }, { { 0, 0, s, s - q }, { 0, s - q, s, q } }, { { 0, 0, q, s }, { q, 0, s - q, s } }, { { 0, 0, s - q, s }, { s - q, 0, q, s } } };
	for (int k = 0; k < 2; ++k)
		if (parts[m][k].w) pus[n_pus++] = (pu_t){ x + parts[m][k].x, y + parts[m][k].y, parts[m][k].w, parts[m][k].h };
}

static void walk(int x, int y, int s)
{
	if (x >= W || y >= H) return;
	if (s > 8 && (x + s > W || y + s > H || hash3((unsigned)x, (unsigned)y, (unsigned)s) % 100u < 55u))
	{
		const int h = s / 2;
		walk(x, y, h); walk(x + h, y, h); walk(x, y + h, h); walk(x + h, y + h, h);
		return;
	}
	cu(x, y, s);
}

static int by_size(const void *a, const void *b)
{
	const pu_t *p = a, *q = b;
	if (p->w != q->w) return p->w - q->w;
	if (p->h != q->h) return p->h - q->h;
	return p->y != q->y ? p->y - q->y : p->x - q->x;
}

static uint8_t cur[W * H], ref[SR * RH], pred_cpu[W * H], pred_gpu[W * H];
static uint32_t cur_off[MAX_PUS], ref_off[MAX_PUS], ref_int[MAX_PUS], ref_sub[MAX_PUS];
static uint8_t frac[MAX_PUS];
static int mv_cpu[3 * MAX_PUS], mv_gpu[3 * MAX_PUS], q_cpu[3 * MAX_PUS], q_gpu[3 * MAX_PUS], cost_gpu[MAX_PUS];
static hevcasm_hip_cost_run runs[64];
static hevcasm_hip_pred_run pred_runs[64];

int main(void)
{
	/* content: a smooth random reference; the current frame is the reference moved by (+3, -2) in its upper half and by half a sample more in the lower */
	srand(7);
	for (int i = 0; i < SR * RH; ++i) ref[i] = (uint8_t)(rand() >> 7);
	for (int pass = 0; pass < 2; ++pass)
		for (int i = SR + 1; i < SR * RH - SR - 1; ++i) ref[i] = (uint8_t)((ref[i - 1] + 2 * ref[i] + ref[i + 1] + ref[i - SR] + 2 * ref[i] + ref[i + SR] + 4) >> 3);
	for (int y = 0; y < H; ++y)
		for (int x = 0; x < W; ++x)
		{
			const uint8_t *p = ref + (y + PAD - 2) * SR + x + PAD + 3;
			cur[y * W + x] = y < H / 2 ? p[0] : (uint8_t)((p[0] + p[1] + 1) >> 1);
		}
	for (int y = 0; y < H; y += 64)
		for (int x = 0; x < W; x += 64) walk(x, y, 64);
	qsort(pus, (size_t)n_pus, sizeof(pu_t), by_size);
	int n_runs = 0;
	for (int i = 0; i < n_pus; ++i)
	{
		cur_off[i] = (uint32_t)(pus[i].y * W + pus[i].x);
		ref_off[i] = (uint32_t)((pus[i].y + PAD) * SR + pus[i].x + PAD);
		if (i == 0 || pus[i].w != pus[i - 1].w || pus[i].h != pus[i - 1].h)
		{
			runs[n_runs] = (hevcasm_hip_cost_run){ (uint32_t)i, 0, HEVCASM_RECT(pus[i].w, pus[i].h) };
			pred_runs[n_runs] = (hevcasm_hip_pred_run){ (uint32_t)i, 0, 8, (uint8_t)pus[i].w, (uint8_t)pus[i].h, 0 };
			++n_runs;
		}
		++runs[n_runs - 1].count; ++pred_runs[n_runs - 1].count;
	}

	/* 1. the codec's loop as it is today: per PU, the reference's table entries */
	hevcasm_table_sad sads; hevcasm_table_pred_uni_8to8 preds;
	hevcasm_populate_sad(&sads, HEVCASM_C_REF | HEVCASM_C_OPT);
	hevcasm_populate_pred_uni_8to8(&preds, HEVCASM_C_REF | HEVCASM_C_OPT);
	for (int i = 0; i < n_pus; ++i)
	{
		const int w = pus[i].w, h = pus[i].h;
		hevcasm_sad *sad = *hevcasm_get_sad(&sads, w, h);
		const uint8_t *c = cur + cur_off[i], *r0 = ref + ref_off[i];
		int best = 1 << 30, bx = 0, by = 0;
		for (int dy = -RANGE; dy <= RANGE; ++dy)          /* full search: first minimum in raster order of (dy, dx) */
			for (int dx = -RANGE; dx <= RANGE; ++dx)
			{
				const int cost = sad(c, W, r0 + dy * SR + dx, SR, HEVCASM_RECT(w, h));
				if (cost < best) { best = cost; bx = dx; by = dy; }
			}
		mv_cpu[3 * i] = bx; mv_cpu[3 * i + 1] = by; mv_cpu[3 * i + 2] = best;
		const uint8_t *r1 = r0 + by * SR + bx;
		uint8_t tmp[64 * 64];
		best = 1 << 30;
		int qx0 = 0, qy0 = 0;
		for (int qy = -3; qy <= 3; ++qy)                  /* the 49 quarter-pel candidates around it, same tie rule */
			for (int qx = -3; qx <= 3; ++qx)
			{
				hevcasm_pred_uni_8to8 *f = *hevcasm_get_pred_uni_8to8(&preds, 8, w, h, qx & 3, qy & 3);
				f(tmp, 64, r1 + (qy >> 2) * SR + (qx >> 2), SR, w, h, qx & 3, qy & 3);
				const int cost = sad(c, W, tmp, 64, HEVCASM_RECT(w, h));
				if (cost < best) { best = cost; qx0 = qx; qy0 = qy; }
			}
		q_cpu[3 * i] = qx0; q_cpu[3 * i + 1] = qy0; q_cpu[3 * i + 2] = best;
		hevcasm_pred_uni_8to8 *f = *hevcasm_get_pred_uni_8to8(&preds, 8, w, h, qx0 & 3, qy0 & 3);
		f(pred_cpu + cur_off[i], W, r1 + (qy0 >> 2) * SR + (qx0 >> 2), SR, w, h, qx0 & 3, qy0 & 3);
	}

	/* 2. the same frame as four device calls */
	if (!hevcasm_hip_available())
	{
		fprintf(stderr, "no gfx950 device: %s\n", hevcasm_hip_last_error());
		return 77;
	}
	uint8_t *d_cur, *d_ref, *d_pred, *d_frac; uint32_t *d_cur_off, *d_ref_off; int *d_res; void *stream;
	const size_t n4 = 4 * (size_t)n_pus;
	int rc = hevcasm_hip_stream_create(&stream);
	rc |= hevcasm_hip_malloc((void **)&d_cur, sizeof(cur)); rc |= hevcasm_hip_malloc((void **)&d_ref, sizeof(ref)); rc |= hevcasm_hip_malloc((void **)&d_pred, sizeof(cur));
	rc |= hevcasm_hip_malloc((void **)&d_cur_off, n4); rc |= hevcasm_hip_malloc((void **)&d_ref_off, n4); rc |= hevcasm_hip_malloc((void **)&d_res, 3 * n4);
	rc |= hevcasm_hip_malloc((void **)&d_frac, (size_t)n_pus);
	rc |= hevcasm_hip_upload(d_cur, cur, sizeof(cur), stream); rc |= hevcasm_hip_upload(d_ref, ref, sizeof(ref), stream);
	rc |= hevcasm_hip_upload(d_cur_off, cur_off, n4, stream); rc |= hevcasm_hip_upload(d_ref_off, ref_off, n4, stream);
	if (!rc) rc = hevcasm_hip_me_full_search_runs(d_cur, W, d_ref, SR, d_cur_off, d_ref_off, d_res, runs, n_runs, RANGE, stream);
	if (!rc) rc = hevcasm_hip_download(mv_gpu, d_res, 3 * n4, stream);
	if (!rc) rc = hevcasm_hip_stream_synchronize(stream);
	for (int i = 0; i < n_pus; ++i) ref_int[i] = (uint32_t)((int)ref_off[i] + mv_gpu[3 * i + 1] * SR + mv_gpu[3 * i]);      /* the codec's step: vectors -> offsets */
	if (!rc) rc = hevcasm_hip_upload(d_ref_off, ref_int, n4, stream);
	if (!rc) rc = hevcasm_hip_subpel_refine_runs(d_cur, W, d_ref, SR, d_cur_off, d_ref_off, d_res, runs, n_runs, 0, stream);
	if (!rc) rc = hevcasm_hip_download(q_gpu, d_res, 3 * n4, stream);
	if (!rc) rc = hevcasm_hip_stream_synchronize(stream);
	for (int i = 0; i < n_pus; ++i)
	{
		ref_sub[i] = (uint32_t)((int)ref_int[i] + (q_gpu[3 * i + 1] >> 2) * SR + (q_gpu[3 * i] >> 2));
		frac[i] = (uint8_t)((q_gpu[3 * i] & 3) | ((q_gpu[3 * i + 1] & 3) << 4));
	}
	const hevcasm_hip_pred_plane plane = { d_pred, W, d_ref, NULL, SR };
	if (!rc) rc = hevcasm_hip_upload(d_ref_off, ref_sub, n4, stream);
	if (!rc) rc = hevcasm_hip_upload(d_frac, frac, (size_t)n_pus, stream);
	if (!rc) rc = hevcasm_hip_pred_uni_8to8_runs(&plane, 1, pred_runs, n_runs, d_cur_off, d_ref_off, d_frac, stream);
	if (!rc) rc = hevcasm_hip_sad_runs(d_cur, W, d_pred, W, d_cur_off, d_cur_off, d_res, runs, n_runs, stream);      /* the PUs' cost against the prediction now in memory */
	if (!rc) rc = hevcasm_hip_download(cost_gpu, d_res, n4, stream);
	if (!rc) rc = hevcasm_hip_download(pred_gpu, d_pred, sizeof(pred_gpu), stream);
	if (!rc) rc = hevcasm_hip_stream_synchronize(stream);
	if (rc)
	{
		fprintf(stderr, "hevcasm_hip call failed (%d): %s\n", rc, hevcasm_hip_last_error());   /* no CPU fallback exists */
		return 1;
	}
	int bad_mv = 0, bad_q = 0, bad_cost = 0, moved = 0, fractional = 0;
	for (int i = 0; i < n_pus; ++i)
	{
		bad_mv += memcmp(mv_cpu + 3 * i, mv_gpu + 3 * i, 12) != 0;
		bad_q += memcmp(q_cpu + 3 * i, q_gpu + 3 * i, 12) != 0;
		bad_cost += cost_gpu[i] != q_cpu[3 * i + 2];
		moved += mv_cpu[3 * i] != 0 || mv_cpu[3 * i + 1] != 0;
		fractional += (q_cpu[3 * i] & 3) != 0 || (q_cpu[3 * i + 1] & 3) != 0;
	}
	const int bad_pred = memcmp(pred_cpu, pred_gpu, sizeof(pred_cpu)) != 0;
	printf("%d PUs in %d sizes (%d moved, %d with a fractional vector): %d vectors, %d refined vectors, %d costs differ; predicted plane %s\n",
		n_pus, n_runs, moved, fractional, bad_mv, bad_q, bad_cost, bad_pred ? "DIFFERS" : "identical");
	hevcasm_hip_free(d_cur); hevcasm_hip_free(d_ref); hevcasm_hip_free(d_pred); hevcasm_hip_free(d_cur_off); hevcasm_hip_free(d_ref_off); hevcasm_hip_free(d_res); hevcasm_hip_free(d_frac);
	hevcasm_hip_stream_destroy(stream);
	return (bad_mv | bad_q | bad_cost | bad_pred) != 0;
}
