/*
 * tu_chain.c -- a C codec's residual loop for one plane cut into transform units of every size, twice: TU by TU through the reference's function-pointer
 * tables (residual_decode.h:82, :54, quantize.h:78, :57 -- the C entries of this library: hevcasm_transform, hevcasm_quantize, hevcasm_quantize_inverse,
 * hevcasm_inverse_transform_add), and as THREE device calls for the whole plane, one per stage, whatever the sizes:
 *
 *     hevcasm_hip_transform_runs  ->  hevcasm_hip_quantize_runs  ->  hevcasm_hip_dequant_inverse_transform_add_runs
 *
 * and once more as ONE call that starts from the source and prediction planes and keeps everything between them in the workgroup:
 *
 *     hevcasm_hip_transform_quantize_reconstruct_runs
 *
 * Each size has its own quantiser parameters (HEVC's shifts depend on the transform size); they travel in the runs.  Levels, coded-block flags and the
 * reconstructed plane must agree bit for bit.
 *
 *   cc -Iinclude examples/tu_chain.c -Lhevcasm_amd/lib -lhevcasm_hip -Wl,-rpath,$PWD/hevcasm_amd/lib -o tu_chain
 *   ./tu_chain          exit status 0 = identical (77 = no gfx950 device visible)
 */
#include "residual_decode.h"
#include "quantize.h"
#include "hevcasm_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { W = 352, H = 200, MAX_TUS = (W / 4) * (H / 4), QP = 24 };

typedef struct { int x, y, log2, dst; } tu_t;
static tu_t tus[MAX_TUS];
static int n_tus;

static unsigned hash3(unsigned a, unsigned b, unsigned c) { unsigned h = a * 2654435761u ^ b * 40503u ^ c * 69069u; h ^= h >> 13; h *= 0x5bd1e995u; return h ^ (h >> 15); }

/* a residual quad-tree from 32x32 down to 4x4 (forced splits at the plane's edges: H is not a multiple of 32); a third of the 4x4 units take the DST */
static void walk(int x, int y, int s)
{
	if (x >= W || y >= H) return;
	if (s > 4 && (x + s > W || y + s > H || hash3((unsigned)x, (unsigned)y, (unsigned)s) % 100u < 60u))
	{
		const int h = s / 2;
		walk(x, y, h); walk(x + h, y, h); walk(x, y + h, h); walk(x + h, y + h, h);
		return;
	}
	int log2 = 2;
	while ((1 << log2) < s) ++log2;
	tus[n_tus++] = (tu_t){ x, y, log2, s == 4 && hash3((unsigned)x, (unsigned)y, 5u) % 3u == 0 };
}

static int by_size(const void *a, const void *b)
{
	const tu_t *p = a, *q = b;
	if (p->log2 != q->log2) return p->log2 - q->log2;
	if (p->dst != q->dst) return q->dst - p->dst;
	return p->y != q->y ? p->y - q->y : p->x - q->x;
}

static int16_t residual[W * H], coeffs[W * H], levels_cpu[W * H], levels_gpu[W * H], levels_one[W * H];
static uint8_t src[W * H], pred[W * H], rec_cpu[W * H], rec_gpu[W * H], rec_one[W * H];
static uint32_t off[MAX_TUS];
static int cbf_cpu[MAX_TUS], cbf_gpu[MAX_TUS], cbf_one[MAX_TUS];
static hevcasm_hip_tu_run runs_q[8], runs_dq[8];

int main(void)
{
	/* HEVC's parameters for 8-bit video, flat scaling lists (H.265 8.6.4.2; forward: quantScales, shift 14 + qp / 6 + 15 - 8 - log2N, rounding 171 / 512) */
	static const int quant_scales[6] = { 26214, 23302, 20560, 18396, 16384, 14564 }, level_scale[6] = { 40, 45, 51, 57, 64, 72 };
	srand(11);
	for (int i = 0; i < W * H; ++i) pred[i] = (uint8_t)(rand() >> 7);
	for (int i = 0; i < W * H; ++i)
	{
		const int v = pred[i] + (rand() >> 6) % 61 - 30 + ((i / W) & 32 ? 40 : 0);      /* the source: the prediction plus a residual with some structure */
		src[i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
		residual[i] = (int16_t)(src[i] - pred[i]);                                     /* the codec's own step in front of hevcasm_transform */
	}
	for (int y = 0; y < H; y += 32)
		for (int x = 0; x < W; x += 32) walk(x, y, 32);
	qsort(tus, (size_t)n_tus, sizeof(tu_t), by_size);
	int n_runs = 0;
	uint32_t coeff_at = 0;
	for (int i = 0; i < n_tus; ++i)
	{
		off[i] = (uint32_t)(tus[i].y * W + tus[i].x);
		if (i == 0 || tus[i].log2 != tus[i - 1].log2 || tus[i].dst != tus[i - 1].dst)
		{
			const int log2 = tus[i].log2;
			runs_q[n_runs] = (hevcasm_hip_tu_run){ (uint32_t)i, 0, coeff_at, (uint8_t)log2, (uint8_t)tus[i].dst, 0, quant_scales[QP % 6], 21 - log2 + QP / 6, 171 << 7 };
			runs_dq[n_runs] = (hevcasm_hip_tu_run){ (uint32_t)i, 0, coeff_at, (uint8_t)log2, (uint8_t)tus[i].dst, 0, level_scale[QP % 6] << (QP / 6), log2 - 1, 0 };
			++n_runs;
		}
		++runs_q[n_runs - 1].count; ++runs_dq[n_runs - 1].count;
		coeff_at += 1u << (2 * tus[i].log2);
	}

	/* 1. the codec's loop as it is today: per TU, the reference's table entries */
	hevcasm_table_transform fwd; hevcasm_table_inverse_transform_add inv; hevcasm_table_quantize q; hevcasm_table_quantize_inverse dq;
	hevcasm_populate_transform(&fwd, HEVCASM_C_REF | HEVCASM_C_OPT);
	hevcasm_populate_inverse_transform_add(&inv, HEVCASM_C_REF | HEVCASM_C_OPT, 1);
	hevcasm_populate_quantize(&q, HEVCASM_C_REF | HEVCASM_C_OPT);
	hevcasm_populate_quantize_inverse(&dq, HEVCASM_C_REF | HEVCASM_C_OPT);
	for (int r = 0; r < n_runs; ++r)
		for (uint32_t k = 0; k < runs_q[r].count; ++k)
		{
			const int i = (int)(runs_q[r].first + k), log2 = tus[i].log2, n = 1 << (2 * log2);
			int16_t *lv = levels_cpu + runs_q[r].coeff_first + (size_t)k * n, c[32 * 32], d[32 * 32];
			(*hevcasm_get_transform(&fwd, tus[i].dst, log2))(c, residual + off[i], W);
			cbf_cpu[i] = (*hevcasm_get_quantize(&q))(lv, c, runs_q[r].scale, runs_q[r].shift, runs_q[r].offset, n);
			(*hevcasm_get_quantize_inverse(&dq))(d, lv, runs_dq[r].scale, runs_dq[r].shift, n);
			(*hevcasm_get_inverse_transform_add(&inv, tus[i].dst, log2))(rec_cpu + off[i], W, pred + off[i], W, d);
		}

	/* 2. the same plane as three device calls */
	if (!hevcasm_hip_available())
	{
		fprintf(stderr, "no gfx950 device: %s\n", hevcasm_hip_last_error());
		return 77;
	}
	int16_t *d_res, *d_coeffs, *d_levels; uint8_t *d_pred, *d_rec, *d_src; uint32_t *d_off; int *d_cbf; void *stream;
	int rc = hevcasm_hip_stream_create(&stream);
	rc |= hevcasm_hip_malloc((void **)&d_res, sizeof(residual)); rc |= hevcasm_hip_malloc((void **)&d_coeffs, sizeof(coeffs)); rc |= hevcasm_hip_malloc((void **)&d_levels, sizeof(coeffs));
	rc |= hevcasm_hip_malloc((void **)&d_pred, sizeof(pred)); rc |= hevcasm_hip_malloc((void **)&d_rec, sizeof(pred)); rc |= hevcasm_hip_malloc((void **)&d_src, sizeof(src));
	rc |= hevcasm_hip_malloc((void **)&d_off, 4 * (size_t)n_tus); rc |= hevcasm_hip_malloc((void **)&d_cbf, 4 * (size_t)n_tus);
	rc |= hevcasm_hip_upload(d_res, residual, sizeof(residual), stream); rc |= hevcasm_hip_upload(d_pred, pred, sizeof(pred), stream);
	rc |= hevcasm_hip_upload(d_off, off, 4 * (size_t)n_tus, stream); rc |= hevcasm_hip_upload(d_src, src, sizeof(src), stream);
	if (!rc) rc = hevcasm_hip_transform_runs(d_coeffs, d_res, W, d_off, runs_q, n_runs, stream);
	if (!rc) rc = hevcasm_hip_quantize_runs(d_levels, d_coeffs, d_cbf, runs_q, n_runs, stream);
	if (!rc) rc = hevcasm_hip_dequant_inverse_transform_add_runs(d_rec, W, d_pred, W, d_off, d_off, d_levels, runs_dq, n_runs, stream);
	if (!rc) rc = hevcasm_hip_download(levels_gpu, d_levels, sizeof(levels_gpu), stream);
	if (!rc) rc = hevcasm_hip_download(cbf_gpu, d_cbf, 4 * (size_t)n_tus, stream);
	if (!rc) rc = hevcasm_hip_download(rec_gpu, d_rec, sizeof(rec_gpu), stream);
	/* 3. and as one call from the source plane (outputs cleared first, so that nothing of step 2 can pass for its result) */
	if (!rc) rc = hevcasm_hip_upload(d_levels, coeffs, sizeof(coeffs), stream);            /* coeffs[] is still all zero on the host */
	if (!rc) rc = hevcasm_hip_upload(d_rec, (const uint8_t *)coeffs, sizeof(rec_one), stream);
	if (!rc) rc = hevcasm_hip_upload(d_cbf, coeffs, 4 * (size_t)n_tus, stream);
	if (!rc) rc = hevcasm_hip_transform_quantize_reconstruct_runs(d_rec, W, d_src, W, d_pred, W, d_off, d_off, d_off, d_levels, d_cbf, runs_q, runs_dq, n_runs, stream);
	if (!rc) rc = hevcasm_hip_download(levels_one, d_levels, sizeof(levels_one), stream);
	if (!rc) rc = hevcasm_hip_download(cbf_one, d_cbf, 4 * (size_t)n_tus, stream);
	if (!rc) rc = hevcasm_hip_download(rec_one, d_rec, sizeof(rec_one), stream);
	if (!rc) rc = hevcasm_hip_stream_synchronize(stream);
	if (rc)
	{
		fprintf(stderr, "hevcasm_hip call failed (%d): %s\n", rc, hevcasm_hip_last_error());   /* no CPU fallback exists */
		return 1;
	}
	int bad_cbf = 0, coded = 0;
	for (int i = 0; i < n_tus; ++i)
	{
		bad_cbf += (cbf_cpu[i] != 0) != (cbf_gpu[i] != 0);      /* zero / non-zero is the flag's meaning (quantize.c:257) */
		bad_cbf += (cbf_cpu[i] != 0) != (cbf_one[i] != 0);
		coded += cbf_cpu[i] != 0;
	}
	const int bad_levels = memcmp(levels_cpu, levels_gpu, sizeof(levels_cpu)) != 0 || memcmp(levels_cpu, levels_one, sizeof(levels_cpu)) != 0;
	const int bad_rec = memcmp(rec_cpu, rec_gpu, sizeof(rec_cpu)) != 0 || memcmp(rec_cpu, rec_one, sizeof(rec_cpu)) != 0;
	printf("%d TUs in %d runs (%d coded), three calls and one call: levels %s, %d coded-block flags differ, reconstructed plane %s\n",
		n_tus, n_runs, coded, bad_levels ? "DIFFER" : "identical", bad_cbf, bad_rec ? "DIFFERS" : "identical");
	hevcasm_hip_free(d_res); hevcasm_hip_free(d_coeffs); hevcasm_hip_free(d_levels); hevcasm_hip_free(d_pred); hevcasm_hip_free(d_rec); hevcasm_hip_free(d_off); hevcasm_hip_free(d_cbf); hevcasm_hip_free(d_src);
	hevcasm_hip_stream_destroy(stream);
	return (bad_levels | bad_cbf | bad_rec) != 0;
}
