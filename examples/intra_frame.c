/*
 * intra_frame.c -- a C codec's intra reconstruction loop, twice: TU by TU through the reference's function-pointer tables (pred_intra.h:55,
 * residual_decode.h:54 / :82, quantize.h:57 / :78 -- the C entries of this library), and as ONE call of hevcasm_hip_intra_reconstruct_frames, which
 * does the same loop on the device in wavefront order.  Every TU is INTRA_DC, the one mode the reference's table holds,
 * so the two must agree bit for bit.
 *
 *   cc -Iinclude examples/intra_frame.c -Lhevcasm_amd/lib -lhevcasm_hip -Wl,-rpath,$PWD/hevcasm_amd/lib -o intra_frame
 *   ./intra_frame          exit status 0 = reconstruction, levels and cbf identical (77 = no gfx950 device visible)
 */
#include "pred_intra.h"
#include "residual_decode.h"
#include "quantize.h"
#include "hevcasm_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { W = 328, H = 200, CTU = 64, COLS = (W + CTU - 1) / CTU, ROWS = (H + CTU - 1) / CTU, MAX_TUS = (W / 4) * (H / 4) };

static hevcasm_hip_intra_tu tus[MAX_TUS];
static uint32_t ctu_first[COLS * ROWS + 1];
static int n_tus;

/* a quad-tree in decoding order: split while the square crosses the picture edge or a cheap hash of its position says so */
static void walk(int x, int y, int s)
{
	if (x >= W || y >= H) return;
	const int crosses = x + s > W || y + s > H, wants = ((x * 7 + y * 13 + s) % 5) < 2;
	if (s > 4 && (s > 32 || crosses || wants))
	{
		const int h = s / 2;
		walk(x, y, h); walk(x + h, y, h); walk(x, y + h, h); walk(x + h, y + h, h);      /* z-scan */
		return;
	}
	int log2 = 2;
	while ((1 << log2) < s) ++log2;
	tus[n_tus++] = (hevcasm_hip_intra_tu){ (uint16_t)x, (uint16_t)y, (uint8_t)log2, 1 /* INTRA_DC */, 0 };
}

/* H.265 8.4.4.2.2: the neighbour record of the n x n block at (x0, y0), from the samples reconstructed so far */
static void neighbours(uint8_t *nb, const uint8_t *recon, const uint8_t *decoded, int x0, int y0, int n)
{
	int found = 0;
	uint8_t last = 128;
	for (int pass = 0; pass < 2; ++pass)
		for (int s = 0; s <= 4 * n; ++s)
		{
			const int x = s <= 2 * n ? x0 - 1 : x0 + s - 2 * n - 1, y = s < 2 * n ? y0 + 2 * n - 1 - s : y0 - 1;
			const int ok = x >= 0 && y >= 0 && x < W && y < H && decoded[(y / 4) * (W / 4) + x / 4];
			if (pass == 0) { if (ok && !found) { found = 1; last = recon[y * W + x]; } continue; }      /* pass 0: what the scan meets first */
			if (ok) last = recon[y * W + x];
			nb[s < 2 * n ? 63 - (2 * n - 1 - s) : s == 2 * n ? 128 : 64 + s - 2 * n - 1] = last;
		}
}

static unsigned z_index(unsigned x, unsigned y)
{
	unsigned z = 0;
	for (int b = 0; b < 4; ++b) z |= ((x >> b) & 1u) << (2 * b) | ((y >> b) & 1u) << (2 * b + 1);
	return z;
}

int main(void)
{
	static uint8_t src[W * H], rec_host[W * H], rec_gpu[W * H], decoded[(W / 4) * (H / 4)];
	static int16_t lev_host[COLS * ROWS * CTU * CTU], lev_gpu[COLS * ROWS * CTU * CTU];
	static int cbf_host[MAX_TUS], cbf_gpu[MAX_TUS];
	srand(11);
	for (int y = 0; y < H; ++y)
		for (int x = 0; x < W; ++x) src[y * W + x] = (uint8_t)(128 + 60 * ((x / 24 + y / 16) & 1) - 30 + rand() % 9);
	for (int cy = 0; cy < ROWS; ++cy)
		for (int cx = 0; cx < COLS; ++cx)
		{
			ctu_first[cy * COLS + cx] = (uint32_t)n_tus;
			walk(cx * CTU, cy * CTU, CTU);
		}
	ctu_first[COLS * ROWS] = (uint32_t)n_tus;
	/* HEVC's quantiser at QP 27 for 8-bit video, per TU size: exactly the arguments of hevcasm_quantize / hevcasm_quantize_inverse */
	hevcasm_hip_intra_qp qp[4];
	for (int k = 0; k < 4; ++k) qp[k] = (hevcasm_hip_intra_qp){ 18396, 21 - (k + 2) + 4, 171 << 7, 57 << 4, k + 1 };

	/* 1. the codec's loop as it is today: one table call per step and TU (C entries) */
	hevcasm_table_pred_intra t_intra; hevcasm_table_transform t_fwd; hevcasm_table_inverse_transform_add t_inv;
	hevcasm_table_quantize t_q; hevcasm_table_quantize_inverse t_dq;
	const hevcasm_instruction_set c_bits = (hevcasm_instruction_set)(HEVCASM_C_REF | HEVCASM_C_OPT);
	hevcasm_populate_pred_intra(&t_intra, c_bits); hevcasm_populate_transform(&t_fwd, c_bits); hevcasm_populate_inverse_transform_add(&t_inv, c_bits, 1);
	hevcasm_populate_quantize(&t_q, c_bits); hevcasm_populate_quantize_inverse(&t_dq, c_bits);
	for (int i = 0; i < n_tus; ++i)
	{
		const int x = tus[i].x, y = tus[i].y, log2 = tus[i].log2TrafoSize, n = 1 << log2, packed = hevcasm_pred_intra_pack(0, log2), dst4 = log2 == 2;
		uint8_t record[192] = { 0 }, pred[32 * 32];
		int16_t res[32 * 32], coef[32 * 32], deq[32 * 32];
		neighbours(record + 32, rec_host, decoded, x, y, n);
		(*hevcasm_get_pred_intra(&t_intra, 1, packed))(pred, record + 32, 1, packed);
		for (int r = 0; r < n; ++r)
			for (int c = 0; c < n; ++c) res[r * n + c] = (int16_t)(src[(y + r) * W + x + c] - pred[r * n + c]);
		int16_t *lev = lev_host + ((y / CTU) * COLS + x / CTU) * CTU * CTU + 16 * z_index((unsigned)(x % CTU) / 4, (unsigned)(y % CTU) / 4);
		(*hevcasm_get_transform(&t_fwd, dst4, log2))(coef, res, n);
		cbf_host[i] = (*hevcasm_get_quantize(&t_q))(lev, coef, qp[log2 - 2].q_scale, qp[log2 - 2].q_shift, qp[log2 - 2].q_offset, n * n);
		(*hevcasm_get_quantize_inverse(&t_dq))(deq, lev, qp[log2 - 2].dq_scale, qp[log2 - 2].dq_shift, n * n);
		(*hevcasm_get_inverse_transform_add(&t_inv, dst4, log2))(rec_host + y * W + x, W, pred, n, deq);
		for (int cy = y / 4; cy < (y + n) / 4; ++cy)
			for (int cx = x / 4; cx < (x + n) / 4; ++cx) decoded[cy * (W / 4) + cx] = 1;
	}

	/* 2. the same loop as one call: the frame, its TU map and the quantiser parameters in; reconstruction, levels and cbf out */
	if (!hevcasm_hip_available())
	{
		fprintf(stderr, "no gfx950 device: %s\n", hevcasm_hip_last_error());
		return 77;
	}
	uint8_t *d_src, *d_rec; int16_t *d_lev; int *d_cbf; hevcasm_hip_intra_tu *d_tus; uint32_t *d_first; void *d_ws, *stream;
	const size_t ws = hevcasm_hip_intra_frames_workspace_bytes(W, H, CTU, 1);
	int rc = hevcasm_hip_stream_create(&stream);
	rc |= hevcasm_hip_malloc((void **)&d_src, sizeof(src)); rc |= hevcasm_hip_malloc((void **)&d_rec, sizeof(rec_gpu));
	rc |= hevcasm_hip_malloc((void **)&d_lev, sizeof(lev_gpu)); rc |= hevcasm_hip_malloc((void **)&d_cbf, sizeof(cbf_gpu));
	rc |= hevcasm_hip_malloc((void **)&d_tus, sizeof(tus)); rc |= hevcasm_hip_malloc((void **)&d_first, sizeof(ctu_first));
	rc |= hevcasm_hip_malloc(&d_ws, ws);
	rc |= hevcasm_hip_upload(d_src, src, sizeof(src), stream); rc |= hevcasm_hip_upload(d_tus, tus, sizeof(tus[0]) * (size_t)n_tus, stream);
	rc |= hevcasm_hip_upload(d_first, ctu_first, sizeof(ctu_first), stream);
	rc |= hevcasm_hip_upload(d_lev, lev_gpu, sizeof(lev_gpu), stream);      /* zeros: the cells outside the picture are never written */
	if (!rc) rc = hevcasm_hip_intra_reconstruct_frames(d_rec, W, (size_t)W * H, d_src, W, (size_t)W * H, d_lev, d_cbf, NULL, d_tus, d_first, qp,
	                                                    W, H, CTU, 0, 0, 1, d_ws, stream);
	if (!rc) rc = hevcasm_hip_intra_frames_status(d_ws, stream);
	if (!rc) rc = hevcasm_hip_download(rec_gpu, d_rec, sizeof(rec_gpu), stream);
	if (!rc) rc = hevcasm_hip_download(lev_gpu, d_lev, sizeof(lev_gpu), stream);
	if (!rc) rc = hevcasm_hip_download(cbf_gpu, d_cbf, sizeof(cbf_gpu[0]) * (size_t)n_tus, stream);
	if (!rc) rc = hevcasm_hip_stream_synchronize(stream);
	if (rc)
	{
		fprintf(stderr, "hevcasm_hip call failed (%d): %s\n", rc, hevcasm_hip_last_error());   /* no CPU fallback exists */
		return 1;
	}
	int bad = memcmp(rec_host, rec_gpu, sizeof(rec_host)) != 0;
	bad += memcmp(lev_host, lev_gpu, sizeof(lev_host)) != 0;
	bad += memcmp(cbf_host, cbf_gpu, sizeof(cbf_host[0]) * (size_t)n_tus) != 0;
	printf("%d TUs of a %dx%d frame: %d of (reconstruction, levels, cbf) differ between the per-TU C tables and hevcasm_hip_intra_reconstruct_frames\n", n_tus, W, H, bad);
	hevcasm_hip_free(d_src); hevcasm_hip_free(d_rec); hevcasm_hip_free(d_lev); hevcasm_hip_free(d_cbf); hevcasm_hip_free(d_tus); hevcasm_hip_free(d_first); hevcasm_hip_free(d_ws);
	hevcasm_hip_stream_destroy(stream);
	return bad != 0;
}
