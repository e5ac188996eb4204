"""Seeded random sweep of the batch API against the oracle: sizes the fixed case lists do not name (every multiple
of 4 for the block costs, odd prediction sizes, odd batch lengths), random frame geometry, random parameters.
Deterministic (fixed seed) so that a failure reproduces; about 250 small cases."""
import os

import numpy as np
import pytest

import cases
from hevcasm_amd import workload as wl

pytestmark = pytest.mark.gpu

# tools/soak.py re-runs these with other seeds and longer batches (several waves, row segments) in one process
SEED = int(os.environ.get("HEVCASM_FUZZ_SEED", "0"))
NMUL = int(os.environ.get("HEVCASM_FUZZ_NMUL", "1"))


def _frame(rng, w, h):
    # frames that the block size tiles, of random extent
    return (w * int(rng.integers(1, 5)), h * int(rng.integers(1, 4)))


def test_block_costs_random_geometry():
    rng = np.random.default_rng(20261004 + SEED)
    for i in range(90):
        w, h = 4 * int(rng.integers(1, 17)), 4 * int(rng.integers(1, 17))
        n = int(rng.integers(1, 70)) * NMUL
        ways = int(rng.choice([1, 4]))
        c = wl.sad_case(w, h, n, 5000 + i, ways=ways, max_disp=int(rng.integers(0, 8)), frame=_frame(rng, w, h))
        fam = rng.choice(["sad", "ssd"]) if ways == 1 else "sad4"
        cases.assert_same("fuzz %s %dx%d n=%d" % (fam, w, h, n), cases.run_hip(fam, c), cases.run_oracle(fam, c))


def test_prediction_random_sizes_and_fractions():
    rng = np.random.default_rng(20261005 + SEED)
    for i in range(70):
        taps = int(rng.choice([8, 4]))
        lim = 64 if taps == 8 else 32
        w, h = int(rng.integers(1, lim + 1)), int(rng.integers(1, lim + 1))   # the C bodies take any size up to the limit
        n = int(rng.integers(1, 40)) * NMUL
        bi = bool(rng.integers(0, 2))
        c = wl.pred_case(taps, w, h, n, 6000 + i, bi=bi, frame=(w * int(rng.integers(1, 4)), h * int(rng.integers(1, 4))))
        fam = "pred_bi" if bi else "pred_uni"
        cases.assert_same("fuzz %s taps=%d %dx%d n=%d" % (fam, taps, w, h, n), cases.run_hip(fam, c), cases.run_oracle(fam, c))


def test_prediction_extreme_content():
    """Planes of 0 / 255 only -- independent per pixel, or periodic in both directions with random 8-pixel masks -- drive the filters to the ends of
    their range: both clip rails of uni-prediction, and the 14-bit intermediates of bi-prediction past int16, which the reference wraps
    (pred_inter.c:125: an int stored to an int16_t; the periodic planes reach it, see cases._wrap14).  Random sizes, every fraction pair, both tap counts."""
    rng = np.random.default_rng(20261009 + SEED)
    for i in range(40):
        taps = int(rng.choice([8, 4]))
        lim = 64 if taps == 8 else 32
        w, h = int(rng.integers(1, lim + 1)), int(rng.integers(1, lim + 1))
        n = int(rng.integers(1, 24)) * NMUL
        bi = bool(rng.integers(0, 4))                      # mostly bi: that is where the wrap lives
        c = wl.pred_case(taps, w, h, n, 6500 + i, bi=bi, frame=(w * int(rng.integers(1, 4)), h * int(rng.integers(1, 4))))
        stride = c["stride_ref"]
        for r in range(2 if bi else 1):
            plane = c["ref%d" % r]
            if rng.integers(0, 3) == 0:
                plane[:] = np.where(rng.integers(0, 2, size=plane.size) == 1, 255, 0).astype(np.uint8)
            else:
                mx, my = rng.integers(0, 2, size=8), rng.integers(0, 2, size=8)
                idx = np.arange(plane.size)
                plane[:] = np.where(mx[(idx % stride) % 8] == my[(idx // stride) % 8], 255, 0).astype(np.uint8)
        fam = "pred_bi" if bi else "pred_uni"
        cases.assert_same("fuzz extreme %s taps=%d %dx%d n=%d" % (fam, taps, w, h, n), cases.run_hip(fam, c), cases.run_oracle(fam, c))


def test_transforms_and_quantisers_random_parameters():
    rng = np.random.default_rng(20261006 + SEED)
    for i in range(60):
        tr, lg = wl.TRANSFORMS[int(rng.integers(0, len(wl.TRANSFORMS)))]
        n = int(rng.integers(1, 50)) * NMUL
        size = 1 << lg
        frame = (size * int(rng.integers(1, 6)), size * int(rng.integers(1, 6)))
        kind = str(rng.choice(["full", "natural"]))
        c = wl.inverse_case(tr, lg, n, 7000 + i, kind, frame=frame)
        cases.assert_same("fuzz inverse %d/%d n=%d" % (tr, lg, n), cases.run_hip("inverse", c), cases.run_oracle("inverse", c))
        c["scale"], c["shift"] = int(rng.integers(-300, 20000)), int(rng.integers(1, 16))
        cases.assert_same("fuzz dequant_inverse %d/%d scale=%d shift=%d" % (tr, lg, c["scale"], c["shift"]),
                          cases.run_hip("dequant_inverse", c), cases.run_oracle("dequant_inverse", c))
        f = wl.forward_case(tr, lg, n, 7100 + i, str(rng.choice(["residual9", "full"])), frame=frame)
        cases.assert_same("fuzz forward %d/%d n=%d" % (tr, lg, n), cases.run_hip("forward", f), cases.run_oracle("forward", f))
    for i in range(30):
        npb = int(rng.choice([16, 64, 256, 1024]))
        nb = int(rng.integers(1, 20))
        src = wl.fill_i16(npb * nb, 7200 + i, "full")
        q = dict(src=src, scale=int(rng.integers(1, 0x8000)), shift=int(rng.integers(16, 28)), offset=int(rng.integers(0, 0x8000)), n_per_block=npb)
        cases.assert_same("fuzz quant %d x %d" % (npb, nb), cases.run_hip("quant", q), cases.run_oracle("quant", q))
        d = dict(src=wl.fill_i16(int(rng.integers(1, 3000)), 7300 + i, "full"), scale=int(rng.integers(-32768, 32768)), shift=int(rng.integers(1, 20)))
        cases.assert_same("fuzz dequant n=%d" % d["src"].size, cases.run_hip("dequant", d), cases.run_oracle("dequant", d))


def test_search_operators_and_small_families_random_parameters():
    rng = np.random.default_rng(20261007 + SEED)
    for i in range(24):
        w, h = 4 * int(rng.integers(1, 17)), 4 * int(rng.integers(1, 17))
        c = cases._subpel_case(w, h, int(rng.integers(1, 6)), 8000 + i, "noise") | dict(metric=int(rng.integers(0, 2)))
        cases.assert_same("fuzz subpel %dx%d metric=%d" % (w, h, c["metric"]), cases.run_hip("subpel", c), cases.run_oracle("subpel", c))
    for i in range(12):
        w, h = 4 * int(rng.integers(1, 17)), 4 * int(rng.integers(1, 17))
        c = cases._me_case(w, h, int(rng.integers(0, 17)), int(rng.integers(1, 5)), 8100 + i)
        cases.assert_same("fuzz me %dx%d +-%d" % (w, h, c["range"]), cases.run_hip("me", c), cases.run_oracle("me", c))
    for i in range(12):
        size = int(rng.choice([2, 4, 8]))
        c = wl.sad_case(max(size, 4), max(size, 4), int(rng.integers(1, 300)), 8200 + i) | dict(size=size)
        cases.assert_same("fuzz satd %d" % size, cases.run_hip("satd", c), cases.run_oracle("satd", c))
        # full-amplitude random checkerboards: every Hadamard coefficient can reach 64 * 255 (the 8x8 kernel works in int16)
        src, ref = np.asarray(c["src"]).copy(), np.asarray(c["ref"]).copy()
        src[:] = np.where(rng.integers(0, 2, src.shape) > 0, 255, 0)
        ref[:] = np.where(rng.integers(0, 2, ref.shape) > 0, 255, 0)
        c2 = dict(c, src=src, ref=ref)
        cases.assert_same("fuzz satd %d full amplitude" % size, cases.run_hip("satd", c2), cases.run_oracle("satd", c2))
    for i in range(20):
        size, n = int(rng.integers(0, 5000)), int(rng.integers(1, 12))
        span = max(size, 1) + 5
        c = dict(src0=wl.fill_u8(span * n + 8, 8300 + i), src1=wl.fill_u8(span * n + 8, 8400 + i),
                 off0=(np.arange(n) * span + rng.integers(0, 4, n)).astype(np.uint32), off1=(np.arange(n) * span + rng.integers(0, 4, n)).astype(np.uint32), size=size)
        cases.assert_same("fuzz ssd_linear %d" % size, cases.run_hip("ssd_linear", c), cases.run_oracle("ssd_linear", c))
    for i in range(16):
        log2 = int(rng.integers(2, 6))
        edge = int(rng.integers(0, 2)) if log2 < 5 else 0
        stride, n = int(rng.integers(2 << log2, 300)), int(rng.integers(1, 90))
        c = dict(neighbours=wl.fill_u8(stride * n + 128, 8500 + i), neighbour_stride=stride, mode=1, packed=(8 << 8) | (log2 << 1) | edge, n=n)
        cases.assert_same("fuzz intra %d edge=%d stride=%d" % (1 << log2, edge, stride), cases.run_hip("intra", c), cases.run_oracle("intra", c))
    # round 5: whole-frame SSE -- random widths (every lane deal, rows longer than 256 chunks included), heights, strides, bottom-up planes, frame counts
    for i in range(24):
        w, h, n = int(rng.integers(0, 5000)) if i % 4 else int(rng.integers(0, 40)), int(rng.integers(0, 70)), int(rng.integers(1, 7 * NMUL))
        sa, sb = w + int(rng.integers(0, 40)), w + int(rng.integers(0, 40))
        span_a, span_b = sa * max(h, 1) + 9, sb * max(h, 1) + 9
        up = bool(rng.integers(0, 2)) and h > 0
        c = dict(a=wl.fill_u8(span_a * n + 16, 8600 + i), stride_a=-sa if up else sa, b=wl.fill_u8(span_b * n + 16, 8700 + i), stride_b=sb,
                 a_off=(np.arange(n) * span_a + rng.integers(0, 4, n) + (sa * (h - 1) if up else 0)).astype(np.uint32),
                 b_off=(np.arange(n) * span_b + rng.integers(0, 4, n)).astype(np.uint32), width=w, height=h)
        cases.assert_same("fuzz frame_sse %dx%d x%d" % (w, h, n), cases.run_hip("frame_sse", c), cases.run_oracle("frame_sse", c))
    # round 5 (unpinned beyond DC): a mode per block, random record strides and batch lengths (several workgroup iterations and the partial tail), random
    # and two-level content; the neighbour smoothing with and without the strong form
    for i in range(16):
        log2 = int(rng.integers(2, 6))
        edge = int(rng.integers(0, 2)) if log2 < 5 else 0
        stride, n = int(rng.integers(129, 300)), int(rng.integers(1, 2500 * NMUL))
        nb = wl.fill_u8(stride * n + 160, 8800 + i)
        if i % 3 == 0:
            nb = np.where(nb > 127, 255, 0).astype(np.uint8)
        c = dict(neighbours=nb, neighbour_stride=stride, modes=rng.integers(0, 35, n).astype(np.uint8), packed=(8 << 8) | (log2 << 1) | edge)
        cases.assert_same("fuzz intra modes %d edge=%d stride=%d n=%d" % (1 << log2, edge, stride, n), cases.run_hip("intra_modes", c), cases.run_oracle("intra_modes", c))
        smooth = (nb.astype(np.int32) // 16 + 100).astype(np.uint8) if i % 2 else nb           # nearly flat records take the bi-linear form at 32x32
        f = dict(neighbours=smooth, neighbour_stride=stride, log2=log2, strong=int(rng.integers(0, 2)), n=min(n, 400))
        cases.assert_same("fuzz intra filter %d strong=%d" % (1 << log2, f["strong"]), cases.run_hip("intra_filter", f), cases.run_oracle("intra_filter", f))


def test_frame_search_and_two_plane_prediction_random_geometry():
    """Round-3 entry points: the frame-rows full search on random frames (dimensions any multiple of 4, CTU 8 ... 64, ranges up to
    the LDS limit, any number of ranks) against the oracle over the block lists of the same CTUs; block-list full search at ranges
    beyond 16; Cb + Cr prediction in one launch against two single-plane launches."""
    import torch
    from hevcasm_amd import batch as B, me_frame
    from oracle import pyoracle
    rng = np.random.default_rng(20261008 + SEED)
    for i in range(10):
        ctu = int(rng.choice([8, 16, 32, 64]))
        r = int(rng.integers(0, 41))
        fw, fh = 4 * int(rng.integers(1, 60)), 4 * int(rng.integers(1, 40))
        world = int(rng.integers(1, 5))
        f = me_frame.synthetic_frame_pair(fw, fh, r, seed=8600 + i + SEED, ctu=ctu, max_motion=min(r, 24))
        rows, cols = -(-fh // ctu), -(-fw // ctu)
        want = np.zeros((rows * cols, 3), dtype=np.int32)
        for j in me_frame.ctu_jobs(fh, fw, ctu, f["stride_cur"], f["stride_ref"], r, 0, rows):
            want[j["index"]] = pyoracle.me_full_search(f["cur"], f["stride_cur"], f["ref"], f["stride_ref"], j["cur_off"], j["ref_off"], j["w"], j["h"], r)
        fn = me_frame.hip_rows_fn()
        parts = [me_frame.search_rank(f, ctu, r, rank, world, fn) for rank in range(world)]
        cases.assert_same("fuzz me frame %dx%d ctu %d +-%d x%d" % (fw, fh, ctu, r, world), (me_frame.assemble(parts, fw, fh, ctu),), (want,))
    for i in range(6):
        w, h = 4 * int(rng.integers(1, 17)), 4 * int(rng.integers(1, 17))
        c = cases._me_case(w, h, int(rng.integers(17, 65)), int(rng.integers(1, 4)) * NMUL, 8700 + i, plant=bool(rng.integers(0, 2)))
        cases.assert_same("fuzz me %dx%d +-%d" % (w, h, c["range"]), cases.run_hip("me", c), cases.run_oracle("me", c))
    for i in range(20):
        taps = int(rng.choice([8, 4]))
        lim = 64 if taps == 8 else 32
        w, h = int(rng.integers(1, lim + 1)), int(rng.integers(1, lim + 1))
        n = int(rng.integers(1, 40)) * NMUL
        bi = bool(rng.integers(0, 2))
        frame = (w * int(rng.integers(1, 4)), h * int(rng.integers(1, 4)))
        ca = wl.pred_case(taps, w, h, n, 8800 + i, bi=bi, frame=frame)
        cb = wl.pred_case(taps, w, h, n, 8900 + i, bi=bi, frame=frame)
        refs = ["ref0", "ref1"] if bi else ["ref0"]
        da, db = (torch.zeros(ca["dst_bytes"], dtype=torch.uint8, device="cuda") for _ in range(2))
        wa, wb = (torch.zeros(ca["dst_bytes"], dtype=torch.uint8, device="cuda") for _ in range(2))
        ra, rb = [B.u8(ca[k]) for k in refs], [B.u8(cb[k]) for k in refs]
        doff, offs, fr = B.off32(ca["dst_off"]), [B.off32(ca[k + "_off"]) for k in refs], [B.u8(ca["frac%d" % q]) for q in range(len(refs))]
        if bi:
            B.pred_bi_2planes(da, db, ca["stride_dst"], ra[0], rb[0], ra[1], rb[1], ca["stride_ref"], doff, offs[0], offs[1], fr[0], fr[1], taps, w, h)
            B.pred_bi(wa, ca["stride_dst"], ra[0], ra[1], ca["stride_ref"], doff, offs[0], offs[1], fr[0], fr[1], taps, w, h)
            B.pred_bi(wb, ca["stride_dst"], rb[0], rb[1], ca["stride_ref"], doff, offs[0], offs[1], fr[0], fr[1], taps, w, h)
        else:
            B.pred_uni_2planes(da, db, ca["stride_dst"], ra[0], rb[0], ca["stride_ref"], doff, offs[0], fr[0], taps, w, h)
            B.pred_uni(wa, ca["stride_dst"], ra[0], ca["stride_ref"], doff, offs[0], fr[0], taps, w, h)
            B.pred_uni(wb, ca["stride_dst"], rb[0], ca["stride_ref"], doff, offs[0], fr[0], taps, w, h)
        torch.cuda.synchronize()
        assert torch.equal(da, wa) and torch.equal(db, wb), ("fuzz 2 planes", taps, w, h, bi, n)


def test_intra_frames_random_geometry_batches_and_parameters():
    """Round 6: whole intra frames through the persistent kernel -- random frame sizes (multiples of 4, partial CTUs), the three CTU sizes, luma / chroma, random QP
    and quad-tree shape, every mode, strong smoothing on and off, 1 ... 9 frames per call (NMUL: more), encoder then decoder on the levels just produced.  The
    hand-offs between workgroups are exercised by every multi-CTU frame; batches of unequal frames give them uneven load."""
    import torch
    from hevcasm_amd import batch as B
    from oracle import pyoracle
    rng = np.random.default_rng(20261010 + SEED)
    for i in range(14):
        ctu = int(rng.choice([16, 32, 64]))
        w, h = 4 * int(rng.integers(2, 70)), 4 * int(rng.integers(2, 50))
        cidx, strong, qp = int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 52))
        frames = int(rng.integers(1, 10)) * NMUL
        split = tuple([1.0 if ctu == 64 else float(rng.random())] + [float(rng.random()) for _ in range(3)])
        cs = [cases._intra_frame_case(w, h, ctu, 9000 + 100 * i + f, "mixed" if i % 5 else "dc", qp=qp, cidx=cidx, strong=strong, split=split,
                                      kind=["video", "noise", "flat", "max"][int(rng.integers(0, 4))]) for f in range(frames)]
        c0 = cs[0]
        tus = [c["tus"].view(wl.INTRA_TU_DTYPE).reshape(-1) for c in cs]
        job = B.IntraFrames(w, h, ctu, tus, [c["ctu_first"] for c in cs], c0["qp"], c_idx=cidx, strong=strong, stride=c0["stride"])
        job.run(torch.cat([B.u8(c["src"]) for c in cs]))
        torch.cuda.synchronize()
        job.status()
        plane, per_frame = c0["stride"] * h, job.levels.numel() // frames
        rec, lev, cbf = job.recon.cpu().numpy(), job.levels.cpu().numpy(), job.cbf.cpu().numpy()
        t0 = 0
        for f, c in enumerate(cs):
            want = pyoracle.intra_reconstruct_frame(c["src"], c["stride"], None, c["tus"], c["qp"], w, h, c["stride"], ctu, cidx, strong)
            what = "fuzz intra %dx%d ctu=%d cidx=%d qp=%d frame %d/%d" % (w, h, ctu, cidx, qp, f, frames)
            cases.assert_same(what + " recon", (rec[f * plane:(f + 1) * plane],), (want[0],))
            cases.assert_same(what + " levels", (lev[f * per_frame:(f + 1) * per_frame],), (want[1],))
            cases.assert_same(what + " cbf", (cbf[t0:t0 + len(c["tus"])],), (want[2],))
            t0 += len(c["tus"])
        enc = job.recon.clone()
        job.recon.zero_()
        job.run(None)                                     # the decoder's half on the same levels
        torch.cuda.synchronize()
        job.status()
        assert torch.equal(job.recon, enc), "fuzz intra: decoder != encoder's reconstruction (%dx%d ctu=%d)" % (w, h, ctu)


def test_pred_runs_random_partitions():
    """Round 6: a frame's inter partition in one launch -- random frame sizes (multiples of 8), uni and bi, with and without chroma, against the oracle per run."""
    import test_pred_runs
    from oracle import pyoracle
    rng = np.random.default_rng(20261011 + SEED)
    for i in range(8):
        w, h = 8 * int(rng.integers(1, 40)), 8 * int(rng.integers(1, 30))
        bi, chroma = bool(rng.integers(0, 2)), bool(rng.integers(0, 4))
        c = wl.pred_frame_case(w, h, 9500 + i, bi=bi, chroma=chroma, kind=["noise", "video", "max"][int(rng.integers(0, 3))])
        for plane, (g, x) in enumerate(zip(test_pred_runs._run_hip(c), test_pred_runs._oracle_planes(pyoracle, c))):
            assert np.array_equal(g, x), "fuzz pred runs %dx%d bi=%s plane %d" % (w, h, bi, plane)


def test_cost_runs_random_partitions():
    """Round 6: the block costs of a frame's partition in one launch -- random frame sizes (multiples of 8), the three cost kinds, against the oracle per run."""
    import test_cost_runs
    from oracle import pyoracle
    rng = np.random.default_rng(20261012 + SEED)
    for i in range(8):
        w, h = 8 * int(rng.integers(1, 40)), 8 * int(rng.integers(1, 30))
        c = wl.cost_frame_case(w, h, 9700 + i, kind=["noise", "video", "max"][int(rng.integers(0, 3))], search=int(rng.integers(0, 40)))
        for kind in test_cost_runs.KINDS:
            assert np.array_equal(test_cost_runs._hip(c, kind), test_cost_runs._oracle(pyoracle, c, kind)), "fuzz cost runs %dx%d %s" % (w, h, kind)


def test_tu_runs_random_partitions_and_parameters():
    """Round 6: a plane's transform units in one launch per stage and the whole residual loop in ONE launch -- random plane sizes (multiples of 4), QP, content,
    row padding, unit order, against the oracle per run; the fused launch also against the three stage launches' results (the same numbers by another route)."""
    import test_tu_runs
    from oracle import pyoracle
    rng = np.random.default_rng(20261013 + SEED)
    kinds = ["noise", "video", "max", "zeros"]
    for i in range(8 * NMUL):
        w, h = 4 * int(rng.integers(1, 80)), 4 * int(rng.integers(1, 60))
        c = wl.tu_frame_case(w, h, 9900 + i + SEED, qp=int(rng.integers(0, 52)), kind=kinds[int(rng.integers(0, 4))], src_kind=kinds[int(rng.integers(0, 4))],
                             residual=["residual9", "full", "natural"][int(rng.integers(0, 3))], pad=4 * int(rng.integers(0, 5)), raster=bool(rng.integers(0, 2)))
        (coeffs,), (want,) = test_tu_runs._hip(c, "forward"), test_tu_runs._oracle(pyoracle, c, "forward")
        assert np.array_equal(coeffs, want), "fuzz tu runs %dx%d forward" % (w, h)
        for g, x in zip(test_tu_runs._hip(c, "quantize", coeffs), test_tu_runs._oracle(pyoracle, c, "quantize", coeffs)):
            assert np.array_equal(g, x), "fuzz tu runs %dx%d quantize" % (w, h)
        full = wl.fill_i16(c["n_coeffs"], 77 + i, "full")
        for stage in ("inverse", "dequant_inverse"):
            assert np.array_equal(test_tu_runs._hip(c, stage, full)[0], test_tu_runs._oracle(pyoracle, c, stage, full)[0]), "fuzz tu runs %dx%d %s" % (w, h, stage)
        m = test_tu_runs._mask(c)
        for in_place in (False, True):
            for g, x, what in zip(test_tu_runs._fused_hip(c, in_place=in_place), test_tu_runs._fused_oracle(pyoracle, c), ("levels", "cbf", "reconstruction")):
                ok = np.array_equal(g[m], x[m]) if what == "reconstruction" else np.array_equal(g, x)
                assert ok, "fuzz tu runs %dx%d fused %s in_place=%s" % (w, h, what, in_place)
