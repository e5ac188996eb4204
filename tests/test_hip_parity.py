"""GPU parity proper: every hot-path family through the C-ABI on cuda:0 against the oracle on
identical seeded inputs (bit-exact: all paths are integer), plus size-independent properties at
BASELINE.json's full batch size."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("group", sorted(cases.ALL_CASE_LISTS))
def test_hip_equals_oracle(group):
    for name, family, c in cases.ALL_CASE_LISTS[group]():
        cases.assert_same(name, cases.run_hip(family, c), cases.run_oracle(family, c))


def test_hip_loaded_native_library():
    """The parity above must have gone through the in-tree HIP library, not a fallback."""
    from hevcasm_amd import capi
    assert capi.lib().hevcasm_hip_available() == 1
    with open("/proc/self/maps") as f:
        assert "libhevcasm_hip.so" in f.read()


def test_sad_full_size_properties():
    """1M blocks of 16x16 (config #2 headline): SAD(x, x) = 0; SAD is symmetric; the 4-way kernel
    equals four single-reference launches; a sampled slice equals the oracle."""
    import torch
    from hevcasm_amd import batch as B, workload as wl
    from oracle import pyoracle
    n = 1 << 20
    c = wl.sad_case(16, 16, n, 4242, ways=4, frame=(wl.FRAME_W, wl.FRAME_H), device="cuda")
    src, ref = c["src"], c["ref"]
    so = B.off32(c["src_off"])
    ro4 = B.off32(c["ref_off"])
    ro = ro4.view(n, 4)
    zero = B.sad(src, c["stride_src"], src, c["stride_src"], so, so, 16, 16)
    assert int(zero.abs().max()) == 0
    four = B.sad_multiref(src, c["stride_src"], ref, c["stride_ref"], so, ro4, 16, 16)
    for k in range(4):
        one = B.sad(src, c["stride_src"], ref, c["stride_ref"], so, ro[:, k].contiguous(), 16, 16)
        assert torch.equal(one, four[:, k])
        if k == 0:
            swapped = B.sad(ref, c["stride_ref"], src, c["stride_src"], ro[:, 0].contiguous(), so, 16, 16)
            assert torch.equal(one, swapped)
    # sampled slice against the oracle: blocks of the last frame (exercises large offsets)
    per_frame = (wl.FRAME_W // 16) * (wl.FRAME_H // 16)
    a = n - per_frame
    src_h, ref_h = src.cpu().numpy(), ref.cpu().numpy()
    want = pyoracle.sad_multiref_batch(src_h, c["stride_src"], ref_h, c["stride_ref"], c["src_off"][a:], c["ref_off"][4 * a:], 16, 16)
    assert np.array_equal(four[a:].cpu().numpy(), want)
    # SSD: zero on identical blocks, and >= SAD^2 / (w*h) (Cauchy-Schwarz) elsewhere
    ssd = B.ssd(src, c["stride_src"], ref, c["stride_ref"], so, ro[:, 0].contiguous(), 16, 16)
    assert bool((ssd.double() * 256 >= four[:, 0].double() ** 2).all())
    assert int(B.ssd(src, c["stride_src"], src, c["stride_src"], so, so, 16, 16).abs().max()) == 0


@pytest.mark.parametrize("w,h", [(64, 64), (48, 64), (32, 32), (64, 16), (16, 64), (8, 8)])
def test_sad4_full_size_on_the_bench_geometry(w, h):
    """The kernels bench.py's step is dominated by, at the shape bench.py times them: 1 Mi blocks per size on 1920x1152
    frames (reference stride 1952), built by bench.py's own SizeBatch -- 64x64 as two launches of 524 288 blocks each
    (the 4 GiB offset window), the window kernel for the 64-/48-wide ones.  Per launch: the 4-way result
    equals four single-reference launches; SAD of the source against itself is 0; the blocks of the LAST frame (largest
    offsets) equal the oracle on a host copy of that frame alone."""
    import torch
    import bench
    from hevcasm_amd import batch as B, workload as wl
    from oracle import pyoracle
    sb = bench.SizeBatch(w, h, 1 << 20, 4711 + w + h, torch.device("cuda", 0))
    assert len(sb.chunks) == (2 if (w, h) == (64, 64) else 1)
    per_frame = (wl.FRAME_W // w) * (wl.FRAME_H // h)
    for c in sb.chunks:
        n = c["n"]
        sb.launch(4, c)
        four = c["out4"]
        ro = c["ref_off4"].view(n, 4)
        for k in range(4):
            one = B.sad(c["src"], c["stride_src"], c["ref"], c["stride_ref"], c["src_off"], ro[:, k].contiguous(), w, h)
            assert torch.equal(one, four[:, k]), (w, h, k)
        assert int(B.sad(c["src"], c["stride_src"], c["src"], c["stride_src"], c["src_off"], c["src_off"], w, h).abs().max()) == 0
        # last whole frame of the chunk against the oracle: copy just that frame of each plane to the host and rebase the offsets
        f = (n - 1) // per_frame
        a, b = f * per_frame, min(n, (f + 1) * per_frame)
        src_frame, ref_frame = wl.FRAME_W * wl.FRAME_H, (wl.FRAME_W + 32) * (wl.FRAME_H + 16)
        src_h = c["src"][f * src_frame: (f + 1) * src_frame].cpu().numpy()
        ref_h = c["ref"][f * ref_frame: (f + 1) * ref_frame].cpu().numpy()
        so = (c["src_off"][a:b].cpu().numpy().view(np.uint32).astype(np.int64) - f * src_frame).astype(np.uint32)
        ro_h = (c["ref_off4"][4 * a: 4 * b].cpu().numpy().view(np.uint32).astype(np.int64) - f * ref_frame).astype(np.uint32)
        want = pyoracle.sad_multiref_batch(src_h, c["stride_src"], ref_h, c["stride_ref"], so, ro_h, w, h)
        assert np.array_equal(four[a:b].cpu().numpy(), want), (w, h)
    del sb
    torch.cuda.empty_cache()


def test_transform_roundtrip_full_size():
    """Forward DCT then inverse transform + add onto a zero prediction returns the (clipped)
    residual to within the HEVC integer transform pair's rounding (a few LSB): a property no single kernel can fake."""
    import torch
    from hevcasm_amd import batch as B, workload as wl
    for tr, lg in wl.TRANSFORMS:
        size = 1 << lg
        n = (1 << 20) >> (2 * lg - 4) if lg > 2 else 1 << 18
        c = wl.forward_case(tr, lg, n, 77 + lg, frame=(wl.FRAME_W, wl.FRAME_H))
        res = np.clip(c["src"].astype(np.int32), 0, 255).astype(np.int16)  # non-negative so that clip8 is the identity
        coeffs = B.transform(B.i16(res), c["src_stride"], B.off32(c["src_off"]), tr, lg)
        p = wl.Plane(wl.FRAME_W, wl.FRAME_H, wl.Plane.frames_for(n, size, size, wl.FRAME_W, wl.FRAME_H))
        off = B.off32(p.tile_offsets(n, size, size).astype(np.uint32))
        pred = torch.zeros(p.size, dtype=torch.uint8, device="cuda")
        dst = torch.zeros(p.size, dtype=torch.uint8, device="cuda")
        B.inverse_transform_add(dst, p.stride, pred, p.stride, off, off, coeffs, tr, lg)
        torch.cuda.synchronize()
        got = cases._dst_rows(dst.cpu().numpy(), p.stride, p.tile_offsets(n, size, size), size, size).astype(np.int32)
        want = cases._dst_rows(res.view(np.int16), c["src_stride"], c["src_off"].astype(np.int64), size, size).astype(np.int32)
        assert np.abs(got - want).max() <= 4, (tr, lg, np.abs(got - want).max())


def test_quantize_roundtrip_and_cbf():
    import torch
    from hevcasm_amd import batch as B, workload as wl
    n = 1 << 22
    src = B.i16(wl.fill_i16(n, 5, "full"))
    q, cbf = B.quantize(src, 26214, 21, 171 << 7, 256)
    # cbf == 0 exactly for all-zero blocks
    nz = (q.view(-1, 256) != 0).any(dim=1)
    assert torch.equal(nz, cbf != 0)
    # the quantiser itself, whole batch, against its definition (reference quantize.c:160-186) in numpy
    x = src.cpu().numpy().astype(np.int64)
    lv = np.sign(x) * ((np.abs(x) * 26214 + ((171 << 7) << (21 - 16))) >> 21)
    assert np.array_equal(q.cpu().numpy().astype(np.int64), np.clip(lv, -32768, 32767))
    # dequantising with the matching scale (26214 / 2^21 = 1 / 80.0: step 80 = scale 160 >> 1) lands within one quantiser
    # step of the input -- (q * 160 + 1) >> 1 = 80 q, no clipping since |q| <= 410 -- and on the input's side of zero
    dq = B.quantize_inverse(q, 160, 1)
    err = (dq.to(torch.int32) - src.to(torch.int32))
    assert int(err.abs().max()) < 80, int(err.abs().max())
    assert bool((dq.to(torch.int32).abs() <= src.to(torch.int32).abs() + 27).all())      # rounding offset 0.334 of a step: at most 27 above |x|
    assert torch.equal(dq.to(torch.int32), q.to(torch.int32) * 80)


def test_pred_full_size_properties():
    """Config #3 scale (64 frames of 1080p PUs): zero fractions copy the reference block exactly;
    bi-prediction of a block with itself returns it; a sampled slice equals the oracle."""
    import torch
    from hevcasm_amd import batch as B, workload as wl
    from oracle import pyoracle
    # 64x32 at 16 frames is a small launch of tall PUs (the launcher cuts those into row segments, pred_inter.hip);
    # 8x32 at 67 frames is 16 582 waves, above the threshold, so tall PUs in one piece are covered as well
    for taps, (w, h), (fw, fh), frames in ((8, (16, 16), (1920, 1080), 16), (4, (8, 8), (960, 540), 16), (8, (64, 32), (1920, 1080), 16), (8, (8, 32), (1920, 1080), 67)):
        n = (fw // w) * (fh // h) * frames
        c = wl.pred_case(taps, w, h, n, 99, frame=(fw, fh), fracs=(0, 0))
        ref, dst_off, ref_off, frac = B.u8(c["ref0"]), B.off32(c["dst_off"]), B.off32(c["ref0_off"]), B.u8(c["frac0"])
        dst = torch.zeros(c["dst_bytes"], dtype=torch.uint8, device="cuda")
        B.pred_uni(dst, c["stride_dst"], ref, c["stride_ref"], dst_off, ref_off, frac, taps, w, h)
        torch.cuda.synchronize()
        got = cases._dst_rows(dst.cpu().numpy(), c["stride_dst"], c["dst_off"], w, h)
        want = cases._dst_rows(c["ref0"], c["stride_ref"], c["ref0_off"], w, h)
        assert np.array_equal(got, want), "zero-fraction prediction must be a copy"
        dst2 = torch.zeros_like(dst)
        B.pred_bi(dst2, c["stride_dst"], ref, ref, c["stride_ref"], dst_off, ref_off, ref_off, frac, frac, taps, w, h)
        torch.cuda.synchronize()
        assert torch.equal(dst, dst2), "bi-prediction of a block with itself at zero fraction must return it"
        # random fractions: last 200 PUs against the oracle
        c = wl.pred_case(taps, w, h, n, 98, frame=(fw, fh))
        dst = torch.zeros(c["dst_bytes"], dtype=torch.uint8, device="cuda")
        B.pred_uni(dst, c["stride_dst"], B.u8(c["ref0"]), c["stride_ref"], B.off32(c["dst_off"]), B.off32(c["ref0_off"]), B.u8(c["frac0"]), taps, w, h)
        torch.cuda.synchronize()
        a = n - 200
        want = pyoracle.pred_uni_batch(taps, c["dst_bytes"], c["stride_dst"], c["ref0"], c["stride_ref"], c["dst_off"][a:], c["ref0_off"][a:], c["frac0"][a:], w, h)
        got = dst.cpu().numpy()
        assert np.array_equal(cases._dst_rows(got, c["stride_dst"], c["dst_off"][a:], w, h), cases._dst_rows(want, c["stride_dst"], c["dst_off"][a:], w, h))


@pytest.mark.parametrize("taps", [8, 4])
def test_pred_one_1080p_frame_of_every_pu_size(taps):
    """BASELINE.json configs[2] literally: one 1920x1080 luma frame (8 taps) / one 960x540 chroma plane (4 taps) tiled by each of the 24 HEVC PU
    sizes of reference pred_inter.c:438-444 in turn (22 for bi, :687-693).  Per size: integer vectors copy the reference block exactly, bi-prediction
    of a block with itself returns it, and with fractions over all phases the LAST 200 PUs of the frame equal the oracle, uni and bi (the frame's
    tail: partial waves, partial workgroups, the strip / stream / segment launch choices of every size at its real launch size)."""
    import torch
    from hevcasm_amd import batch as B, workload as wl
    from oracle import pyoracle
    fw, fh = (1920, 1080) if taps == 8 else (960, 540)
    for (lw, lh) in wl.PU_SIZES:
        w, h = lw * taps // 8, lh * taps // 8
        n = (fw // w) * (fh // h)
        c = wl.pred_case(taps, w, h, n, 3100 + lw + 3 * lh, frame=(fw, fh), fracs=(0, 0))
        ref, dst_off, ref_off, frac = B.u8(c["ref0"]), B.off32(c["dst_off"]), B.off32(c["ref0_off"]), B.u8(c["frac0"])
        dst = torch.zeros(c["dst_bytes"], dtype=torch.uint8, device="cuda")
        B.pred_uni(dst, c["stride_dst"], ref, c["stride_ref"], dst_off, ref_off, frac, taps, w, h)
        torch.cuda.synchronize()
        assert np.array_equal(cases._dst_rows(dst.cpu().numpy(), c["stride_dst"], c["dst_off"], w, h), cases._dst_rows(c["ref0"], c["stride_ref"], c["ref0_off"], w, h)), (taps, w, h, "copy")
        bi_ok = (lw, lh) in wl.PU_SIZES_BI
        if bi_ok:
            dst2 = torch.zeros_like(dst)
            B.pred_bi(dst2, c["stride_dst"], ref, ref, c["stride_ref"], dst_off, ref_off, ref_off, frac, frac, taps, w, h)
            torch.cuda.synchronize()
            assert torch.equal(dst, dst2), (taps, w, h, "bi of a block with itself")
        a = max(0, n - 200)
        c = wl.pred_case(taps, w, h, n, 3200 + lw + 3 * lh, bi=bi_ok, frame=(fw, fh))
        dst = torch.zeros(c["dst_bytes"], dtype=torch.uint8, device="cuda")
        B.pred_uni(dst, c["stride_dst"], B.u8(c["ref0"]), c["stride_ref"], B.off32(c["dst_off"]), B.off32(c["ref0_off"]), B.u8(c["frac0"]), taps, w, h)
        torch.cuda.synchronize()
        want = pyoracle.pred_uni_batch(taps, c["dst_bytes"], c["stride_dst"], c["ref0"], c["stride_ref"], c["dst_off"][a:], c["ref0_off"][a:], c["frac0"][a:], w, h)
        assert np.array_equal(cases._dst_rows(dst.cpu().numpy(), c["stride_dst"], c["dst_off"][a:], w, h), cases._dst_rows(want, c["stride_dst"], c["dst_off"][a:], w, h)), (taps, w, h, "uni")
        if bi_ok:
            dst = torch.zeros(c["dst_bytes"], dtype=torch.uint8, device="cuda")
            B.pred_bi(dst, c["stride_dst"], B.u8(c["ref0"]), B.u8(c["ref1"]), c["stride_ref"], B.off32(c["dst_off"]), B.off32(c["ref0_off"]), B.off32(c["ref1_off"]),
                      B.u8(c["frac0"]), B.u8(c["frac1"]), taps, w, h)
            torch.cuda.synchronize()
            want = pyoracle.pred_bi_batch(taps, c["dst_bytes"], c["stride_dst"], c["ref0"], c["ref1"], c["stride_ref"], c["dst_off"][a:], c["ref0_off"][a:], c["ref1_off"][a:],
                                          c["frac0"][a:], c["frac1"][a:], w, h)
            assert np.array_equal(cases._dst_rows(dst.cpu().numpy(), c["stride_dst"], c["dst_off"][a:], w, h), cases._dst_rows(want, c["stride_dst"], c["dst_off"][a:], w, h)), (taps, w, h, "bi")


def test_pred_two_planes_equal_two_single_plane_calls():
    """Cb and Cr in one launch (hevcasm_hip_pred_*_batch_2planes: one PU list, two planes of the destination and of each
    reference) give exactly the bytes of two single-plane calls, which the cases above compare with the oracle; plane a
    also goes against the oracle directly.  Every chroma PU size (incl. the 2- and 6-wide ones), luma for good measure,
    a launch small enough to be cut into row segments, and planes at odd distances from each other."""
    import torch
    from hevcasm_amd import batch as B, workload as wl
    from oracle import pyoracle
    todo = [(4, w // 2, h // 2, 40) for (w, h) in wl.PU_SIZES] + [(8, 16, 16, 50), (8, 64, 64, 8), (8, 4, 8, 33)]
    for taps, w, h, n in todo:
        for bi in (False, True):
            if bi and (w, h) in ((4, 2), (2, 4)):
                continue
            ca = wl.pred_case(taps, w, h, n, 1300 + w + 7 * h, bi=bi)
            cb = wl.pred_case(taps, w, h, n, 2300 + w + 7 * h, bi=bi)      # same geometry (offsets depend on the seed only through the motion vectors: use ca's)
            pad = 3                                                        # plane b starts 3 bytes into its buffer: plane distances are not multiples of anything
            def dev(x, shift=0):
                t = torch.zeros(x.size + 8, dtype=torch.uint8, device="cuda")
                t[shift:shift + x.size] = torch.from_numpy(x).cuda()
                return t[shift:shift + x.size]
            ra0, rb0 = dev(ca["ref0"]), dev(cb["ref0"], pad)
            doff, r0off, f0 = B.off32(ca["dst_off"]), B.off32(ca["ref0_off"]), B.u8(ca["frac0"])
            da, db = torch.zeros(ca["dst_bytes"], dtype=torch.uint8, device="cuda"), torch.zeros(ca["dst_bytes"] + 8, dtype=torch.uint8, device="cuda")[5:5 + ca["dst_bytes"]]
            wa, wb = torch.zeros_like(da), torch.zeros(ca["dst_bytes"], dtype=torch.uint8, device="cuda")
            if bi:
                ra1, rb1 = dev(ca["ref1"]), dev(cb["ref1"], 1)
                r1off, f1 = B.off32(ca["ref1_off"]), B.u8(ca["frac1"])
                B.pred_bi_2planes(da, db, ca["stride_dst"], ra0, rb0, ra1, rb1, ca["stride_ref"], doff, r0off, r1off, f0, f1, taps, w, h)
                B.pred_bi(wa, ca["stride_dst"], ra0, ra1, ca["stride_ref"], doff, r0off, r1off, f0, f1, taps, w, h)
                B.pred_bi(wb, ca["stride_dst"], rb0, rb1, ca["stride_ref"], doff, r0off, r1off, f0, f1, taps, w, h)
                want = pyoracle.pred_bi_batch(taps, ca["dst_bytes"], ca["stride_dst"], ca["ref0"], ca["ref1"], ca["stride_ref"], ca["dst_off"], ca["ref0_off"], ca["ref1_off"], ca["frac0"], ca["frac1"], w, h)
            else:
                B.pred_uni_2planes(da, db, ca["stride_dst"], ra0, rb0, ca["stride_ref"], doff, r0off, f0, taps, w, h)
                B.pred_uni(wa, ca["stride_dst"], ra0, ca["stride_ref"], doff, r0off, f0, taps, w, h)
                B.pred_uni(wb, ca["stride_dst"], rb0, ca["stride_ref"], doff, r0off, f0, taps, w, h)
                want = pyoracle.pred_uni_batch(taps, ca["dst_bytes"], ca["stride_dst"], ca["ref0"], ca["stride_ref"], ca["dst_off"], ca["ref0_off"], ca["frac0"], w, h)
            torch.cuda.synchronize()
            assert torch.equal(da, wa) and torch.equal(db, wb), (taps, w, h, bi)
            assert np.array_equal(cases._dst_rows(da.cpu().numpy(), ca["stride_dst"], ca["dst_off"], w, h), cases._dst_rows(want, ca["stride_dst"], ca["dst_off"], w, h)), (taps, w, h, bi)


def test_full_search_finds_planted_block():
    """Config #5 geometry (64x64 CTUs, +-64) on a frame whose search windows overlap as they do in a
    real frame: a CTU copied into its own window at a known displacement is found with cost 0 at the
    raster-first zero-cost position, which is the planted one unless an earlier candidate is also a
    perfect match (checked by recomputing that candidate's SAD); every CTU equals the oracle."""
    import torch
    from hevcasm_amd import batch as B, workload as wl
    from oracle import pyoracle
    fw, fh, r, w, h = 1024, 512, 64, 64, 64
    rng = np.random.default_rng(5)
    src = wl.fill_u8(fw * fh, 1)
    stride_r = fw + 2 * r
    ref = wl.fill_u8(stride_r * (fh + 2 * r), 2)
    by, bx = np.meshgrid(np.arange(fh // h), np.arange(fw // w), indexing="ij")
    so = (by * h * fw + bx * w).reshape(-1)
    ro = ((by * h + r) * stride_r + bx * w + r).reshape(-1)
    # plant every other CTU; windows of neighbouring CTUs overlap, so a later planting may overwrite an earlier one
    planted = {}
    for i in range(0, so.size, 2):
        dx, dy = (int(v) for v in rng.integers(-r, r + 1, 2))
        blk = cases._dst_rows(src, fw, so[i:i + 1], w, h)[0]
        base = int(ro[i]) + dy * stride_r + dx
        for y in range(h):
            ref[base + y * stride_r: base + y * stride_r + w] = blk[y]
        planted[i] = (dx, dy)
    out = B.me_full_search(B.u8(src), fw, B.u8(ref), stride_r, B.off32(so.astype(np.uint32)), B.off32(ro.astype(np.uint32)), w, h, r)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    hits = 0
    for i, (dx, dy) in planted.items():
        blk = cases._dst_rows(src, fw, so[i:i + 1], w, h)[0]
        still = np.array_equal(cases._dst_rows(ref, stride_r, np.array([int(ro[i]) + dy * stride_r + dx]), w, h)[0], blk)
        if still:   # not overwritten by a later neighbour's planting
            assert out[i, 2] == 0, (i, out[i], dx, dy)
            gx, gy = int(out[i, 0]), int(out[i, 1])
            assert (gy, gx) <= (dy, dx), (i, out[i], dx, dy)            # raster order of (dy, dx): never after the planted one
            if (gy, gx) != (dy, dx):                                        # an earlier perfect match must really be one
                assert np.array_equal(cases._dst_rows(ref, stride_r, np.array([int(ro[i]) + gy * stride_r + gx]), w, h)[0], blk)
            hits += 1
    assert hits >= len(planted) // 2
    want = pyoracle.me_full_search(src, fw, ref, stride_r, so.astype(np.uint32), ro.astype(np.uint32), w, h, r)
    cases.assert_same("me 1024x512 +-64", (out,), (want,))


def test_full_search_4k_frame_equals_oracle():
    """BASELINE.json config #5 at its full size: one 3840x2160 luma frame, 60 x 34 CTUs (33 rows of
    64x64 and a bottom row of 64x48, reference sad.c:231-232), +-64 around every CTU, the reference
    frame padded by 64 replicated pixels.  Three ways to the same table, all compared over the WHOLE
    frame: the oracle (hevcasm_sad, sad.c:47-60, per candidate, raster-first minimum: 33.9 M SAD
    evaluations on video-like content, non-zero minima everywhere); the block-list entry point (two
    launches, one per CTU shape); the frame-rows entry point, unsharded (one launch) and as the 8 ranks
    of config #5 would run it (each on its own slab = its CTU rows + the +-64-row halo, uploaded
    separately; the 34 rows split 5,5,4,4,4,4,4,4 so the 64x48 row is the last rank's)."""
    import torch
    from hevcasm_amd import batch as B, me_frame
    from oracle import pyoracle
    f = me_frame.synthetic_frame_pair(3840, 2160, 64, seed=77)
    jobs = me_frame.ctu_jobs(2160, 3840, 64, f["stride_cur"], f["stride_ref"], 64, 0, 34)
    assert sorted((j["w"], j["h"], len(j["index"])) for j in jobs) == [(64, 48, 60), (64, 64, 1980)]
    cur_d, ref_d = B.u8(f["cur"]), B.u8(f["ref"])
    want = np.zeros((2040, 3), dtype=np.int32)
    listed = np.zeros((2040, 3), dtype=np.int32)
    for j in jobs:
        want[j["index"]] = pyoracle.me_full_search(f["cur"], f["stride_cur"], f["ref"], f["stride_ref"], j["cur_off"], j["ref_off"], 64, j["h"], 64)
        listed[j["index"]] = B.me_full_search(cur_d, f["stride_cur"], ref_d, f["stride_ref"], B.off32(j["cur_off"]), B.off32(j["ref_off"]), 64, j["h"], 64).cpu().numpy()
    assert (want[:, 2] > 0).mean() > 0.9          # the comparison is about non-trivial minima ...
    assert len(set(map(tuple, want[:, :2]))) > 1000   # ... at many different displacements
    cases.assert_same("me 4K frame, block lists", (listed,), (want,))
    whole = B.me_full_search_frame_rows(cur_d, 0, f["stride_cur"], ref_d, 64 * f["stride_ref"] + 64, f["stride_ref"], 3840, 2160, 64, 64, 0, 34).cpu().numpy()
    cases.assert_same("me 4K frame, one launch", (whole,), (want,))
    fn = me_frame.hip_rows_fn()
    parts = [me_frame.search_rank(f, 64, 64, rank, 8, fn) for rank in range(8)]
    assert [p[1] - p[0] for p in parts] == [5, 5, 4, 4, 4, 4, 4, 4]
    cases.assert_same("me 4K frame, 8 slabs", (me_frame.assemble(parts, 3840, 2160, 64),), (want,))
    # and split by CTUs (255 per rank: runs that begin and end inside a row, launches small enough to be cut into parts)
    fn = me_frame.hip_ctus_fn()
    parts = [me_frame.search_rank_ctus(f, 64, 64, rank, 8, fn) for rank in range(8)]
    assert [p[1] - p[0] for p in parts] == [255] * 8
    cases.assert_same("me 4K frame, 8 runs of 255 CTUs", (me_frame.assemble_ctus(parts, 3840, 2160, 64),), (want,))


def test_full_search_result_buffer_need_only_be_int_aligned():
    """Small launches merge several workgroups' minima with a 64-bit atomic inside each block's 12-byte result slot: the slot's
    aligned 8 bytes are found from the ADDRESS, so a result array that starts on an odd int (a view into a larger buffer) works."""
    import torch
    from hevcasm_amd import batch as B
    c = [x for x in cases.me_cases() if x[0] == "me_64x64_r64_corners"][0][2]
    want = cases.run_oracle("me", c)[0]
    n = c["n"]
    for shift in (0, 1, 2, 3):
        big = torch.full((3 * n + 8,), 77, dtype=torch.int32, device="cuda")
        out = big[shift: shift + 3 * n].view(n, 3)
        B.me_full_search(B.u8(c["src"]), c["stride_src"], B.u8(c["ref"]), c["stride_ref"], B.off32(c["src_off"]), B.off32(c["ref_off"]), c["w"], c["h"], c["range"], out=out)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want), shift
        assert int(big[:shift].eq(77).all()) and int(big[shift + 3 * n:].eq(77).all())        # nothing written beside the results


def test_full_search_frame_rows_ragged_edges():
    """Frames whose width and height are not multiples of the CTU (right column, bottom row and the corner CTU are
    narrower / shorter), ranges that are not multiples of 16, more ranks than CTU rows: every split equals the oracle."""
    from hevcasm_amd import batch as B, me_frame
    from oracle import pyoracle
    for (fw, fh, ctu, r, world) in ((200, 176, 64, 24, 3), (96, 80, 32, 37, 2), (64, 48, 64, 64, 4), (128, 128, 16, 9, 5), (72, 40, 64, 16, 1),
                                    (16, 16, 4, 0, 2), (4, 4, 64, 3, 1), (260, 12, 64, 70, 2), (36, 300, 20, 100, 7)):
        f = me_frame.synthetic_frame_pair(fw, fh, r, seed=fw + r, ctu=ctu, max_motion=min(r, 20))
        rows, cols = -(-fh // ctu), -(-fw // ctu)
        want = np.zeros((rows * cols, 3), dtype=np.int32)
        for j in me_frame.ctu_jobs(fh, fw, ctu, f["stride_cur"], f["stride_ref"], r, 0, rows):
            want[j["index"]] = pyoracle.me_full_search(f["cur"], f["stride_cur"], f["ref"], f["stride_ref"], j["cur_off"], j["ref_off"], j["w"], j["h"], r)
        fn = me_frame.hip_rows_fn()
        parts = [me_frame.search_rank(f, ctu, r, rank, world, fn) for rank in range(world)]
        cases.assert_same("me frame %dx%d ctu %d +-%d x%d" % (fw, fh, ctu, r, world), (me_frame.assemble(parts, fw, fh, ctu),), (want,))
        fn = me_frame.hip_ctus_fn()
        parts = [me_frame.search_rank_ctus(f, ctu, r, rank, world + 1, fn) for rank in range(world + 1)]
        cases.assert_same("me frame %dx%d ctu %d +-%d, %d runs" % (fw, fh, ctu, r, world + 1), (me_frame.assemble_ctus(parts, fw, fh, ctu),), (want,))


def test_subpel_refine_equals_unfused_composition():
    """Full 1080p frame of 16x16 blocks: the fused kernel's (qx, qy, cost) must be the raster-first
    arg-min of the 49 costs obtained the long way round -- pred_uni_8to8_batch into a scratch plane,
    then sad_batch -- through kernels that are themselves parity-tested against the oracle."""
    import torch
    from hevcasm_amd import batch as B, workload as wl
    fw, fh, w, h, m = 1920, 1088, 16, 16, 16
    stride_r = fw + 2 * m
    src = wl.fill_u8(fw * fh, 21, device="cuda")
    ref = wl.fill_u8(stride_r * (fh + 2 * m), 22, device="cuda")
    by, bx = np.meshgrid(np.arange(fh // h), np.arange(fw // w), indexing="ij")
    so = (by * h * fw + bx * w).reshape(-1).astype(np.uint32)
    ro = ((by * h + m) * stride_r + bx * w + m).reshape(-1).astype(np.int64)
    n = so.size
    so_d = B.off32(so)
    got = B.subpel_refine(src, fw, ref, stride_r, so_d, B.off32(ro.astype(np.uint32)), w, h).cpu().numpy()
    scratch = torch.zeros(fw * fh, dtype=torch.uint8, device="cuda")
    costs = np.empty((49, n), dtype=np.int64)
    for qy in range(-3, 4):
        for qx in range(-3, 4):
            off = B.off32((ro + (qy >> 2) * stride_r + (qx >> 2)).astype(np.uint32))
            frac = torch.full((n,), (qx & 3) | ((qy & 3) << 4), dtype=torch.uint8, device="cuda")
            B.pred_uni(scratch, fw, ref, stride_r, so_d, off, frac, 8, w, h)
            costs[(qy + 3) * 7 + qx + 3] = B.sad(src, fw, scratch, fw, so_d, so_d, w, h).cpu().numpy()
    best = costs.argmin(axis=0)       # numpy returns the first minimum: raster order of (qy, qx)
    assert np.array_equal(got[:, 2], costs.min(axis=0))
    assert np.array_equal(got[:, 0], best % 7 - 3) and np.array_equal(got[:, 1], best // 7 - 3)


def test_sad_misaligned_plane_base():
    """The 4-way kernel rounds the reference base down to a dword and works with offsets from there:
    a plane pointer that is itself byte-misaligned (a view into a larger buffer) must give the same
    sums as the aligned copy.  All three misalignments, 1-way and 4-way, wide and narrow blocks."""
    import torch
    from hevcasm_amd import batch as B, workload as wl
    from oracle import pyoracle
    for (w, h) in ((64, 64), (48, 64), (32, 32), (16, 16), (8, 8)):
        c = wl.sad_case(w, h, 50, 700 + w, ways=4)
        want4 = pyoracle.sad_multiref_batch(c["src"], c["stride_src"], c["ref"], c["stride_ref"], c["src_off"], c["ref_off"], w, h)
        want1 = pyoracle.sad_batch(c["src"], c["stride_src"], c["ref"], c["stride_ref"], c["src_off"], c["ref_off"].reshape(-1, 4)[:, 0].copy(), w, h)
        so, ro4 = B.off32(c["src_off"]), B.off32(c["ref_off"])
        ro1 = B.off32(c["ref_off"].reshape(-1, 4)[:, 0].copy())
        for k in (1, 2, 3):
            big_r = torch.zeros(c["ref"].size + 8, dtype=torch.uint8, device="cuda")
            big_s = torch.zeros(c["src"].size + 8, dtype=torch.uint8, device="cuda")
            big_r[k:k + c["ref"].size] = torch.from_numpy(c["ref"]).cuda()
            big_s[k:k + c["src"].size] = torch.from_numpy(c["src"]).cuda()
            ref_v, src_v = big_r[k:], big_s[k:]
            assert ref_v.data_ptr() % 4 == k
            got4 = B.sad_multiref(src_v, c["stride_src"], ref_v, c["stride_ref"], so, ro4, w, h).cpu().numpy().reshape(-1, 4)
            got1 = B.sad(src_v, c["stride_src"], ref_v, c["stride_ref"], so, ro1, w, h).cpu().numpy()
            assert np.array_equal(got4, want4), (w, h, k)
            assert np.array_equal(got1, want1), (w, h, k)


def test_sad_narrow_blocks_bottom_up_and_tight_planes():
    """The narrow-block kernels load reference rows as aligned words wider than the row piece (sad.hip, OverlapRefChunk):
    positive and negative strides, displacement ranges 0 / 4 / 7, and planes that hold exactly the blocks with rows
    8 bytes apart (nothing around them to over-read) or 4 bytes apart (overlapping rows: the byte-address kernels)."""
    from hevcasm_amd import workload as wl
    for (w, h) in ((8, 8), (8, 16), (8, 32), (8, 4), (8, 64), (12, 16), (20, 8), (4, 8)):
        for ways in (1, 4):
            fam = "sad4" if ways == 4 else "sad"
            for md in (0, 4, 7):
                c = wl.sad_case(w, h, 37, 1000 + w + h + md, ways=ways, max_disp=md)
                cases.assert_same("%dx%d x%d disp %d" % (w, h, ways, md), cases.run_hip(fam, c), cases.run_oracle(fam, c))
                c2 = dict(c)
                c2["src_off"] = (c["src_off"].astype(np.int64) + (h - 1) * c["stride_src"]).astype(np.uint32)
                c2["ref_off"] = (c["ref_off"].astype(np.int64) + (h - 1) * c["stride_ref"]).astype(np.uint32)
                c2["stride_src"], c2["stride_ref"] = -c["stride_src"], -c["stride_ref"]
                cases.assert_same("bottom-up %dx%d x%d disp %d" % (w, h, ways, md), cases.run_hip(fam, c2), cases.run_oracle(fam, c2))
    for h in (4, 8, 16, 32):
        for ways in (1, 4):
            fam = "sad4" if ways == 4 else "sad"
            n = 9
            src, ref = wl.fill_u8(8 * h * n, 77), wl.fill_u8(8 * h * n, 78)
            so = (np.arange(n) * 8 * h).astype(np.uint32)
            c = dict(src=src, stride_src=8, ref=ref, stride_ref=8, src_off=so, ref_off=np.repeat(so, ways).astype(np.uint32), w=8, h=h, n=n, ways=ways)
            cases.assert_same("tight 8x%d x%d" % (h, ways), cases.run_hip(fam, c), cases.run_oracle(fam, c))
            ref2 = wl.fill_u8(4 * h * n + 8, 79)
            c = dict(src=src, stride_src=8, ref=ref2, stride_ref=4, src_off=so, ref_off=np.repeat((np.arange(n) * 4 * h).astype(np.uint32), ways), w=8, h=h, n=n, ways=ways)
            cases.assert_same("rows 4 bytes apart 8x%d x%d" % (h, ways), cases.run_hip(fam, c), cases.run_oracle(fam, c))
    # 12-wide blocks are one lane of 12 bytes per row with aligned 16-byte loads (round 2): planes that hold exactly the blocks, rows 12 bytes
    # apart (every over-read must stay inside the block), 16 apart, and 8 apart (overlapping rows: the 4-byte-chunk kernels at the byte
    # address); 24-wide (two lanes of 12) and 48-wide (four lanes of 12, one reference) with rows exactly W apart; both stride signs
    for (w, tight) in ((12, 12), (12, 16), (12, 8), (24, 24), (48, 48), (32, 32), (16, 16)):
        for h in (4, 8, 16, 32, 48, 64):
            for ways in (1, 4):
                fam = "sad4" if ways == 4 else "sad"
                n = 7
                src = wl.fill_u8(w * h * n, 81)
                ref = wl.fill_u8(tight * (h - 1) * n + tight * n + w + 64, 82)[:tight * h * n + (w - tight if w > tight else 0)]
                so = (np.arange(n) * w * h).astype(np.uint32)
                ro = np.repeat((np.arange(n) * tight * h).astype(np.uint32), ways)
                c = dict(src=src, stride_src=w, ref=ref, stride_ref=tight, src_off=so, ref_off=ro, w=w, h=h, n=n, ways=ways)
                cases.assert_same("tight %dx%d rows %d apart x%d" % (w, h, tight, ways), cases.run_hip(fam, c), cases.run_oracle(fam, c))
                c2 = dict(c)
                c2["src_off"] = (so.astype(np.int64) + (h - 1) * w).astype(np.uint32)
                c2["ref_off"] = (ro.astype(np.int64) + (h - 1) * tight).astype(np.uint32)
                c2["stride_src"], c2["stride_ref"] = -w, -tight
                cases.assert_same("tight bottom-up %dx%d rows %d apart x%d" % (w, h, tight, ways), cases.run_hip(fam, c2), cases.run_oracle(fam, c2))


def test_sad_offsets_at_the_top_of_the_4gib_window():
    """The SAD kernels address with base + 32-bit offset: blocks whose bytes end just below 4 GiB - 64 must come out
    right (no wrap), both planes, 1-way and 4-way, negative stride included."""
    import torch
    from oracle import pyoracle
    from hevcasm_amd import batch as B
    size = (1 << 32) - 64
    tail = 1 << 20                                   # the last MiB of each plane carries the data
    rng = np.random.default_rng(77)
    src_t, ref_t = rng.integers(0, 256, tail, dtype=np.uint8), rng.integers(0, 256, tail, dtype=np.uint8)
    src = torch.zeros(size, dtype=torch.uint8, device="cuda")
    ref = torch.zeros(size, dtype=torch.uint8, device="cuda")
    src[size - tail:] = torch.from_numpy(src_t).cuda()
    ref[size - tail:] = torch.from_numpy(ref_t).cuda()
    stride = 2048
    for (w, h) in ((64, 64), (16, 16), (48, 64)):
        n = 64
        rows_avail = tail // stride - h - 16
        so_t = (rng.integers(8, rows_avail, n) * stride + rng.integers(0, stride - w - 16, n)).astype(np.int64)
        ro_t = (so_t[:, None] + rng.integers(-7, 8, (n, 4)) * stride + rng.integers(-7, 8, (n, 4))).astype(np.int64)
        # the very last block ends at the plane's last byte
        so_t[0] = tail - (h - 1) * stride - w
        ro_t[0, :] = tail - (h - 1) * stride - w - np.array([0, 1, stride, stride + 3])
        want4 = pyoracle.sad_multiref_batch(src_t, stride, ref_t, stride, so_t.astype(np.uint32), ro_t.reshape(-1).astype(np.uint32), w, h)
        base = size - tail
        so_d = B.off32((so_t + base).astype(np.uint32))
        ro_d = B.off32((ro_t + base).reshape(-1).astype(np.uint32))
        got4 = B.sad_multiref(src, stride, ref, stride, so_d, ro_d, w, h).cpu().numpy().reshape(-1, 4)
        assert np.array_equal(got4, want4), (w, h)
        got1 = B.sad(src, stride, ref, stride, so_d, B.off32((ro_t[:, 1] + base).astype(np.uint32)), w, h).cpu().numpy()
        assert np.array_equal(got1, want4[:, 1]), (w, h)
        # bottom-up: offsets name the first row, which is the highest address
        so_n = (so_t + (h - 1) * stride + base).astype(np.uint32)
        ro_n = (ro_t + (h - 1) * stride + base).reshape(-1).astype(np.uint32)
        want_n = pyoracle.sad_multiref_batch(src_t, -stride, ref_t, -stride, (so_t + (h - 1) * stride).astype(np.uint32),
                                             (ro_t + (h - 1) * stride).reshape(-1).astype(np.uint32), w, h)
        got_n = B.sad_multiref(src, -stride, ref, -stride, B.off32(so_n), B.off32(ro_n), w, h).cpu().numpy().reshape(-1, 4)
        assert np.array_equal(got_n, want_n), (w, h)
    del src, ref
    torch.cuda.empty_cache()


def test_concurrent_host_threads_and_streams():
    """hevcasm_hip.h promises that the entry points may be called from several host threads at once (one
    stream per thread): four threads, each on its own stream, each a different family, many rounds."""
    import threading
    import torch
    from hevcasm_amd import batch as B, workload as wl
    from oracle import pyoracle
    c_sad = wl.sad_case(16, 16, 4000, 801, ways=4)
    c_inv = wl.inverse_case(0, 4, 500, 802, "natural")
    c_pred = wl.pred_case(8, 16, 16, 400, 803)
    want = dict(
        sad4=pyoracle.sad_multiref_batch(c_sad["src"], c_sad["stride_src"], c_sad["ref"], c_sad["stride_ref"], c_sad["src_off"], c_sad["ref_off"], 16, 16),
        inv=cases.run_oracle("inverse", c_inv)[0], pred=cases.run_oracle("pred_uni", c_pred)[0],
        sub=cases.run_oracle("subpel", [x for x in cases.subpel_cases() if x[0] == "subpel_16x16"][0][2])[0])
    errors = []

    def worker(kind):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for _ in range(20):
                    if kind == "sad4":
                        got = cases.run_hip("sad4", c_sad)[0]
                    elif kind == "inv":
                        got = cases.run_hip("inverse", c_inv)[0]
                    elif kind == "pred":
                        got = cases.run_hip("pred_uni", c_pred)[0]
                    else:
                        got = cases.run_hip("subpel", [x for x in cases.subpel_cases() if x[0] == "subpel_16x16"][0][2])[0]
                    if not np.array_equal(np.asarray(got).reshape(-1), np.asarray(want[kind]).reshape(-1)):
                        errors.append(kind)
                        return
        except Exception as e:   # noqa: BLE001 -- reported through the list
            errors.append("%s: %r" % (kind, e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in ("sad4", "inv", "pred", "sub")]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_entry_points_capture_into_a_hip_graph():
    """The batch entry points only enqueue work on the caller's stream, so a launch-bound inner loop can be
    captured once into a HIP graph and replayed: SAD + 4-way SAD + inverse transform captured together, inputs
    overwritten in place between replays, every replay checked against the oracle."""
    import torch
    from hevcasm_amd import batch as B, workload as wl
    from oracle import pyoracle
    n = 3000
    ca = wl.sad_case(16, 16, n, 901, ways=4)
    cb = wl.sad_case(16, 16, n, 902, ways=4)          # same geometry, other pixels and displacements
    src, ref = B.u8(ca["src"]).clone(), B.u8(ca["ref"]).clone()
    so, ro = B.off32(ca["src_off"]).clone(), B.off32(ca["ref_off"]).clone()
    ro1 = ro[::4].contiguous()
    out1 = torch.zeros(n, dtype=torch.int32, device="cuda")
    out4 = torch.zeros(4 * n, dtype=torch.int32, device="cuda")
    ci = wl.inverse_case(0, 3, 2000, 903, "natural")
    coeffs = B.i16(ci["coeffs"]).clone()
    pred, dst = B.u8(ci["pred"]), torch.zeros(ci["dst_bytes"], dtype=torch.uint8, device="cuda")
    doff, poff = B.off32(ci["dst_off"]), B.off32(ci["pred_off"])
    want_inv = np.asarray(cases.run_oracle("inverse", ci)[0])
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):      # warm-up outside capture (the first launch loads the code object)
        B.sad(src, ca["stride_src"], ref, ca["stride_ref"], so, ro1, 16, 16, out=out1)
        B.sad_multiref(src, ca["stride_src"], ref, ca["stride_ref"], so, ro, 16, 16, out=out4)
        B.inverse_transform_add(dst, ci["stride_dst"], pred, ci["stride_pred"], doff, poff, coeffs, 0, 3)
    torch.cuda.current_stream().wait_stream(stream)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        B.sad(src, ca["stride_src"], ref, ca["stride_ref"], so, ro1, 16, 16, out=out1)
        B.sad_multiref(src, ca["stride_src"], ref, ca["stride_ref"], so, ro, 16, 16, out=out4)
        B.inverse_transform_add(dst, ci["stride_dst"], pred, ci["stride_pred"], doff, poff, coeffs, 0, 3)
    for c in (ca, cb, ca):
        src.copy_(B.u8(c["src"])); ref.copy_(B.u8(c["ref"])); ro.copy_(B.off32(c["ref_off"])); ro1.copy_(ro[::4])
        out1.zero_(); out4.zero_(); dst.zero_()
        graph.replay()
        torch.cuda.synchronize()
        want4 = np.asarray(pyoracle.sad_multiref_batch(c["src"], c["stride_src"], c["ref"], c["stride_ref"], c["src_off"], c["ref_off"], 16, 16)).reshape(-1)
        assert np.array_equal(out4.cpu().numpy().reshape(-1), want4)
        assert np.array_equal(out1.cpu().numpy(), want4.reshape(-1, 4)[:, 0])
        assert np.array_equal(cases._dst_rows(dst.cpu().numpy(), ci["stride_dst"], ci["dst_off"], 8, 8), want_inv)


def test_build_then_smoke_in_one_process():
    """The driver may call build() and smoke() from one interpreter: the library must not be loaded before torch (two HIP
    runtimes in one process leave the second without devices; capi.lib() orders the loads)."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "smoke OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_error_codes_not_fallbacks():
    """Arguments outside the reference's domain are refused, never silently computed."""
    import torch
    from hevcasm_amd import capi
    L = capi.lib()
    t = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    o = torch.zeros(16, dtype=torch.int32, device="cuda")
    p = t.data_ptr()
    assert L.hevcasm_hip_sad_batch(p, 64, p, 64, o.data_ptr(), o.data_ptr(), o.data_ptr(), capi.rect(10, 16), 1, None) == capi.EINVAL
    assert L.hevcasm_hip_sad_batch(p, 64, p, 64, o.data_ptr(), o.data_ptr(), o.data_ptr(), capi.rect(16, 68), 1, None) == capi.EINVAL
    assert L.hevcasm_hip_inverse_transform_add_batch(p, 64, p, 64, o.data_ptr(), o.data_ptr(), p, 1, 3, 1, None) == capi.EINVAL
    assert L.hevcasm_hip_quantize_batch(p, p, 0x8000, 20, 14, 16, 1, o.data_ptr(), None) == capi.EINVAL
    assert L.hevcasm_hip_quantize_batch(p, p, -1, 20, 14, 16, 1, o.data_ptr(), None) == capi.EINVAL      # negative scale / offset: refused,
    assert L.hevcasm_hip_quantize_batch(p, p, 51, 20, -14, 16, 1, o.data_ptr(), None) == capi.EINVAL     # not computed differently from the C body
    assert L.hevcasm_hip_pred_uni_8to8_batch(p, 64, p, 64, o.data_ptr(), o.data_ptr(), p, 6, 8, 8, 1, None) == capi.EINVAL
    assert b"hevcasm_hip error" in L.hevcasm_hip_last_error()
    # the round-3 entry points: frame-rows search (CTU size, frame dimensions, rows outside the frame, a window that does not fit LDS), two planes
    fr = L.hevcasm_hip_me_full_search_frame_rows
    assert fr(p, 64, p, 64, 64, 64, 10, 8, 0, 1, o.data_ptr(), None) == capi.EINVAL
    assert fr(p, 64, p, 64, 62, 64, 16, 8, 0, 1, o.data_ptr(), None) == capi.EINVAL
    assert fr(p, 64, p, 64, 64, 64, 16, 8, 0, 5, o.data_ptr(), None) == capi.EINVAL      # 4 CTU rows only
    assert fr(p, 64, p, 64, 64, 64, 16, 8, 3, 2, o.data_ptr(), None) == capi.EINVAL
    assert fr(p, 64, p, 64, 64, 64, 64, 127, 0, 1, o.data_ptr(), None) == capi.EINVAL    # (64 + 254)^2 bytes of window: not in 64 KiB
    assert fr(p, 64, p, 64, 64, 64, 16, 8, 2, 2, o.data_ptr(), None) == 0                # an idle rank's empty range
    assert fr(None, 64, p, 64, 64, 64, 16, 8, 0, 1, o.data_ptr(), None) == capi.EINVAL
    assert L.hevcasm_hip_pred_uni_8to8_batch_2planes(p, None, 64, p, p, 64, o.data_ptr(), o.data_ptr(), p, 4, 8, 8, 1, None) == capi.EINVAL
    assert L.hevcasm_hip_pred_uni_8to8_batch_2planes(p, p, 64, p, p, 64, o.data_ptr(), o.data_ptr(), p, 4, 40, 8, 1, None) == capi.EINVAL
    # the round-5 entry points: frame SSE (dimensions, the 8-byte alignment of the sums), intra (mode range, edge flag at 32x32, missing mode array)
    o64 = torch.zeros(4, dtype=torch.int64, device="cuda")
    fs = L.hevcasm_hip_frame_sse_batch
    assert fs(p, 64, p, 64, o.data_ptr(), o.data_ptr(), o64.data_ptr(), 40000, 4, 1, None) == capi.EINVAL
    assert fs(p, 64, p, 64, o.data_ptr(), o.data_ptr(), o64.data_ptr(), 16, -1, 1, None) == capi.EINVAL
    assert fs(p, 64, p, 64, o.data_ptr(), o.data_ptr(), o64.data_ptr() + 4, 16, 4, 1, None) == capi.EINVAL
    assert fs(p, 64, p, 64, o.data_ptr(), o.data_ptr(), o64.data_ptr(), 16, 4, 0, None) == 0
    assert L.hevcasm_hip_pred_intra_batch(p, p, 144, 35, (8 << 8) | (3 << 1), 1, None) == capi.EINVAL
    assert L.hevcasm_hip_pred_intra_batch(p, p, 144, 26, (8 << 8) | (5 << 1) | 1, 1, None) == capi.EINVAL
    assert L.hevcasm_hip_pred_intra_modes_batch(p, p, 144, None, (8 << 8) | (3 << 1), 1, None) == capi.EINVAL
    assert L.hevcasm_hip_intra_filter_neighbours_batch(p, 144, p + 2048, 144, 6, 0, 1, None) == capi.EINVAL
    # round 6 (ADVICE): output records closer together than they are long are refused; a mode byte above 34 is mode 34, not an index past the tables
    assert L.hevcasm_hip_intra_filter_neighbours_batch(p, 100, p + 2048, 144, 3, 0, 2, None) == capi.EINVAL
    assert L.hevcasm_hip_intra_filter_neighbours_batch(p, 33, p + 2048, 33, 3 | 0x10000, 0, 2, None) == 0       # compact 8x8 records: 4N + 1 = 33
    nb = torch.randint(0, 256, (144,), dtype=torch.uint8, device="cuda")
    for log2 in (2, 3, 4, 5):
        outs = []
        for mode in (34, 200):
            m = torch.tensor([mode], dtype=torch.uint8, device="cuda")
            d = torch.zeros(1 << (2 * log2), dtype=torch.uint8, device="cuda")
            assert L.hevcasm_hip_pred_intra_modes_batch(d.data_ptr(), nb.data_ptr(), 144, m.data_ptr(), (8 << 8) | (log2 << 1), 1, None) == 0
            outs.append(d.cpu())
        assert torch.equal(outs[0], outs[1]), log2
    # empty batches are legal no-ops
    assert L.hevcasm_hip_sad_batch(p, 64, p, 64, o.data_ptr(), o.data_ptr(), o.data_ptr(), capi.rect(16, 16), 0, None) == 0
    assert L.hevcasm_hip_pred_bi_8to8_batch_2planes(p, p, 64, p, p, p, p, 64, o.data_ptr(), o.data_ptr(), o.data_ptr(), p, p, 8, 16, 16, 0, None) == 0


def test_in_place_reconstruction():
    """dst == pred with the same offsets and stride (include/hevcasm_hip.h promises it): inverse transform + add of every
    size, the fused dequantising form and hevcasm_quantize_reconstruct write over the prediction they read."""
    import torch
    from hevcasm_amd import batch as B, workload as wl
    for tr_type, log2n in ((1, 2), (0, 2), (0, 3), (0, 4), (0, 5)):
        c = wl.inverse_case(tr_type, log2n, 300, 31 + log2n)
        c["dst_off"], c["stride_dst"], c["dst_bytes"] = c["pred_off"], c["stride_pred"], c["pred"].size
        want = cases.run_oracle("inverse", c)
        s = 1 << log2n
        plane = B.u8(c["pred"].copy())
        B.inverse_transform_add(plane, c["stride_pred"], plane, c["stride_pred"], B.off32(c["pred_off"]), B.off32(c["pred_off"]), B.i16(c["coeffs"]), tr_type, log2n)
        torch.cuda.synchronize()
        cases.assert_same("inverse_in_place_%d_%d" % (tr_type, log2n), (cases._dst_rows(plane.cpu().numpy(), c["stride_pred"], c["pred_off"], s, s),), want)
        c2 = dict(c, scale=51, shift=14)
        want = cases.run_oracle("dequant_inverse", c2)
        plane = B.u8(c["pred"].copy())
        B.dequant_inverse_transform_add(plane, c["stride_pred"], plane, c["stride_pred"], B.off32(c["pred_off"]), B.off32(c["pred_off"]), B.i16(c["coeffs"]), 51, 14, tr_type, log2n)
        torch.cuda.synchronize()
        cases.assert_same("dequant_inverse_in_place_%d_%d" % (tr_type, log2n), (cases._dst_rows(plane.cpu().numpy(), c["stride_pred"], c["pred_off"], s, s),), want)
    for name, family, c in cases.quant_cases():
        if family != "recon":
            continue
        c = dict(c, rec_off=c["pred_off"], stride_rec=c["stride_pred"], rec_bytes=c["pred"].size)
        want = cases.run_oracle("recon", c)
        plane = B.u8(c["pred"].copy())
        B.quantize_reconstruct(plane, c["stride_pred"], plane, c["stride_pred"], B.off32(c["pred_off"]), B.off32(c["pred_off"]), B.i16(c["res"]), c["ncbs"])
        torch.cuda.synchronize()
        cases.assert_same(name + "_in_place", (cases._dst_rows(plane.cpu().numpy(), c["stride_pred"], c["pred_off"], c["ncbs"], c["ncbs"]),), want)


def test_selftest_hip_column():
    """hevcasm_test() with the HIP bit: the per-block entries behind the tables agree with the C
    path on the reference's own self-test inputs (reference grammar, no -MISMATCH)."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); from hevcasm_amd import capi; L = capi.lib(); e = C.c_int(0);\n"
            "m = capi.C_REF | capi.HIP\n"
            "for f in ('sad', 'sad_multiref', 'ssd', 'hadamard_satd', 'quantize_inverse', 'quantize', 'quantize_reconstruct', 'pred_uni', 'pred_bi', 'inverse_transform_add', 'transform', 'pred_intra', 'ssd_linear'):\n"
            "    getattr(L, 'hevcasm_test_' + f)(C.byref(e), m)\n"
            "print('ERRORS', e.value)\n") % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ERRORS 0" in r.stdout, r.stdout[-3000:]
    assert "MISMATCH" not in r.stdout
    assert r.stdout.count(" HIP:") == 23 + 23 + 5 + 3 + 4 + 4 + 4 + 24 * 8 + 22 * 4 + 5 + 5 + 7 + 1


def test_bottom_up_planes_every_strided_family():
    """The reference takes ptrdiff_t strides, so a codec may hand it bottom-up planes (negative strides).  The same PUs / TUs
    with every plane flipped top to bottom, every stride negated and every offset moved to the flipped position of the
    block's first row must give the same blocks -- from the kernels (32-bit offset arithmetic that wraps) and from the oracle."""
    import cases
    from hevcasm_amd import workload as wl

    def flip_plane(a, stride):
        rows = a.size // stride
        assert rows * stride == a.size
        return np.ascontiguousarray(a.reshape(rows, stride)[::-1]).reshape(-1), rows

    def flip_off(off, stride, rows):
        o = off.astype(np.int64)
        return ((rows - 1 - o // stride) * stride + o % stride).astype(np.uint32)

    for taps, w, h, bi in ((8, 16, 16, False), (8, 64, 64, True), (8, 8, 4, False), (4, 4, 8, True), (8, 12, 16, False), (4, 2, 4, False),
                           (8, 5, 3, False), (8, 32, 32, True), (4, 16, 16, False),
                           # the strip kernel (4 taps, W <= 8 or 16; 8 rows in flight where H % 8 == 0) and the 16-rows-in-flight luma loop
                           (4, 8, 8, True), (4, 6, 8, False), (4, 8, 16, True), (4, 3, 5, False), (4, 16, 4, True), (8, 16, 32, False), (8, 24, 32, False)):
        c = wl.pred_case(taps, w, h, 37, 4242 + w, bi=bi, frame=(max(2 * w, 64), max(3 * h, 64)))
        fam = "pred_bi" if bi else "pred_uni"
        f = dict(c)
        for r in range(2 if bi else 1):
            f["ref%d" % r], rows = flip_plane(c["ref%d" % r], c["stride_ref"])
            f["ref%d_off" % r] = flip_off(c["ref%d_off" % r], c["stride_ref"], rows)
        f["dst_off"] = flip_off(c["dst_off"], c["stride_dst"], c["dst_bytes"] // c["stride_dst"])
        f["stride_ref"], f["stride_dst"] = -c["stride_ref"], -c["stride_dst"]
        want = cases.run_oracle(fam, c)
        cases.assert_same("%s %d-tap %dx%d top-down" % (fam, taps, w, h), cases.run_hip(fam, c), want)
        cases.assert_same("%s %d-tap %dx%d bottom-up" % (fam, taps, w, h), cases.run_hip(fam, f), want)
    for tr, lg in wl.TRANSFORMS:
        size = 1 << lg
        c = wl.inverse_case(tr, lg, 29, 5151 + lg, "natural", frame=(4 * size, 3 * size))
        f = dict(c)
        f["pred"], prow = flip_plane(c["pred"], c["stride_pred"])
        f["pred_off"] = flip_off(c["pred_off"], c["stride_pred"], prow)
        f["dst_off"] = flip_off(c["dst_off"], c["stride_dst"], c["dst_bytes"] // c["stride_dst"])
        f["stride_pred"], f["stride_dst"] = -c["stride_pred"], -c["stride_dst"]
        want = cases.run_oracle("inverse", c)
        cases.assert_same("inverse %d/%d bottom-up" % (tr, lg), cases.run_hip("inverse", f), want)
        c["scale"], c["shift"] = 51, 6
        f["scale"], f["shift"] = 51, 6
        cases.assert_same("dequant_inverse %d/%d bottom-up" % (tr, lg), cases.run_hip("dequant_inverse", f), cases.run_oracle("dequant_inverse", c))
        c = wl.forward_case(tr, lg, 23, 970 + lg, frame=(4 * size, 3 * size))
        f = dict(c)
        f["src"], rows = flip_plane(c["src"], c["src_stride"])
        f["src_off"], f["src_stride"] = flip_off(c["src_off"], c["src_stride"], rows), -c["src_stride"]
        cases.assert_same("forward %d/%d bottom-up" % (tr, lg), cases.run_hip("forward", f), cases.run_oracle("forward", c))

    def flip_both(c):
        f = dict(c)
        for pl, st, of in (("src", "stride_src", "src_off"), ("ref", "stride_ref", "ref_off")):
            f[pl], rows = flip_plane(c[pl], c[st])
            f[of], f[st] = flip_off(c[of], c[st], rows), -c[st]
        return f

    for (w, h) in ((16, 16), (64, 64), (8, 8), (32, 24), (12, 16), (4, 4)):
        c = wl.sad_case(w, h, 41, 900 + w, ways=1)
        cases.assert_same("ssd %dx%d bottom-up" % (w, h), cases.run_hip("ssd", flip_both(c)), cases.run_oracle("ssd", c))
    for size in (2, 4, 8):
        c = wl.sad_case(max(size, 4), max(size, 4), 77, 950 + size) | dict(size=size)
        cases.assert_same("satd %d bottom-up" % size, cases.run_hip("satd", flip_both(c)), cases.run_oracle("satd", c))
    for (w, h, r) in ((64, 64, 8), (16, 16, 5), (32, 16, 16)):
        c = cases._me_case(w, h, r, 3, 990 + w)
        cases.assert_same("me %dx%d bottom-up" % (w, h), cases.run_hip("me", flip_both(c)), cases.run_oracle("me", c))
    for (w, h) in ((16, 16), (8, 8), (64, 32)):
        for metric in (0, 1):
            c = cases._subpel_case(w, h, 4, 995 + w, "noise") | dict(metric=metric)
            cases.assert_same("subpel %dx%d metric %d bottom-up" % (w, h, metric), cases.run_hip("subpel", flip_both(c)), cases.run_oracle("subpel", c))


def test_prediction_and_reconstruction_at_the_top_of_the_4gib_window():
    """As for SAD: prediction, inverse transform + add and the forward transform address with base + 32-bit offset.  PUs / TUs whose
    windows end just below 4 GiB - 64 above the plane's base (the last one ON the plane's last byte), reference and destination
    planes alike, positive and negative strides."""
    import torch
    from oracle import pyoracle
    from hevcasm_amd import batch as B
    size = (1 << 32) - 64
    tail = 1 << 20
    base = size - tail
    rng = np.random.default_rng(78)
    ref_t = rng.integers(0, 256, tail, dtype=np.uint8)
    ref = torch.zeros(size, dtype=torch.uint8, device="cuda")
    ref[base:] = torch.from_numpy(ref_t).cuda()
    dst = torch.zeros(size, dtype=torch.uint8, device="cuda")
    ref_o = np.concatenate([ref_t, np.zeros(1 << 16, dtype=np.uint8)])   # the oracle may walk the whole filter window (zero taps) past the plane's end
    stride = 2048
    rows_avail = tail // stride

    def blocks(plane_rows, w, h, n):
        # n non-overlapping w x h blocks in the tail, margins of 8 around them; the last one's window may be flush with the end
        r = 8 + (np.arange(n) // 8) * (h + 16)
        c = 8 + (np.arange(n) % 8) * 96 + rng.integers(0, 8, n)
        assert r.max() + h + 8 < plane_rows
        return (r * stride + c).astype(np.int64)

    for taps, w, h in ((8, 16, 16), (8, 64, 64), (4, 8, 8), (8, 8, 4)):
        n = 24
        ro_t = blocks(rows_avail, w, h, n)
        do_t = blocks(rows_avail, w, h, n)
        frac = ((rng.integers(0, taps // 2 if taps == 8 else 8, n)) | (rng.integers(0, 4 if taps == 8 else 8, n) << 4)).astype(np.uint8)
        after = taps // 2
        # the last PU: integer vector, its block ends on the reference plane's last byte; the one before: a 2-D fraction whose window does
        ro_t[-1], frac[-1] = tail - (h - 1) * stride - w, 0
        ro_t[-2], frac[-2] = tail - (h - 1 + after) * stride - w - after, 0x11
        do_t[-1] = tail - (h - 1) * stride - w
        for sgn in (1, -1):
            if sgn == 1:
                ro, do = ro_t, do_t
                want = pyoracle.pred_uni_batch(taps, tail, stride, ref_o, stride, do.astype(np.uint32), ro.astype(np.uint32), frac, w, h)
                want_blocks = np.stack([want.reshape(-1, stride)[d // stride:d // stride + h, d % stride:d % stride + w] for d in do])
            else:
                # bottom-up: the same blocks, offsets name their first row = the highest address
                ro, do = ro_t + (h - 1) * stride, do_t + (h - 1) * stride
                ro[-2:] = ro_t[-2:]
                ro, frac_n = ro[:-2], frac[:-2]
                do = do[:-2]
                want = pyoracle.pred_uni_batch(taps, tail, -stride, ref_o, -stride, do.astype(np.uint32), ro.astype(np.uint32), frac_n, w, h)
                want_blocks = np.stack([want.reshape(-1, stride)[d // stride - h + 1:d // stride + 1, d % stride:d % stride + w][::-1] for d in do])
            f = frac if sgn == 1 else frac[:-2]
            dst.zero_()
            B.pred_uni(dst, sgn * stride, ref, sgn * stride, B.off32((do + base).astype(np.uint32)), B.off32((ro + base).astype(np.uint32)), B.u8(f), taps, w, h)
            got = dst[base:].cpu().numpy().reshape(-1, stride)
            if sgn == 1:
                got_blocks = np.stack([got[d // stride:d // stride + h, d % stride:d % stride + w] for d in do])
            else:
                got_blocks = np.stack([got[d // stride - h + 1:d // stride + 1, d % stride:d % stride + w][::-1] for d in do])
            assert np.array_equal(got_blocks, want_blocks), (taps, w, h, sgn)
    # inverse transform + add, and the forward transform reading a residual plane, at the top of the window
    for lg in (2, 3, 4, 5):
        nblk, s = 16, 1 << lg
        po_t, do_t = blocks(rows_avail, s, s, nblk), blocks(rows_avail, s, s, nblk)
        po_t[-1] = do_t[-1] = tail - (s - 1) * stride - s
        coeffs = rng.integers(-300, 300, nblk * s * s).astype(np.int16)
        want = pyoracle.inverse_transform_add_batch(0, lg, tail, stride, ref_o, stride, do_t.astype(np.uint32), po_t.astype(np.uint32), coeffs)
        dst.zero_()
        B.inverse_transform_add(dst, stride, ref, stride, B.off32((do_t + base).astype(np.uint32)), B.off32((po_t + base).astype(np.uint32)), B.i16(coeffs), 0, lg)
        got = dst[base:].cpu().numpy().reshape(-1, stride)
        w2 = want.reshape(-1, stride)
        for d in do_t:
            assert np.array_equal(got[d // stride:d // stride + s, d % stride:d % stride + s], w2[d // stride:d // stride + s, d % stride:d % stride + s]), (lg, d)
    del ref, dst
    torch.cuda.empty_cache()


def test_descriptor_arrays_at_odd_word_alignment():
    """Round 4 fetches a full wave's (workgroup's) descriptors with SCALAR loads of 8 / 16 / 32 consecutive words (sad.hip: block_cost_sums;
    transform.hip).  The API promises nothing beyond uint32 alignment for src_off / ref_off / pred_off / dst_off: arrays that start 1, 2 and 3
    words into an allocation (views into a larger buffer) must give the same answers -- SAD one reference and four, SSD, inverse 16x16 / 32x32
    (+ fused), forward 16x16 -- with batch sizes that leave a partial last wave (the per-lane path) behind the full ones."""
    import torch
    from hevcasm_amd import batch as B, workload as wl
    from oracle import pyoracle

    def shifted(a, words):
        t = torch.zeros(a.size + 8, dtype=torch.int32, device="cuda")
        t[words:words + a.size] = torch.from_numpy(a.astype(np.int64).astype(np.uint32).view(np.int32)).cuda()
        return t[words:words + a.size]

    for words in (1, 2, 3):
        for (w, h, ways, n) in ((16, 16, 1, 1000 + words), (16, 16, 4, 777), (8, 8, 1, 515), (32, 32, 4, 130), (24, 32, 4, 99), (64, 64, 1, 41)):
            c = wl.sad_case(w, h, n, 5100 + w + words, ways=ways)
            src, ref = B.u8(c["src"]), B.u8(c["ref"])
            so, ro = shifted(c["src_off"], words), shifted(c["ref_off"], (words + 1) % 4)
            if ways == 1:
                got = B.sad(src, c["stride_src"], ref, c["stride_ref"], so, ro, w, h).cpu().numpy()
                want = pyoracle.sad_batch(c["src"], c["stride_src"], c["ref"], c["stride_ref"], c["src_off"], c["ref_off"], w, h)
                ssd = B.ssd(src, c["stride_src"], ref, c["stride_ref"], so, ro, w, h).cpu().numpy()
                assert np.array_equal(ssd, pyoracle.ssd_batch(c["src"], c["stride_src"], c["ref"], c["stride_ref"], c["src_off"], c["ref_off"], w, h)), (words, w, h, "ssd")
            else:
                got = B.sad_multiref(src, c["stride_src"], ref, c["stride_ref"], so, ro, w, h).cpu().numpy()
                want = pyoracle.sad_multiref_batch(c["src"], c["stride_src"], c["ref"], c["stride_ref"], c["src_off"], c["ref_off"], w, h)
            assert np.array_equal(got, want), (words, w, h, ways)
        for lg in (4, 5):
            n = 100 + 7 * words
            c = wl.inverse_case(0, lg, n, 5200 + lg + words, "full")
            pred, coeffs = B.u8(c["pred"]), B.i16(c["coeffs"])
            dof, pof = shifted(c["dst_off"], words), shifted(c["pred_off"], 4 - words)
            dst = torch.zeros(c["dst_bytes"], dtype=torch.uint8, device="cuda")
            B.inverse_transform_add(dst, c["stride_dst"], pred, c["stride_pred"], dof, pof, coeffs, 0, lg)
            want = pyoracle.inverse_transform_add_batch(0, lg, c["dst_bytes"], c["stride_dst"], c["pred"], c["stride_pred"], c["dst_off"], c["pred_off"], c["coeffs"])
            assert np.array_equal(dst.cpu().numpy(), want), (words, lg, "inverse")
            dst.zero_()
            B.dequant_inverse_transform_add(dst, c["stride_dst"], pred, c["stride_pred"], dof, pof, coeffs, 51, 14, 0, lg)
            want = pyoracle.dequant_inverse_transform_add_batch(0, lg, c["dst_bytes"], c["stride_dst"], c["pred"], c["stride_pred"], c["dst_off"], c["pred_off"], c["coeffs"], 51, 14)
            assert np.array_equal(dst.cpu().numpy(), want), (words, lg, "fused")
        f = wl.forward_case(0, 4, 90 + words, 5300 + words)
        got = B.transform(B.i16(f["src"]), f["src_stride"], shifted(f["src_off"], words), 0, 4).cpu().numpy()
        assert np.array_equal(got, pyoracle.transform_batch(0, 4, f["src"], f["src_stride"], f["src_off"])), (words, "forward")


def test_pred_strip_segments_keep_every_row():
    """Round 4 cuts small launches of tall 4-tap PUs into row segments (pred_strip4_kernel).  A height that the segment count does not divide must not
    lose its last rows: found by the extreme-content fuzz group at 16x17 (the first form of the launcher's test accepted H = 17 as 2 x 8 rows).
    Every height 8 .. 33 at the strip kernel's widths, uni and bi, few PUs (= a small launch: segments wherever the launcher allows them)."""
    from hevcasm_amd import workload as wl
    for w in (4, 8, 16):
        for h in list(range(8, 34)):
            if h > 32:
                continue
            for bi in (False, True):
                c = wl.pred_case(4, w, h, 9, 8800 + 40 * w + h, bi=bi, frame=(2 * w, 2 * h))
                fam = "pred_bi" if bi else "pred_uni"
                cases.assert_same("strip segments %s %dx%d" % (fam, w, h), cases.run_hip(fam, c), cases.run_oracle(fam, c))


def test_pred_bi_hfirst_kernel_sizes_and_edges():
    """Round 4: luma bi-prediction of 16- and 32-wide PUs and of 64x64 runs pred_hfirst_bi8_kernel (8-wide ones stay on the stream kernel: an 8-wide form was measured and lost;
    its cases stay here) (a lane per strip and reference, horizontal pass first).  Every
    height it takes (standard and not), batches that leave spare lanes and a partial last wave, bottom-up planes, the rails (all-0 / all-255 planes: the
    14-bit intermediates wrap) and every fraction pair including zero fractions on either reference."""
    from hevcasm_amd import workload as wl
    for w in (8, 16, 32):
        for h in (16, 24, 32, 40, 48, 64):
            for n in (1, 37):
                c = wl.pred_case(8, w, h, n, 9100 + w + h + n, bi=True, frame=(4 * w, 2 * h))
                cases.assert_same("hfirst bi %dx%d n=%d" % (w, h, n), cases.run_hip("pred_bi", c), cases.run_oracle("pred_bi", c))
    for n in (1, 9):                 # 64x64: 16 lanes to a PU, 4 PUs to a wave
        c = wl.pred_case(8, 64, 64, n, 9250 + n, bi=True, frame=(192, 128))
        cases.assert_same("hfirst bi 64x64 n=%d" % n, cases.run_hip("pred_bi", c), cases.run_oracle("pred_bi", c))
    for kind in ("zeros", "max"):
        c = wl.pred_case(8, 16, 16, 50, 9300, bi=True, frame=(64, 32), kind=kind)
        cases.assert_same("hfirst bi rails " + kind, cases.run_hip("pred_bi", c), cases.run_oracle("pred_bi", c))
    c = cases._wrap14(wl.pred_case(8, 32, 32, 7, 9350, bi=True, frame=(64, 64), fracs=(2, 2)))
    cases.assert_same("hfirst bi wrap 32x32", cases.run_hip("pred_bi", c), cases.run_oracle("pred_bi", c))
    # plane strides that are not a multiple of 4 (the byte-aligned form of the kernel) and plane bases at every byte alignment (the word-aligned form's shifts)
    for w in (8, 16):
        c = wl.pred_case(8, w, 32, 23, 9360 + w, bi=True, frame=(66, 64))
        assert c["stride_ref"] % 4 == 2
        cases.assert_same("hfirst bi odd stride %d" % w, cases.run_hip("pred_bi", c), cases.run_oracle("pred_bi", c))
    for fx in (0, 1):            # 8-wide PUs, filtered and unfiltered columns
        for fy in (0, 2):
            c = wl.pred_case(8, 8, 16, 45, 9380 + 2 * fx + fy, bi=True, frame=(32, 32), fracs=(fx, fy))
            cases.assert_same("hfirst bi 8-wide fractions %d %d" % (fx, fy), cases.run_hip("pred_bi", c), cases.run_oracle("pred_bi", c))
    import torch
    from hevcasm_amd import batch as B
    for k in (1, 2, 3):
        c = wl.pred_case(8, (8, 16, 32)[k - 1], 16, 29, 9370 + k, bi=True, frame=(128, 32))
        shifted = [torch.from_numpy(np.concatenate([np.zeros(s, np.uint8), c["ref%d" % r]])).cuda()[s:] for r, s in ((0, k), (1, (k + 1) % 4))]
        d = B.pred_bi(torch.zeros(c["dst_bytes"], dtype=torch.uint8, device="cuda"), c["stride_dst"], shifted[0], shifted[1], c["stride_ref"], B.off32(c["dst_off"]), B.off32(c["ref0_off"]),
                      B.off32(c["ref1_off"]), B.u8(c["frac0"]), B.u8(c["frac1"]), 8, c["w"], c["h"])
        got = cases._only_blocks(d.cpu().numpy(), c["stride_dst"], c["dst_off"], c["w"], c["h"])
        cases.assert_same("hfirst bi plane base + %d" % k, got, cases.run_oracle("pred_bi", c))
    for fx0 in range(4):
        for fy0 in range(4):
            c = wl.pred_case(8, 32, 24, 21, 9400 + 4 * fx0 + fy0, bi=True, frame=(96, 48), fracs=(fx0, fy0))
            c["frac1"] = ((3 - fx0) | ((fy0 ^ 1) << 4)) * np.ones_like(c["frac1"])
            cases.assert_same("hfirst bi fractions %d %d" % (fx0, fy0), cases.run_hip("pred_bi", c), cases.run_oracle("pred_bi", c))



@pytest.mark.gpu
@pytest.mark.parametrize("tr,lg", [(1, 2), (0, 2), (0, 3), (0, 4), (0, 5)])
def test_config4_fused_pipeline_at_its_own_size(tr, lg):
    """BASELINE.json configs[3] at SURVEY 8(d)'s size -- 2^24 coefficients per transform size -- through hevcasm_hip_dequant_inverse_transform_add_batch
    (full workgroups with scalar offsets, then the partial tail: launch shapes the 24..130-block oracle cases never reach).  Composition to match:
    hevcasm_quantize_inverse (reference quantize.c:53-62) followed by hevcasm_inverse_transform_add (residual_decode.c:371-413).
      whole output == quantize_inverse -> inverse_transform_add as two HIP launches (each oracle-tested on its own);
      first and last 200 blocks == the oracle's composition."""
    import torch
    from hevcasm_amd import batch as B, workload as wl
    from oracle import pyoracle
    size = 1 << lg
    n = (1 << 24) >> (2 * lg)
    n += 3                       # not a multiple of any workgroup's block count: the tail launch runs too
    for kind, (scale, shift) in (("natural", (51, 14)), ("full", (51, 14)), ("natural", (57 << 8, lg + 3)), ("full", (40 << 3, max(1, lg + 3)))):
        c = wl.inverse_case(tr, lg, n, 4400 + lg + shift, kind, frame=(wl.FRAME_W, wl.FRAME_H))
        pred, levels = B.u8(c["pred"]), B.i16(c["coeffs"])
        doff, poff = B.off32(c["dst_off"]), B.off32(c["pred_off"])
        fused = torch.zeros(c["dst_bytes"], dtype=torch.uint8, device="cuda")
        B.dequant_inverse_transform_add(fused, c["stride_dst"], pred, c["stride_pred"], doff, poff, levels, scale, shift, tr, lg)
        two = torch.zeros(c["dst_bytes"], dtype=torch.uint8, device="cuda")
        B.inverse_transform_add(two, c["stride_dst"], pred, c["stride_pred"], doff, poff, B.quantize_inverse(levels, scale, shift), tr, lg)
        torch.cuda.synchronize()
        assert torch.equal(fused, two), (tr, lg, kind, scale, shift, int((fused != two).sum()))
        assert int(fused.count_nonzero()) > fused.numel() // 4          # something was written
        got = fused.cpu().numpy()
        for sl in (slice(0, 200), slice(n - 200, n)):
            want = pyoracle.dequant_inverse_transform_add_batch(tr, lg, c["dst_bytes"], c["stride_dst"], c["pred"], c["stride_pred"], c["dst_off"][sl], c["pred_off"][sl],
                                                                c["coeffs"][sl.start * size * size: sl.stop * size * size], scale, shift)
            cases.assert_same("config4 %s %d/%d blocks %d.." % (kind, scale, shift, sl.start),
                              (cases._dst_rows(got, c["stride_dst"], c["dst_off"][sl], size, size),), (cases._dst_rows(want, c["stride_dst"], c["dst_off"][sl], size, size),))
        del fused, two, pred, levels
        torch.cuda.empty_cache()


@pytest.mark.gpu
@pytest.mark.parametrize("lg", [2, 3, 4, 5])
def test_config4_reconstruct_at_its_own_size(lg):
    """hevcasm_hip_quantize_reconstruct_batch (reference quantize.c:292-302) over 2^24 residuals: the whole output against the definition
    clip8(pred + res) computed by torch on the same device arrays (gather / scatter by the same offsets), first and last 200 blocks against the oracle."""
    import torch
    from hevcasm_amd import batch as B, workload as wl
    from oracle import pyoracle
    size = 1 << lg
    n = ((1 << 24) >> (2 * lg)) + 5
    for kind in ("residual9", "full"):
        c = wl.inverse_case(0, lg, n, 4500 + lg, kind, frame=(wl.FRAME_W, wl.FRAME_H))
        pred, res = B.u8(c["pred"]), B.i16(c["coeffs"])
        rec = torch.zeros(c["dst_bytes"], dtype=torch.uint8, device="cuda")
        B.quantize_reconstruct(rec, c["stride_dst"], pred, c["stride_pred"], B.off32(c["dst_off"]), B.off32(c["pred_off"]), res, size)
        torch.cuda.synchronize()
        rows = torch.arange(size, device="cuda", dtype=torch.int64)
        pidx = (torch.from_numpy(c["pred_off"].astype(np.int64)).cuda()[:, None, None] + rows[None, :, None] * c["stride_pred"] + rows[None, None, :])
        didx = (torch.from_numpy(c["dst_off"].astype(np.int64)).cuda()[:, None, None] + rows[None, :, None] * c["stride_dst"] + rows[None, None, :])
        want = (pred[pidx].to(torch.int32) + res.view(n, size, size).to(torch.int32)).clamp_(0, 255).to(torch.uint8)
        assert torch.equal(rec[didx], want), (lg, kind)
        del pidx, didx, want
        got = rec.cpu().numpy()
        for sl in (slice(0, 200), slice(n - 200, n)):
            o = pyoracle.quantize_reconstruct_batch(c["dst_bytes"], c["stride_dst"], c["pred"], c["stride_pred"], c["dst_off"][sl], c["pred_off"][sl],
                                                    c["coeffs"][sl.start * size * size: sl.stop * size * size], size)
            cases.assert_same("recon %d %s" % (size, kind), (cases._dst_rows(got, c["stride_dst"], c["dst_off"][sl], size, size),),
                              (cases._dst_rows(o, c["stride_dst"], c["dst_off"][sl], size, size),))
        del rec, pred, res
        torch.cuda.empty_cache()


@pytest.mark.gpu
def test_frame_sse_full_size():
    """hevcasm_hip_frame_sse_batch on 64 frames of 3840 x 2160 inside padded planes: equals the per-row hevcasm_hip_ssd_linear_batch sums (each row exact
    in 32 bits) added up in 64 bits by torch, equals the oracle on the first and last frame, and PSNR of a known offset comes out as the formula says."""
    import torch
    from hevcasm_amd import batch as B, workload as wl
    from oracle import pyoracle
    fw, fh, pad, frames = 3840, 2160, 64, 24
    stride = fw + 2 * pad
    span = stride * (fh + 2)
    a = wl.fill_u8(span * frames + 64, 31, device="cuda")
    b = wl.fill_u8(span * frames + 64, 32, device="cuda")
    a_off = (np.arange(frames, dtype=np.int64) * span + pad + 1).astype(np.uint32)
    b_off = (np.arange(frames, dtype=np.int64) * span + stride + 3).astype(np.uint32)
    got = B.frame_sse(a, stride, b, stride, B.off32(a_off), B.off32(b_off), fw, fh)
    ro_a = (a_off.astype(np.int64)[:, None] + np.arange(fh)[None, :] * stride).reshape(-1).astype(np.uint32)
    ro_b = (b_off.astype(np.int64)[:, None] + np.arange(fh)[None, :] * stride).reshape(-1).astype(np.uint32)
    rows = B.ssd_linear(a, b, B.off32(ro_a), B.off32(ro_b), fw)
    torch.cuda.synchronize()
    assert torch.equal(got, rows.to(torch.int64).view(frames, fh).sum(dim=1))
    host_a, host_b = a.cpu().numpy(), b.cpu().numpy()
    for f in (0, frames - 1):
        want = pyoracle.frame_sse_batch(host_a, stride, host_b, stride, a_off[f:f + 1], b_off[f:f + 1], fw, fh)
        assert int(got[f]) == int(want[0]) > 2 ** 32
    # a plane against itself + 3 (no wrap: values capped at 252): SSE = 9 W H, PSNR = 10 log10(255^2 / 9)
    c = a.clamp(max=252)
    d = c + 3
    s = B.frame_sse(c, stride, d, stride, B.off32(a_off), B.off32(a_off), fw, fh)
    torch.cuda.synchronize()
    assert bool((s == 9 * fw * fh).all())


@pytest.mark.gpu
@pytest.mark.parametrize("size", [2, 4, 8])
def test_satd_full_size_properties(size):
    """Hadamard SATD over 2^26 pixels per launch (the bench's size; the oracle cases stop at a few hundred blocks): SATD(x, x) = 0; symmetric; bounded below
    by the DC term -- |sum of the block's differences| is one of the N^2 coefficients the sum of absolutes runs over, so SATD >= (|sum d| + N / 4) / (N / 2)
    (hadamard.c:98, :110) -- and above by N^2 times the largest coefficient; first and last 4096 blocks against the oracle."""
    import torch
    from hevcasm_amd import batch as B, workload as wl
    from oracle import pyoracle
    n = (1 << 26) // (size * size)
    side = max(size, 4)
    c = wl.sad_case(side, side, n, 5150 + size, frame=(wl.FRAME_W, wl.FRAME_H), device="cuda")
    src, ref, so, ro = c["src"], c["ref"], B.off32(c["src_off"]), B.off32(c["ref_off"])
    log2 = size.bit_length() - 1
    zero = B.hadamard_satd(src, c["stride_src"], src, c["stride_src"], so, so, log2)
    assert int(zero.abs().max()) == 0
    ab = B.hadamard_satd(src, c["stride_src"], ref, c["stride_ref"], so, ro, log2)
    ba = B.hadamard_satd(ref, c["stride_ref"], src, c["stride_src"], ro, so, log2)
    assert torch.equal(ab, ba)
    rows = torch.arange(size, device="cuda", dtype=torch.int64)
    ia = torch.from_numpy(c["src_off"].astype(np.int64)).cuda()[:, None, None] + rows[None, :, None] * c["stride_src"] + rows[None, None, :]
    ib = torch.from_numpy(c["ref_off"].astype(np.int64)).cuda()[:, None, None] + rows[None, :, None] * c["stride_ref"] + rows[None, None, :]
    d = src[ia].to(torch.int32) - ref[ib].to(torch.int32)
    dc = d.sum(dim=(1, 2)).abs()
    half = max(size // 2, 1)
    assert bool((ab >= (dc + size // 4) // half).all()) and bool((ab <= (size * size * d.abs().sum(dim=(1, 2)) + size // 4) // half).all())
    src_h, ref_h, got = src.cpu().numpy(), ref.cpu().numpy(), ab.cpu().numpy()
    for sl in (slice(0, 4096), slice(n - 4096, n)):
        want = pyoracle.hadamard_satd_batch(size, src_h, c["stride_src"], ref_h, c["stride_ref"], c["src_off"][sl], c["ref_off"][sl])
        assert np.array_equal(got[sl], want), (size, sl)


@pytest.mark.gpu
def test_ssd_linear_full_size_properties():
    """hevcasm_hip_ssd_linear_batch over 2 x 256 MiB cut into runs of every kernel variant's length: zero on identical runs, symmetric, the runs of a
    plane add up to the plane's frame SSE (64-bit), first and last runs against the oracle."""
    import torch
    from hevcasm_amd import batch as B, workload as wl
    from oracle import pyoracle
    total = 1 << 28
    a, b = wl.fill_u8(total, 61, device="cuda"), wl.fill_u8(total, 62, device="cuda")
    zero32 = B.off32(np.zeros(1, dtype=np.uint32))
    whole = int(B.frame_sse(a, 1 << 14, b, 1 << 14, zero32, zero32, 1 << 14, 1 << 14)[0])        # the same bytes as 16384 rows of 16384
    a_h, b_h = a[: 1 << 20].cpu().numpy(), b[: 1 << 20].cpu().numpy()
    for size in (48, 0x200, 1920, 32768):
        n = total // size
        off = B.off32((np.arange(n, dtype=np.int64) * size).astype(np.uint32))
        ab = B.ssd_linear(a, b, off, off, size)
        assert torch.equal(ab, B.ssd_linear(b, a, off, off, size)) and int(B.ssd_linear(a, a, off, off, size).abs().max()) == 0
        if n * size == total:
            assert int(ab.to(torch.int64).sum()) == whole, size
        k = min(n, (1 << 20) // size)
        want = pyoracle.ssd_linear_batch(a_h, b_h, (np.arange(k, dtype=np.int64) * size).astype(np.uint32), (np.arange(k, dtype=np.int64) * size).astype(np.uint32), size)
        assert np.array_equal(ab[:k].cpu().numpy(), want), size
