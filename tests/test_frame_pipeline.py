"""The whole hot path as ONE chain on one frame (hevcasm_amd/pipeline.py): every stage's output feeds the next, on the GPU through the batch C-ABI and
on the CPU through the oracle, and every intermediate must be identical -- a composition no single kernel test exercises (a wrong vector, a prediction
off by one row, a coefficient order or a cbf convention would all surface downstream)."""
import numpy as np
import pytest

from hevcasm_amd import me_frame, pipeline
from oracle import pyoracle


class OracleOps:
    """The same stages through the oracle (numpy arrays throughout)."""

    def plan(self, frame, size):
        cur_off, ref_base = pipeline.block_grid(frame["w"], frame["h"], size, frame["stride_cur"], frame["stride_ref"], frame["pad"])
        return frame["cur"], frame["ref"], cur_off, ref_base

    def me_full_search(self, cur, sc, ref, sr, cur_off, ref_off, w, h, search_range):
        return pyoracle.me_full_search(cur, sc, ref, sr, cur_off, ref_off, w, h, search_range)

    def offsets_after_search(self, ref_base, mv, sr):
        return (ref_base.astype(np.int64) + mv[:, 1].astype(np.int64) * sr + mv[:, 0]).astype(np.uint32)

    def subpel_refine(self, cur, sc, ref, sr, cur_off, ref_off, w, h):
        return pyoracle.subpel_refine(cur, sc, ref, sr, cur_off, ref_off, w, h)

    def subpel_descriptors(self, ref_int, qmv, sr):
        ref_sub = (ref_int.astype(np.int64) + (qmv[:, 1] >> 2).astype(np.int64) * sr + (qmv[:, 0] >> 2)).astype(np.uint32)
        return ref_sub, ((qmv[:, 0] & 3) | ((qmv[:, 1] & 3) << 4)).astype(np.uint8)

    def pred_uni(self, dst_bytes, sd, ref, sr, dst_off, ref_off, frac, taps, w, h):
        return pyoracle.pred_uni_batch(taps, dst_bytes, sd, ref, sr, dst_off, ref_off, frac, w, h)

    def residual(self, cur, pred):
        return cur.astype(np.int16) - pred.astype(np.int16)

    def transform(self, residual, stride, off, tr_type, log2n):
        return pyoracle.transform_batch(tr_type, log2n, residual, stride, off)

    def quantize(self, coeffs, scale, shift, offset, n_per_block):
        return pyoracle.quantize_batch(coeffs, scale, shift, offset, n_per_block)

    def dequant_inverse_transform_add(self, dst_bytes, sd, pred, sp, dst_off, pred_off, levels, scale, shift, tr_type, log2n):
        return pyoracle.dequant_inverse_transform_add_batch(tr_type, log2n, dst_bytes, sd, pred, sp, dst_off, pred_off, levels, scale, shift)

    def ssd(self, a, sa, b, sb, a_off, b_off, w, h):
        return pyoracle.ssd_batch(a, sa, b, sb, a_off, b_off, w, h)

    def frame_sse(self, a, sa, b, sb, w, h):
        z = np.zeros(1, dtype=np.uint32)
        return pyoracle.frame_sse_batch(a, sa, b, sb, z, z, w, h)


    # the mixed-partition chain (pipeline.stages_partition): every run of a size through the oracle's one-size batches
    def plan_partition(self, frame, pus, tu):
        cur_off, ref_base, sizes, order = pipeline.partition_grid(pus, frame["stride_cur"], frame["stride_ref"], frame["pad"])
        tu_off, _ = pipeline.block_grid(frame["w"], frame["h"], tu, frame["stride_cur"], frame["stride_ref"], frame["pad"])
        return frame["cur"], frame["ref"], cur_off, ref_base, sizes, order, tu_off

    def me_full_search_runs(self, cur, sc, ref, sr, cur_off, ref_off, sizes, search_range):
        return np.concatenate([pyoracle.me_full_search(cur, sc, ref, sr, cur_off[f:f + n], ref_off[f:f + n], w, h, search_range) for (w, h, f, n) in sizes])

    def subpel_refine_runs(self, cur, sc, ref, sr, cur_off, ref_off, sizes):
        return np.concatenate([pyoracle.subpel_refine(cur, sc, ref, sr, cur_off[f:f + n], ref_off[f:f + n], w, h) for (w, h, f, n) in sizes])

    def pred_uni_runs(self, dst_bytes, sd, ref, sr, dst_off, ref_off, frac, sizes):
        dst = np.zeros(dst_bytes, dtype=np.uint8)
        for (w, h, f, n) in sizes:
            dst |= pyoracle.pred_uni_batch(8, dst_bytes, sd, ref, sr, dst_off[f:f + n], ref_off[f:f + n], frac[f:f + n], w, h)      # PUs tile the plane
        return dst

    def cost_runs(self, kind, a, sa, b, sb, a_off, b_off, sizes):
        fn = pyoracle.sad_batch if kind == "sad" else pyoracle.ssd_batch
        return np.concatenate([fn(a, sa, b, sb, a_off[f:f + n], b_off[f:f + n], w, h) for (w, h, f, n) in sizes])


    # the transform stages on a residual quad-tree (stages_partition(tus=...)): every run through the oracle's one-size batches with the run's parameters
    def plan_tus(self, frame, tus, qp):
        off, sizes, _ = pipeline.tu_grid(tus, frame["stride_cur"], qp)
        return off, sizes

    def transform_runs(self, residual, stride, off, tu_sizes, n_coeffs):
        return np.concatenate([pyoracle.transform_batch(tr, l2, residual, stride, off[f:f + n]) for (l2, tr, f, n, cf, q) in tu_sizes])

    def quantize_runs(self, coeffs, tu_sizes, n_tus):
        parts = [pyoracle.quantize_batch(coeffs[cf:cf + (n << (2 * l2))], q[0], q[1], q[2], 1 << (2 * l2)) for (l2, tr, f, n, cf, q) in tu_sizes]
        return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])

    def dequant_inverse_transform_add_runs(self, dst_bytes, sd, pred, sp, off, tu_sizes, levels):
        dst = np.zeros(dst_bytes, dtype=np.uint8)
        for (l2, tr, f, n, cf, q) in tu_sizes:
            dst |= pyoracle.dequant_inverse_transform_add_batch(tr, l2, dst_bytes, sd, pred, sp, off[f:f + n], off[f:f + n], levels[cf:cf + (n << (2 * l2))], q[3], q[4])      # TUs tile the plane
        return dst


    def transform_quantize_reconstruct_runs(self, n_bytes, stride, cur, pred, off, tu_sizes, n_tus):
        res = self.residual(cur, pred)
        levels, cbf = self.quantize_runs(self.transform_runs(res, stride, off, tu_sizes, n_bytes), tu_sizes, n_tus)
        return self.dequant_inverse_transform_add_runs(n_bytes, stride, pred, stride, off, tu_sizes, levels), levels, cbf


def _with_half_pel_motion(frame):
    """The generator moves blocks by whole pixels; here the lower half of the current frame becomes the mean of itself and its right-hand neighbour,
    i.e. content half a pixel away from anything in the reference, so that the refinement has fractional winners."""
    w, h = frame["w"], frame["h"]
    cur = frame["cur"].reshape(h, w).astype(np.uint16)
    low = cur[h // 2:]
    low[:, :-1] = (low[:, :-1] + low[:, 1:] + 1) >> 1
    frame["cur"] = np.ascontiguousarray(cur.astype(np.uint8)).reshape(-1)
    return frame


def test_pipeline_on_the_oracle_is_a_sane_encode(oracle_lib):
    """CPU: the chain hangs together -- motion is found, some vectors are fractional, reconstruction is close to the source."""
    frame = _with_half_pel_motion(me_frame.synthetic_frame_pair(192, 128, 16, seed=4, ctu=16, max_motion=6))
    r = pipeline.stages(frame, OracleOps(), size=16, search_range=8)
    assert (r["qmv"][:, :2] & 3).any()
    assert r["mv"].shape == (96, 3) and (np.abs(r["mv"][:, :2]) <= 8).all() and (np.abs(r["qmv"][:, :2]) <= 3).all()
    assert (r["mv"][:, :2] != 0).any() and pipeline.psnr(int(r["frame_sse"][0]), 192, 128) > 30.0
    assert int(r["frame_sse"][0]) == int(r["block_ssd"].astype(np.int64).sum())          # the frame is tiled exactly by its blocks
    assert np.array_equal(r["cbf"] != 0, (r["levels"].reshape(96, -1) != 0).any(axis=1))


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,size,rng", [(960, 544, 16, 8), (512, 256, 32, 12), (384, 256, 8, 6)])
def test_pipeline_hip_equals_oracle_stage_by_stage(w, h, size, rng):
    frame = _with_half_pel_motion(me_frame.synthetic_frame_pair(w, h, rng + 8, seed=w + size, ctu=size, max_motion=rng - 2))
    want = pipeline.stages(frame, OracleOps(), size=size, search_range=rng)
    got = pipeline.HipOps.to_numpy(pipeline.stages(frame, pipeline.HipOps(), size=size, search_range=rng))
    _same_stages(got, want)
    assert (got["qmv"][:, :2] & 3).any(), "the case must exercise fractional prediction"


def _same_stages(got, want):
    for key in ("mv", "qmv", "pred", "residual", "coeffs", "levels", "cbf", "recon", "block_ssd", "frame_sse"):
        g, wv = np.asarray(got[key]), np.asarray(want[key])
        if key == "cbf":                                     # only zero / non-zero is meaningful (reference quantize.c:257)
            g, wv = g != 0, wv != 0
        assert g.shape == wv.shape and np.array_equal(g, wv), "stage %s differs (%d of %d)" % (key, int((g != wv).sum()), g.size)


@pytest.mark.gpu
def test_pipeline_as_one_hip_graph():
    """The chain has no host step between its launches, so it captures into ONE HIP graph; replayed (twice), its outputs are the eager run's and the
    oracle's, and after the frame's content is changed IN PLACE on the device a replay follows it."""
    import torch
    frame = _with_half_pel_motion(me_frame.synthetic_frame_pair(640, 352, 16, seed=21, ctu=16, max_motion=6))
    want = pipeline.stages(frame, OracleOps(), size=16, search_range=8)
    ops = pipeline.HipOps()
    graph, results = ops.capture(frame, size=16, search_range=8)
    for _ in range(2):
        graph.replay()
    _same_stages(pipeline.HipOps.to_numpy(results), want)
    # new content in the resident planes (the next frame of a sequence): same graph, new answers
    frame2 = _with_half_pel_motion(me_frame.synthetic_frame_pair(640, 352, 16, seed=22, ctu=16, max_motion=6))
    cur_dev, ref_dev, _, _ = ops.plan(frame, 16)
    cur_dev.copy_(torch.from_numpy(frame2["cur"]))
    ref_dev.copy_(torch.from_numpy(frame2["ref"]))
    graph.replay()
    _same_stages(pipeline.HipOps.to_numpy(results), pipeline.stages(frame2, OracleOps(), size=16, search_range=8))


PARTITION_KEYS = ("mv", "qmv", "pred", "pu_sad", "residual", "coeffs", "levels", "cbf", "recon", "pu_ssd", "frame_sse")


def _same_partition_stages(got, want):
    for key in PARTITION_KEYS:
        g, wv = np.asarray(got[key]), np.asarray(want[key])
        if key == "cbf":
            g, wv = g != 0, wv != 0
        assert g.shape == wv.shape and np.array_equal(g, wv), "stage %s differs (%d of %d)" % (key, int((g != wv).sum()), g.size)


def test_partition_chain_on_the_oracle_hangs_together(oracle_lib):
    """CPU: the chain on a quad-tree partition -- the PUs' prediction cost IS the refinement's winning cost (two independent routes to one number: the fused
    49-candidate kernel's minimum, and prediction written to memory + SAD), and the PUs' SSDs add up to the frame's."""
    from hevcasm_amd import workload as wl
    frame = _with_half_pel_motion(me_frame.synthetic_frame_pair(192, 128, 16, seed=5, ctu=16, max_motion=6))
    pus = wl.inter_partition(192, 128, 3)
    r = pipeline.stages_partition(frame, OracleOps(), pus, tu=16, search_range=8)
    assert len(r["mv"]) == len(pus) and len({(p[2], p[3]) for p in pus}) >= 12
    assert np.array_equal(r["pu_sad"], r["qmv"][:, 2])
    assert int(r["pu_ssd"].astype(np.int64).sum()) == int(r["frame_sse"][0])
    assert (r["qmv"][:, :2] & 3).any() and pipeline.psnr(int(r["frame_sse"][0]), 192, 128) > 30.0


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,tu,rng", [(960, 576, 16, 8), (512, 256, 32, 6), (384, 256, 8, 6)])
def test_partition_chain_hip_equals_oracle_stage_by_stage(w, h, tu, rng):
    """VERDICT round 5, item 2, last clause: the pipeline's chain on a MIXED partition -- prediction, the PUs' SAD and the PUs' SSD each as ONE launch over every
    PU size present -- equals the oracle stage by stage."""
    from hevcasm_amd import workload as wl
    frame = _with_half_pel_motion(me_frame.synthetic_frame_pair(w, h, rng + 8, seed=w + tu, ctu=16, max_motion=rng - 2))
    pus = wl.inter_partition(w, h, w + 1)
    want = pipeline.stages_partition(frame, OracleOps(), pus, tu=tu, search_range=rng)
    got = pipeline.HipOps.to_numpy(pipeline.stages_partition(frame, pipeline.HipOps(), pus, tu=tu, search_range=rng))
    _same_partition_stages(got, want)
    assert np.array_equal(got["pu_sad"], got["qmv"][:, 2])


def test_partition_chain_with_a_tu_quad_tree_on_the_oracle(oracle_lib):
    """CPU: the chain with the transform stages on a residual quad-tree (4x4 DST / DCT ... 32x32, each size with its own quantiser parameters): the levels of a
    run are what the one-size chain gives for that run's TUs, the cbf flags say which TUs have levels, and the reconstruction is a sane encode."""
    from hevcasm_amd import workload as wl
    frame = _with_half_pel_motion(me_frame.synthetic_frame_pair(192, 128, 16, seed=5, ctu=16, max_motion=6))
    pus, tus = wl.inter_partition(192, 128, 3), wl.tu_partition(192, 128, 4)
    r = pipeline.stages_partition(frame, OracleOps(), pus, search_range=8, tus=tus, qp=22)
    off, sizes, order = pipeline.tu_grid(tus, 192, 22)
    assert len(sizes) == 5 and r["coeffs"].size == 192 * 128 == r["levels"].size and r["cbf"].size == len(tus)
    for (l2, tr, f, n, cf, q) in sizes:
        assert np.array_equal(r["cbf"][f:f + n] != 0, (r["levels"][cf:cf + (n << (2 * l2))].reshape(n, -1) != 0).any(axis=1))
    assert int(r["pu_ssd"].astype(np.int64).sum()) == int(r["frame_sse"][0]) and pipeline.psnr(int(r["frame_sse"][0]), 192, 128) > 32.0
    assert r["cbf"].any() and not r["cbf"].all()
    f = pipeline.stages_partition(frame, OracleOps(), pus, search_range=8, tus=tus, qp=22, fused=True)
    assert np.array_equal(f["recon"], r["recon"]) and np.array_equal(f["levels"], r["levels"]) and f["coeffs"] is None


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,rng,qp", [(960, 576, 8, 27), (384, 256, 6, 12)])
def test_partition_chain_with_a_tu_quad_tree_hip_equals_oracle(w, h, rng, qp):
    """The last per-size stages of the inter chain as ONE launch each: PUs of every size AND transform units of every size, stage by stage == the oracle."""
    from hevcasm_amd import workload as wl
    frame = _with_half_pel_motion(me_frame.synthetic_frame_pair(w, h, rng + 8, seed=w + 3, ctu=16, max_motion=rng - 2))
    pus, tus = wl.inter_partition(w, h, w + 1), wl.tu_partition(w, h, w + 2)
    want = pipeline.stages_partition(frame, OracleOps(), pus, search_range=rng, tus=tus, qp=qp)
    ops = pipeline.HipOps()
    got = pipeline.HipOps.to_numpy(pipeline.stages_partition(frame, ops, pus, search_range=rng, tus=tus, qp=qp))
    _same_partition_stages(got, want)
    graph, results = ops.capture(frame, pus=pus, search_range=rng, tus=tus, qp=qp)          # and as one HIP graph
    graph.replay()
    _same_partition_stages(pipeline.HipOps.to_numpy(results), want)
    # the residual loop as ONE launch from the current and predicted planes: the same levels, flags, reconstruction and costs
    fused = pipeline.HipOps.to_numpy(pipeline.stages_partition(frame, ops, pus, search_range=rng, tus=tus, qp=qp, fused=True))
    assert fused["residual"] is None and fused["coeffs"] is None
    for key in ("levels", "cbf", "recon", "pu_ssd", "frame_sse"):
        assert np.array_equal(np.asarray(fused[key]), np.asarray(got[key])), "fused residual loop: %s differs" % key


@pytest.mark.gpu
def test_partition_chain_as_one_hip_graph():
    from hevcasm_amd import workload as wl
    frame = _with_half_pel_motion(me_frame.synthetic_frame_pair(640, 384, 16, seed=31, ctu=16, max_motion=6))
    pus = wl.inter_partition(640, 384, 8)
    want = pipeline.stages_partition(frame, OracleOps(), pus, tu=16, search_range=8)
    ops = pipeline.HipOps()
    graph, results = ops.capture(frame, pus=pus, tu=16, search_range=8)
    for _ in range(2):
        graph.replay()
    _same_partition_stages(pipeline.HipOps.to_numpy(results), want)
