"""ONE launch for the block costs of a frame's partition (round 6): hevcasm_hip_sad_runs / _sad_multiref_runs / _ssd_runs on a
quad-tree partition over every PU size of pred_inter.c:438-444, against the oracle's one-size batches per run."""
import numpy as np
import pytest

from hevcasm_amd import capi, workload as wl

KINDS = ("sad", "sad4", "ssd")


def _oracle(oracle_lib, c, kind, runs=None):
    n = len(c["src_off"])
    want = np.zeros((n, 4) if kind == "sad4" else (n,), dtype=np.int32)
    for (w, h, first, count) in c["sizes"]:
        s = slice(first, first + count)
        if kind == "sad":
            want[s] = oracle_lib.sad_batch(c["src"], c["stride_src"], c["ref"], c["stride_ref"], c["src_off"][s], c["ref_off"][s], w, h)
        elif kind == "ssd":
            want[s] = oracle_lib.ssd_batch(c["src"], c["stride_src"], c["ref"], c["stride_ref"], c["src_off"][s], c["ref_off"][s], w, h)
        else:
            want[s] = oracle_lib.sad_multiref_batch(c["src"], c["stride_src"], c["ref"], c["stride_ref"], c["src_off"][s], c["ref_off4"][4 * first:4 * (first + count)], w, h).reshape(-1, 4)
    return want


def _hip(c, kind, runs=None):
    import torch
    from hevcasm_amd import batch as B
    out = B.cost_runs(kind, B.u8(c["src"]), c["stride_src"], B.u8(c["ref"]), c["stride_ref"], B.off32(c["src_off"]), B.off32(c["ref_off4"] if kind == "sad4" else c["ref_off"]),
                      c["runs"] if runs is None else runs)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_cost_case_holds_every_size_once_per_run():
    c = wl.cost_frame_case(1920, 1080, 5)
    assert {(w, h) for (w, h, _, _) in c["sizes"]} == set(wl.PU_SIZES)
    assert sum(w * h * n for (w, h, _, n) in c["sizes"]) == 1920 * 1080
    assert len(c["runs"]) == 24 and int(c["runs"]["count"].sum()) == c["n_pus"]
    lo = c["ref_off4"].astype(np.int64).reshape(-1, 4)
    assert lo.min() >= 0 and lo.max() + 63 * c["stride_ref"] + 64 <= len(c["ref"])


def test_runs_entry_points_refuse_without_a_gpu():
    L = capi.lib()
    if not L.hevcasm_hip_available():
        for f in (L.hevcasm_hip_sad_runs, L.hevcasm_hip_sad_multiref_runs, L.hevcasm_hip_ssd_runs):
            assert f(None, 0, None, 0, None, None, None, None, 1, None) == capi.ENODEV


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
def test_hip_one_launch_for_a_1080p_partition_equals_the_oracle(kind, oracle_lib):
    c = wl.cost_frame_case(1920, 1080, 31)
    got, want = _hip(c, kind), _oracle(oracle_lib, c, kind)
    assert np.array_equal(got, want), "%s: %d of %d differ" % (kind, int((got != want).sum()), want.size)


@pytest.mark.gpu
def test_hip_cost_runs_small_frames_many_runs_and_bad_arguments(oracle_lib):
    import torch
    from hevcasm_amd import batch as B
    for (w, h, kind_of_fill) in ((200, 136, "video"), (64, 64, "max"), (128, 72, "noise")):
        c = wl.cost_frame_case(w, h, 40 + w, kind=kind_of_fill)
        for kind in KINDS:
            assert np.array_equal(_hip(c, kind), _oracle(oracle_lib, c, kind)), (w, h, kind)
    # more runs than one launch's table holds, with empty ones in between; the one-size batch calls give the same numbers
    c = wl.cost_frame_case(256, 192, 77)
    pieces = []
    for (w, h, first, count) in c["sizes"]:
        for s in range(0, count, 2):
            pieces.append((first + s, min(2, count - s), (w << 8) | h))
            pieces.append((0, 0, (16 << 8) | 16))
    pieces = np.array(pieces, dtype=wl.COST_RUN_DTYPE)
    assert int((pieces["count"] > 0).sum()) > 96
    for kind in KINDS:
        assert np.array_equal(_hip(c, kind, runs=pieces), _oracle(oracle_lib, c, kind)), kind
    (w, h, first, count) = c["sizes"][3]
    one = B.sad(B.u8(c["src"]), c["stride_src"], B.u8(c["ref"]), c["stride_ref"], B.off32(c["src_off"][first:first + count]), B.off32(c["ref_off"][first:first + count]), w, h)
    assert np.array_equal(one.cpu().numpy(), _hip(c, "sad")[first:first + count])
    # argument checks
    L = capi.lib()
    t = torch.zeros(65536, dtype=torch.uint8, device="cuda")
    o = torch.zeros(64, dtype=torch.int32, device="cuda")
    def f(*rows, fn=L.hevcasm_hip_sad_runs):
        r = np.array(list(rows), dtype=wl.COST_RUN_DTYPE)
        return fn(t.data_ptr(), 256, t.data_ptr(), 256, o.data_ptr(), o.data_ptr(), o.data_ptr(), r.ctypes.data, len(r), None)
    assert f((0, 1, (16 << 8) | 16)) == 0
    assert f((0, 1, (20 << 8) | 16)) == capi.EINVAL           # not one of HEVC's widths
    assert f((0, 1, (16 << 8) | 6)) == capi.EINVAL            # height not a multiple of 4
    assert f((0, 1, (16 << 8) | 68)) == capi.EINVAL
    assert f((0xffffffff, 2, (16 << 8) | 16)) == capi.EINVAL
    assert L.hevcasm_hip_sad_runs(t.data_ptr(), 256, t.data_ptr(), 256, o.data_ptr(), o.data_ptr(), o.data_ptr(), None, 1, None) == capi.EINVAL
    assert L.hevcasm_hip_sad_runs(t.data_ptr(), 256, t.data_ptr(), 256, o.data_ptr(), o.data_ptr(), o.data_ptr(), None, 0, None) == 0
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_hip_search_and_refinement_of_a_partition_in_one_launch_each(oracle_lib):
    """hevcasm_hip_me_full_search_runs / hevcasm_hip_subpel_refine_runs (both metrics): a frame's PUs of every size, one launch, per PU the one-size entry
    point's answer == the oracle's; > 96 runs; bad arguments."""
    import torch
    from hevcasm_amd import batch as B
    c = wl.cost_frame_case(320, 192, 51, border=32, kind="video")
    src, ref, so, ro = B.u8(c["src"]), B.u8(c["ref"]), B.off32(c["src_off"]), B.off32(c["ref_off"])
    for rng in (6, 13):
        got = B.me_full_search_runs(src, c["stride_src"], ref, c["stride_ref"], so, ro, c["runs"], rng).cpu().numpy()
        want = np.concatenate([oracle_lib.me_full_search(c["src"], c["stride_src"], c["ref"], c["stride_ref"], c["src_off"][f:f + n], c["ref_off"][f:f + n], w, h, rng) for (w, h, f, n) in c["sizes"]])
        assert np.array_equal(got, want), "search +-%d: %d PUs differ" % (rng, int((got != want).any(axis=1).sum()))
    for metric in (0, 1):
        got = B.subpel_refine_runs(src, c["stride_src"], ref, c["stride_ref"], so, ro, c["runs"], metric).cpu().numpy()
        want = np.concatenate([oracle_lib.subpel_refine(c["src"], c["stride_src"], c["ref"], c["stride_ref"], c["src_off"][f:f + n], c["ref_off"][f:f + n], w, h, metric) for (w, h, f, n) in c["sizes"]])
        assert np.array_equal(got, want), "refinement metric %d: %d PUs differ" % (metric, int((got != want).any(axis=1).sum()))
    # more runs than one table holds
    pieces = []
    for (w, h, first, count) in c["sizes"]:
        for s in range(0, count, 3):
            pieces.append((first + s, min(3, count - s), (w << 8) | h))
    pieces = np.array(pieces, dtype=wl.COST_RUN_DTYPE)
    assert len(pieces) > 96
    a = B.me_full_search_runs(src, c["stride_src"], ref, c["stride_ref"], so, ro, pieces, 6).cpu().numpy()
    b = B.me_full_search_runs(src, c["stride_src"], ref, c["stride_ref"], so, ro, c["runs"], 6).cpu().numpy()
    assert np.array_equal(a, b)
    a = B.subpel_refine_runs(src, c["stride_src"], ref, c["stride_ref"], so, ro, pieces).cpu().numpy()
    b = B.subpel_refine_runs(src, c["stride_src"], ref, c["stride_ref"], so, ro, c["runs"]).cpu().numpy()
    assert np.array_equal(a, b)
    L = capi.lib()
    t = torch.zeros(1 << 18, dtype=torch.uint8, device="cuda")
    o = torch.zeros(64, dtype=torch.int32, device="cuda")
    one = np.array([(0, 1, (16 << 8) | 16)], dtype=wl.COST_RUN_DTYPE)
    bad = np.array([(0, 1, (16 << 8) | 18)], dtype=wl.COST_RUN_DTYPE)
    me = lambda r, rng: L.hevcasm_hip_me_full_search_runs(t.data_ptr(), 512, t.data_ptr(), 512, o.data_ptr(), o.data_ptr(), o.data_ptr(), r.ctypes.data, len(r), rng, None)
    sp = lambda r, m: L.hevcasm_hip_subpel_refine_runs(t.data_ptr(), 512, t.data_ptr(), 512, o.data_ptr(), o.data_ptr(), o.data_ptr(), r.ctypes.data, len(r), m, None)
    assert me(bad, 4) == capi.EINVAL and me(one, 128) == capi.EINVAL and me(one, -1) == capi.EINVAL
    assert sp(bad, 0) == capi.EINVAL and sp(one, 2) == capi.EINVAL
    torch.cuda.synchronize()
