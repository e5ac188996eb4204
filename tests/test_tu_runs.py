"""ONE launch per stage for a frame's transform units (round 6): hevcasm_hip_transform_runs / _quantize_runs / _inverse_transform_add_runs /
_dequant_inverse_transform_add_runs on a residual quad-tree over the four sizes of the reference's tables (residual_decode.h:54, :82: 4x4 DST and DCT, 8x8,
16x16, 32x32), each run with the (de)quantiser parameters of its size, against the oracle's one-size batches per run (residual_decode.c:69-413, :592-892,
quantize.c:53-62, :160-186: pinned, oracle == refdrv on every size)."""
import numpy as np
import pytest

from hevcasm_amd import capi, workload as wl

STAGES = ("forward", "quantize", "inverse", "dequant_inverse")


def _coeffs(c, seed, kind="full"):
    return wl.fill_i16(c["n_coeffs"], seed, kind)


def _oracle(oracle_lib, c, stage, coeffs=None):
    """The stage per run through the oracle's one-size batch calls."""
    if stage == "forward":
        out = np.zeros(c["n_coeffs"], dtype=np.int16)
        for (log2, tr, first, count, cf) in c["sizes"]:
            out[cf:cf + (count << (2 * log2))] = oracle_lib.transform_batch(tr, log2, c["residual"], c["stride"], c["off"][first:first + count])
        return (out,)
    if stage == "quantize":
        out, cbf = np.zeros(c["n_coeffs"], dtype=np.int16), np.zeros(c["n_tus"], dtype=np.int32)
        for (log2, tr, first, count, cf), r in zip(c["sizes"], c["runs_q"]):
            e = cf + (count << (2 * log2))
            out[cf:e], cbf[first:first + count] = oracle_lib.quantize_batch(coeffs[cf:e], int(r["scale"]), int(r["shift"]), int(r["offset"]), 1 << (2 * log2))
        return out, cbf
    dst = np.zeros(c["stride"] * c["h"], dtype=np.uint8)
    for (log2, tr, first, count, cf), r in zip(c["sizes"], c["runs_dq"]):
        o, e = c["off"][first:first + count], cf + (count << (2 * log2))
        if stage == "inverse":
            part = oracle_lib.inverse_transform_add_batch(tr, log2, dst.size, c["stride"], c["pred"], c["stride"], o, o, coeffs[cf:e])
        else:
            part = oracle_lib.dequant_inverse_transform_add_batch(tr, log2, dst.size, c["stride"], c["pred"], c["stride"], o, o, coeffs[cf:e], int(r["scale"]), int(r["shift"]))
        dst |= part          # the runs' TUs are disjoint and each call starts from zeros
    return (dst,)


def _hip(c, stage, coeffs=None, runs=None, in_place=False):
    import torch
    from hevcasm_amd import batch as B
    off = B.off32(c["off"])
    if stage == "forward":
        out = (B.transform_runs(torch.from_numpy(c["residual"]).cuda(), c["stride"], off, c["runs_q"] if runs is None else runs, c["n_coeffs"],
                                out=torch.zeros(c["n_coeffs"], dtype=torch.int16, device="cuda")),)
    elif stage == "quantize":
        out = B.quantize_runs(torch.from_numpy(coeffs).cuda(), c["runs_q"] if runs is None else runs, c["n_tus"],
                              out=torch.zeros(c["n_coeffs"], dtype=torch.int16, device="cuda"), cbf=torch.zeros(c["n_tus"], dtype=torch.int32, device="cuda"))
    else:
        pred = B.u8(c["pred"])
        dst = pred if in_place else torch.zeros(c["stride"] * c["h"], dtype=torch.uint8, device="cuda")
        out = (B.inverse_transform_add_runs(dst, c["stride"], pred, c["stride"], off, off, torch.from_numpy(coeffs).cuda(), c["runs_dq"] if runs is None else runs,
                                            dequant=stage == "dequant_inverse"),)
    torch.cuda.synchronize()
    return tuple(t.cpu().numpy() for t in out)


def _mask(c):
    """Bytes of the destination plane the TUs cover (everything but the row padding)."""
    m = np.zeros((c["h"], c["stride"]), dtype=bool)
    m[:, :c["w"]] = True
    return m.reshape(-1)


def test_tu_case_tiles_the_plane_with_every_size():
    c = wl.tu_frame_case(1920, 1080, 5)
    assert {(s[0], s[1]) for s in c["sizes"]} == {(2, 1), (2, 0), (3, 0), (4, 0), (5, 0)}
    assert sum(n << (2 * log2) for (log2, _, _, n, _) in c["sizes"]) == 1920 * 1080 == c["n_coeffs"]
    cover = np.zeros((1080, 1920), dtype=np.int32)
    for (x, y, log2, _) in c["tus"]:
        cover[y:y + (1 << log2), x:x + (1 << log2)] += 1
    assert (cover == 1).all()
    assert wl.TU_RUN_DTYPE.itemsize == 28 and c["runs_q"]["count"].sum() == c["n_tus"]
    # the parameters are inside the reference's asserts (quantize.c:162-168) for every size at every QP
    for qp in range(52):
        q = wl.intra_qp(qp)
        assert (q[:, 0] < 0x8000).all() and (q[:, 1] >= 16).all() and (q[:, 1] <= 27).all() and (q[:, 3] < 0x8000).all() and (q[:, 4] >= 1).all()


def test_tu_run_struct_is_28_bytes_in_c(tmp_path):
    import os
    import subprocess
    src = tmp_path / "s.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "hevcasm_hip.h"\nint main(void){printf("%zu %zu %zu %zu\\n", sizeof(hevcasm_hip_tu_run), '
                   'offsetof(hevcasm_hip_tu_run, coeff_first), offsetof(hevcasm_hip_tu_run, log2TrafoSize), offsetof(hevcasm_hip_tu_run, scale));return 0;}\n')
    exe = tmp_path / "s"
    subprocess.run(["gcc", "-I", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include"), str(src), "-o", str(exe)], check=True)
    assert subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split() == ["28", "8", "12", "16"]
    assert [wl.TU_RUN_DTYPE.fields[k][1] for k in ("coeff_first", "log2", "scale")] == [8, 12, 16]


def test_tu_runs_entry_points_refuse_without_a_gpu():
    L = capi.lib()
    if not L.hevcasm_hip_available():
        assert L.hevcasm_hip_transform_runs(None, None, 0, None, None, 1, None) == capi.ENODEV
        assert L.hevcasm_hip_quantize_runs(None, None, None, None, 1, None) == capi.ENODEV
        for f in (L.hevcasm_hip_inverse_transform_add_runs, L.hevcasm_hip_dequant_inverse_transform_add_runs):
            assert f(None, 0, None, 0, None, None, None, None, 1, None) == capi.ENODEV
        assert L.hevcasm_hip_transform_quantize_reconstruct_runs(None, 0, None, 0, None, 0, None, None, None, None, None, None, None, 1, None) == capi.ENODEV
        assert L.hevcasm_hip_quantize_reconstruct_runs(None, 0, None, 0, None, None, None, None, 1, None) == capi.ENODEV
        assert L.hevcasm_hip_transform_quantize_reconstruct_planes(None, 1, None, None, None, None, None, None, None, 1, None) == capi.ENODEV


@pytest.mark.gpu
@pytest.mark.parametrize("stage", STAGES)
def test_hip_one_launch_for_a_1080p_frames_tus_equals_the_oracle(stage, oracle_lib):
    c = wl.tu_frame_case(1920, 1080, 41, pad=64)
    coeffs = None if stage == "forward" else _coeffs(c, 7, "natural" if stage == "dequant_inverse" else "full")
    got, want = _hip(c, stage, coeffs), _oracle(oracle_lib, c, stage, coeffs)
    keep = _mask(c) if stage in ("inverse", "dequant_inverse") else slice(None)
    for g, w in zip(got, want):
        assert np.array_equal(g[keep] if g.size == c["stride"] * c["h"] else g, w[keep] if w.size == c["stride"] * c["h"] else w), \
            "%s: %d of %d differ" % (stage, int((g != w).sum()), w.size)
    if stage in ("inverse", "dequant_inverse"):
        assert not got[0][~_mask(c)].any(), "bytes outside the TUs were written"


@pytest.mark.gpu
def test_hip_tu_chain_small_frames_extremes_in_place(oracle_lib):
    """forward -> quantise -> dequantise + inverse + add, every stage one launch, on ragged small planes and extreme content; the last stage also in place."""
    for (w, h, qp, residual, kind) in ((200, 136, 27, "residual9", "video"), (64, 64, 0, "full", "max"), (132, 76, 51, "min", "zeros"), (96, 160, 22, "max", "noise")):
        c = wl.tu_frame_case(w, h, 60 + w, qp=qp, kind=kind, residual=residual)
        (coeffs,), (want_c,) = _hip(c, "forward"), _oracle(oracle_lib, c, "forward")
        assert np.array_equal(coeffs, want_c), (w, h, "forward")
        (levels, cbf), (want_l, want_cbf) = _hip(c, "quantize", coeffs), _oracle(oracle_lib, c, "quantize", coeffs)
        assert np.array_equal(levels, want_l) and np.array_equal(cbf != 0, want_cbf != 0), (w, h, "quantize")
        for stage in ("inverse", "dequant_inverse"):
            (want,) = _oracle(oracle_lib, c, stage, levels)
            assert np.array_equal(_hip(c, stage, levels)[0], want), (w, h, stage)
            assert np.array_equal(_hip(c, stage, levels, in_place=True)[0], want), (w, h, stage, "in place")


@pytest.mark.gpu
def test_hip_quantize_reconstruct_runs_equals_the_oracle(oracle_lib):
    """hevcasm_quantize_reconstruct (quantize.h:99) for a plane's units of every size in one launch: per run the oracle's one-size batch; full-range residuals
    (both clips), in place, a bad run refused."""
    import torch
    from hevcasm_amd import batch as B
    for (w, h, kind, pad) in ((1920, 1080, "video", 64), (200, 136, "noise", 0), (64, 64, "max", 4)):
        c = wl.tu_frame_case(w, h, 90 + w, kind=kind, pad=pad)
        res = wl.fill_i16(c["n_coeffs"], 5 + w, "full" if kind != "video" else "residual9")
        want = np.zeros(c["stride"] * c["h"], dtype=np.uint8)
        for (log2, tr, first, count, cf) in c["sizes"]:
            o = c["off"][first:first + count]
            want |= oracle_lib.quantize_reconstruct_batch(want.size, c["stride"], c["pred"], c["stride"], o, o, res[cf:cf + (count << (2 * log2))], 1 << log2)
        off, m = B.off32(c["off"]), _mask(c)
        for in_place in (False, True):
            pred = B.u8(c["pred"])
            rec = pred if in_place else torch.zeros(want.size, dtype=torch.uint8, device="cuda")
            got = B.quantize_reconstruct_runs(rec, c["stride"], pred, c["stride"], off, off, torch.from_numpy(res).cuda(), c["runs_q"]).cpu().numpy()
            assert np.array_equal(got[m], want[m]), (w, h, in_place)
            assert np.array_equal(got[~m], c["pred"][~m] if in_place else np.zeros_like(got[~m]))
    bad = c["runs_q"][:1].copy(); bad["log2"] = 6
    z = torch.zeros(1 << 14, dtype=torch.int16, device="cuda")
    L = capi.lib()
    assert L.hevcasm_hip_quantize_reconstruct_runs(z.data_ptr(), 64, z.data_ptr(), 64, off.data_ptr(), off.data_ptr(), z.data_ptr(), bad.ctypes.data, 1, None) == capi.EINVAL
    assert L.hevcasm_hip_quantize_reconstruct_runs(z.data_ptr(), 64, z.data_ptr(), 64, off.data_ptr(), off.data_ptr(), z.data_ptr(), bad.ctypes.data, 0, None) == 0


def _fused_oracle(oracle_lib, c):
    """src - pred through the oracle's one-size batches, run by run: (levels, cbf, reconstruction)."""
    c2 = dict(c, residual=(c["src"].astype(np.int16) - c["pred"].astype(np.int16)))
    (coeffs,) = _oracle(oracle_lib, c2, "forward")
    levels, cbf = _oracle(oracle_lib, c2, "quantize", coeffs)
    (rec,) = _oracle(oracle_lib, c2, "dequant_inverse", levels)
    return levels, cbf, rec


def _fused_hip(c, in_place=False, runs_q=None, runs_dq=None):
    import torch
    from hevcasm_amd import batch as B
    off, pred = B.off32(c["off"]), B.u8(c["pred"])
    rec = pred if in_place else torch.zeros(c["stride"] * c["h"], dtype=torch.uint8, device="cuda")
    out = B.transform_quantize_reconstruct_runs(rec, c["stride"], B.u8(c["src"]), c["stride"], pred, c["stride"], off, off, off, c["runs_q"] if runs_q is None else runs_q,
                                                c["runs_dq"] if runs_dq is None else runs_dq, c["n_coeffs"], c["n_tus"],
                                                levels=torch.zeros(c["n_coeffs"], dtype=torch.int16, device="cuda"), cbf=torch.zeros(c["n_tus"], dtype=torch.int32, device="cuda"))
    torch.cuda.synchronize()
    rec, levels, cbf = (t.cpu().numpy() for t in out)
    return levels, cbf, rec


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,qp,kind,src_kind,pad", [(1920, 1080, 27, "video", None, 64), (200, 136, 12, "noise", None, 0), (64, 64, 0, "zeros", "max", 0),
                                                      (132, 76, 51, "max", "zeros", 8), (96, 160, 30, "video", "noise", 0), (256, 192, 4, "video", None, 0)])
def test_hip_fused_residual_loop_equals_the_oracles_four_functions(w, h, qp, kind, src_kind, pad, oracle_lib):
    """hevcasm_hip_transform_quantize_reconstruct_runs == transform -> quantise -> dequantise -> inverse + add of src - pred through the oracle, every TU size,
    each with its parameters: the levels, the flags, the reconstruction; nothing outside the TUs written; also in place (rec == pred)."""
    c = wl.tu_frame_case(w, h, 80 + w, qp=qp, kind=kind, src_kind=src_kind, pad=pad)
    want_levels, want_cbf, want_rec = _fused_oracle(oracle_lib, c)
    for in_place in (False, True):
        levels, cbf, rec = _fused_hip(c, in_place=in_place)
        assert np.array_equal(levels, want_levels), "levels: %d of %d differ" % (int((levels != want_levels).sum()), levels.size)
        assert np.array_equal(cbf, want_cbf), "cbf"
        m = _mask(c)
        assert np.array_equal(rec[m], want_rec[m]), "reconstruction: %d of %d differ" % (int((rec[m] != want_rec[m]).sum()), int(m.sum()))
        assert np.array_equal(rec[~m], c["pred"][~m] if in_place else np.zeros_like(rec[~m])), "bytes outside the TUs were written"
    assert want_cbf.any() or qp > 40


@pytest.mark.gpu
def test_hip_fused_large_call_two_launches_and_many_runs(oracle_lib):
    """A call large enough to be split by register class (sizes below 32x32 in one launch, 32x32 on the matrix cores in another: > 8 workgroups per CU), and a call
    of more runs than one table holds with empty runs in between: the same levels, flags and reconstruction as the oracle."""
    c = wl.tu_frame_case(3840, 2176, 17, qp=30, pad=0)
    wgs = sum(-(-n // {2: 256, 3: 32, 4: 16, 5: 8}[log2]) for (log2, _, _, n, _) in c["sizes"])
    assert wgs > 256 * 8 and {s[0] for s in c["sizes"]} == {2, 3, 4, 5}
    want = _fused_oracle(oracle_lib, c)
    for g, w, what in zip(_fused_hip(c), want, ("levels", "cbf", "reconstruction")):
        assert np.array_equal(g, w), "split call: %s (%d differ)" % (what, int((g != w).sum()))
    c = wl.tu_frame_case(256, 192, 77, qp=18)
    def pieces(runs):
        out = []
        for r in runs:
            n2 = 1 << (2 * int(r["log2"]))
            for s in range(0, int(r["count"]), 3):
                out.append((int(r["first"]) + s, min(3, int(r["count"]) - s), int(r["coeff_first"]) + s * n2, r["log2"], r["tr_type"], 0, r["scale"], r["shift"], r["offset"]))
                out.append((7, 0, 64, r["log2"], r["tr_type"], 0, r["scale"], r["shift"], r["offset"]))
        return np.array(out, dtype=wl.TU_RUN_DTYPE)
    pq, pdq = pieces(c["runs_q"]), pieces(c["runs_dq"])
    assert int((pq["count"] > 0).sum()) > 64
    for g, w, what in zip(_fused_hip(c, runs_q=pq, runs_dq=pdq), _fused_oracle(oracle_lib, c), ("levels", "cbf", "reconstruction")):
        assert np.array_equal(g, w), "many runs: " + what


@pytest.mark.gpu
def test_hip_fused_residual_loop_over_the_three_planes_of_a_frame_in_one_launch(oracle_lib):
    """hevcasm_hip_transform_quantize_reconstruct_planes: luma, Cb and Cr of a 4:2:0 frame -- each with its own partition, stride and QP -- in ONE launch: per plane
    the oracle's composition (== the single-plane call); a plane index beyond the call's planes is refused."""
    import torch
    from hevcasm_amd import batch as B
    cs = [wl.tu_frame_case(416, 240, 5, qp=27, pad=32), wl.tu_frame_case(208, 120, 6, qp=30, pad=0, kind="noise"), wl.tu_frame_case(208, 120, 7, qp=33, pad=8, src_kind="max")]
    off, rq, rdq, first, coeff = [], [], [], 0, 0
    for plane, c in enumerate(cs):
        off.append(c["off"])
        for src_runs, out in ((c["runs_q"], rq), (c["runs_dq"], rdq)):
            r = src_runs.copy()
            r["first"] += first; r["coeff_first"] += coeff; r["plane"] = plane
            out.append(r)
        first += c["n_tus"]; coeff += c["n_coeffs"]
    # interleave the planes' runs: the order of the runs must not matter
    order = np.argsort(np.concatenate([np.arange(len(r)) * 3 + i for i, r in enumerate(rq)]), kind="stable")
    rq, rdq = np.concatenate(rq)[order], np.concatenate(rdq)[order]
    o = B.off32(np.concatenate(off))
    planes = [(torch.zeros(c["stride"] * c["h"], dtype=torch.uint8, device="cuda"), c["stride"], B.u8(c["src"]), c["stride"], B.u8(c["pred"]), c["stride"]) for c in cs]
    levels, cbf = torch.zeros(coeff, dtype=torch.int16, device="cuda"), torch.zeros(first, dtype=torch.int32, device="cuda")
    B.transform_quantize_reconstruct_planes(planes, o, o, o, rq, rdq, levels, cbf)
    torch.cuda.synchronize()
    levels, cbf, first, coeff = levels.cpu().numpy(), cbf.cpu().numpy(), 0, 0
    for plane, c in enumerate(cs):
        want_levels, want_cbf, want_rec = _fused_oracle(oracle_lib, c)
        m = _mask(c)
        assert np.array_equal(levels[coeff:coeff + c["n_coeffs"]], want_levels) and np.array_equal(cbf[first:first + c["n_tus"]], want_cbf), "plane %d levels / flags" % plane
        got = planes[plane][0].cpu().numpy()
        assert np.array_equal(got[m], want_rec[m]) and not got[~m].any(), "plane %d reconstruction" % plane
        first += c["n_tus"]; coeff += c["n_coeffs"]
    bad = rq.copy(); bad["plane"][0] = 3
    bad_dq = rdq.copy(); bad_dq["plane"][0] = 3
    with pytest.raises(Exception):
        B.transform_quantize_reconstruct_planes(planes, o, o, o, bad, bad_dq, torch.zeros(coeff, dtype=torch.int16, device="cuda"), torch.zeros(first, dtype=torch.int32, device="cuda"))


@pytest.mark.gpu
def test_hip_fused_residual_loop_equals_the_three_launches_and_refuses_bad_runs():
    import torch
    from hevcasm_amd import batch as B
    c = wl.tu_frame_case(640, 360, 13, qp=20, pad=16)
    levels, cbf, rec = _fused_hip(c)
    res = torch.from_numpy(c["src"].astype(np.int16) - c["pred"].astype(np.int16)).cuda()
    off, pred = B.off32(c["off"]), B.u8(c["pred"])
    coeffs = B.transform_runs(res, c["stride"], off, c["runs_q"], c["n_coeffs"])
    l3, f3 = B.quantize_runs(coeffs, c["runs_q"], c["n_tus"])
    r3 = B.inverse_transform_add_runs(torch.zeros(c["stride"] * c["h"], dtype=torch.uint8, device="cuda"), c["stride"], pred, c["stride"], off, off, l3, c["runs_dq"], dequant=True)
    torch.cuda.synchronize()
    assert np.array_equal(levels, l3.cpu().numpy()) and np.array_equal(cbf, f3.cpu().numpy()) and np.array_equal(rec, r3.cpu().numpy())
    # the two run arrays must describe the same units; both parameter sets are checked
    L = capi.lib()
    def call(rq, rdq, n=None):
        z = torch.zeros(1 << 16, dtype=torch.int16, device="cuda")
        return L.hevcasm_hip_transform_quantize_reconstruct_runs(z.data_ptr(), 64, z.data_ptr(), 64, z.data_ptr(), 64, off.data_ptr(), off.data_ptr(), off.data_ptr(), z.data_ptr(), z.data_ptr(),
                                                                 rq.ctypes.data, rdq.ctypes.data, len(rq) if n is None else n, None)
    rq, rdq = c["runs_q"][:1].copy(), c["runs_dq"][:1].copy()
    rq["count"] = rdq["count"] = 1
    assert call(rq, rdq) == 0 and call(rq, rdq, 0) == 0
    for field, value in (("count", 2), ("first", 1), ("coeff_first", 16), ("log2", 3), ("tr_type", 0), ("plane", 1)):
        bad = rdq.copy(); bad[field] = value
        assert call(rq, bad) == capi.EINVAL, field
    for field, value in (("shift", 15), ("shift", 28), ("scale", 0x8000), ("offset", -1)):
        bad = rq.copy(); bad[field] = value
        assert call(bad, rdq) == capi.EINVAL, field
    for field, value in (("shift", 0), ("shift", 32), ("scale", 40000)):
        bad = rdq.copy(); bad[field] = value
        assert call(rq, bad) == capi.EINVAL, field
    assert call(rq, rdq, -1) == capi.EINVAL
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_hip_tu_runs_many_runs_one_size_calls_and_bad_arguments(oracle_lib):
    import torch
    from hevcasm_amd import batch as B
    c = wl.tu_frame_case(256, 192, 77)
    # more runs than one launch's table holds (pieces of 3 TUs), with empty ones in between
    def pieces(runs):
        out = []
        for r in runs:
            n2 = 1 << (2 * int(r["log2"]))
            for s in range(0, int(r["count"]), 3):
                out.append((int(r["first"]) + s, min(3, int(r["count"]) - s), int(r["coeff_first"]) + s * n2, r["log2"], r["tr_type"], 0, r["scale"], r["shift"], r["offset"]))
                out.append((0, 0, 0, 3, 0, 0, r["scale"], r["shift"], r["offset"]))
        return np.array(out, dtype=wl.TU_RUN_DTYPE)
    pq, pdq = pieces(c["runs_q"]), pieces(c["runs_dq"])
    assert int((pq["count"] > 0).sum()) > 64
    coeffs = _coeffs(c, 9)
    assert np.array_equal(_hip(c, "forward", runs=pq)[0], _oracle(oracle_lib, c, "forward")[0])
    for g, w in zip(_hip(c, "quantize", coeffs, runs=pq), _oracle(oracle_lib, c, "quantize", coeffs)):
        assert np.array_equal(g, w)
    for stage in ("inverse", "dequant_inverse"):
        assert np.array_equal(_hip(c, stage, coeffs, runs=pdq)[0], _oracle(oracle_lib, c, stage, coeffs)[0]), stage
    # a run through the runs entry == the one-size batch call
    (log2, tr, first, count, cf) = c["sizes"][2]
    one = B.transform(torch.from_numpy(c["residual"]).cuda(), c["stride"], B.off32(c["off"][first:first + count]), tr, log2)
    assert np.array_equal(one.cpu().numpy(), _hip(c, "forward")[0][cf:cf + (count << (2 * log2))])
    # argument checks: nothing is launched for a call with a bad run
    L = capi.lib()
    dev = torch.zeros(1 << 16, dtype=torch.int16, device="cuda")
    off = B.off32(np.zeros(4, dtype=np.uint32))
    def run(**kw):
        r = np.zeros(1, dtype=wl.TU_RUN_DTYPE)
        r["count"], r["log2"], r["scale"], r["shift"], r["offset"] = 1, 3, 100, 20, 100
        for k, v in kw.items():
            r[k] = v
        return r
    fwd = lambda r, n=1: L.hevcasm_hip_transform_runs(dev.data_ptr(), dev.data_ptr(), 64, off.data_ptr(), r.ctypes.data, n, None)
    qnt = lambda r: L.hevcasm_hip_quantize_runs(dev.data_ptr(), dev.data_ptr(), dev.data_ptr(), r.ctypes.data, 1, None)
    inv = lambda r, f=L.hevcasm_hip_inverse_transform_add_runs: f(dev.data_ptr(), 64, dev.data_ptr(), 64, off.data_ptr(), off.data_ptr(), dev.data_ptr(), r.ctypes.data, 1, None)
    deq = lambda r: inv(r, L.hevcasm_hip_dequant_inverse_transform_add_runs)
    assert fwd(run()) == 0 and qnt(run()) == 0 and inv(run()) == 0 and deq(run(shift=5)) == 0
    assert fwd(run(), 0) == 0 and fwd(run(count=0)) == 0
    for bad in (run(log2=1), run(log2=6), run(tr_type=1), run(tr_type=2, log2=2), run(plane=1)):
        assert fwd(bad) == capi.EINVAL and inv(bad) == capi.EINVAL and deq(bad) == capi.EINVAL
    assert fwd(run(), -1) == capi.EINVAL
    assert L.hevcasm_hip_transform_runs(dev.data_ptr(), None, 64, off.data_ptr(), run().ctypes.data, 1, None) == capi.EINVAL
    assert L.hevcasm_hip_transform_runs(dev.data_ptr(), dev.data_ptr(), 64, off.data_ptr(), None, 1, None) == capi.EINVAL
    for bad in (run(shift=15), run(shift=28), run(scale=0x8000), run(scale=-1), run(offset=0x8000), run(offset=-1), run(log2=6), run(plane=2)):
        assert qnt(bad) == capi.EINVAL
    for bad in (run(shift=0), run(shift=32), run(scale=40000), run(scale=-40000)):
        assert deq(bad) == capi.EINVAL
    assert inv(run(shift=0, scale=40000)) == 0          # the plain inverse ignores the quantiser fields
    assert fwd(run(first=0xffffffff, count=2)) == capi.EINVAL and fwd(run(coeff_first=0xffffff00, count=8)) == capi.EINVAL
    torch.cuda.synchronize()
