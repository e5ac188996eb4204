"""Whole intra frames (second half): hevcasm_hip_intra_reconstruct_frames.

CPU: the oracle's frame loop is pinned for all-DC frames by refdrv (tests/test_oracle_vs_reference.py, tests/test_golden.py take the `intra_frame` group);
here it is checked against a second restatement in Python that shares no code with it (tests/intra_frame_py.py), and for its own properties.
GPU: the kernel against the oracle at full size, over batches of frames, under uneven load, across re-launches on one workspace and under graph replay."""
import ctypes as C

import numpy as np
import pytest

import cases
import golden_io
import intra_frame_py
from hevcasm_amd import capi, workload as wl


def _oracle_frame(oracle_lib, c, **kw):
    enc = c["src"].size > 0
    return oracle_lib.intra_reconstruct_frame(c["src"] if enc else None, c["stride"], None if enc else c["levels"], c["tus"], c["qp"], c["w"], c["h"], c["stride"],
                                              c["ctu"], c["cidx"], c["strong"], **kw)


@pytest.mark.parametrize("name", ["mixed_strong", "dc", "chroma", "ctu16", "decode", "all4x4_max_qp0", "all8x8_noise", "all4x4_chroma"])
def test_oracle_frame_loop_equals_the_python_restatement(name, oracle_lib):
    """Availability, substitution, the smoothing decision, the bi-linear form, DST / edge-filter conditions and the level layout, written twice."""
    c = dict(mixed_strong=lambda: cases._intra_frame_case(96, 72, 64, 11, "mixed", strong=1, split=(1.0, 0.4, 0.5, 0.4), qp=30),
             dc=lambda: cases._intra_frame_case(72, 40, 64, 12, "dc", qp=20),
             chroma=lambda: cases._intra_frame_case(48, 40, 32, 13, "mixed", cidx=1, qp=33),
             ctu16=lambda: cases._intra_frame_case(52, 36, 16, 14, "mixed", qp=26),
             decode=lambda: cases._intra_frame_case(72, 72, 64, 15, "mixed", decode=True, strong=1),
             all4x4_max_qp0=lambda: cases._intra_frame_case(40, 24, 64, 16, "mixed", split=(1.0, 1.0, 1.0, 1.0), kind="max", qp=0),
             all8x8_noise=lambda: cases._intra_frame_case(72, 40, 64, 17, "mixed", split=(1.0, 1.0, 1.0, 0.0), kind="noise", qp=22),
             all4x4_chroma=lambda: cases._intra_frame_case(40, 24, 32, 18, "angular", cidx=1, split=(1.0, 1.0, 1.0)))[name]()
    enc = c["src"].size > 0
    got = _oracle_frame(oracle_lib, c)
    tus = c["tus"].view(wl.INTRA_TU_DTYPE).reshape(-1)
    src = c["src"].reshape(c["h"], c["stride"])[:, :c["w"]] if enc else None
    rec, lev, cbf = intra_frame_py.reconstruct(src, tus, c["qp"], c["w"], c["h"], c["ctu"], c["cidx"], c["strong"], levels_in=None if enc else c["levels"])
    assert np.array_equal(got[0].reshape(c["h"], c["stride"])[:, :c["w"]], rec)
    assert np.array_equal(got[1], lev)
    if enc:
        assert np.array_equal(got[2] != 0, cbf != 0) and np.array_equal(got[2], cbf)


def test_forward_4x4_intermediates_fit_int16():
    """intra_frame.hip's register path for 4x4 TUs leaves out the int16 casts the reference applies to both forward intermediates (residual_decode.c:631-634),
    as identities.  The bound behind that: |residual| <= 255 for 8-bit video, so |pass 1| <= (A * 255 + 1) >> 1 and |pass 2| <= (A * that + 128) >> 8 with A the
    largest absolute row sum of the matrix -- both must stay inside int16 for the DST and the DCT."""
    dst = np.array([[29, 55, 74, 84], [74, 74, 0, -74], [84, -29, -74, 55], [55, -84, 74, -29]])
    dct = np.array([[64, 64, 64, 64], [83, 36, -36, -83], [64, -64, -64, 64], [36, -83, 83, -36]])
    for m in (dst, dct):
        a = int(np.abs(m).sum(axis=1).max())
        first = (a * 255 + 1) >> 1
        second = (a * first + 128) >> 8
        assert first <= 32767 and second <= 32767, (a, first, second)
        # and the extremes really are reached by sign patterns of +-255 (so the bound is the right one to check)
        worst = max(abs(int((m[r] * np.sign(m[r]) * 255).sum())) for r in range(4))
        assert worst == a * 255


def test_substitution_against_a_scan(oracle_lib):
    """orc_intra_build_neighbours on random availability patterns == the sample-by-sample scan of the text."""
    rng = np.random.default_rng(5)
    w, h = 96, 96
    recon = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    for trial in range(300):
        n = int(rng.choice([4, 8, 16, 32]))
        x, y = int(rng.integers(0, (w - n) // n + 1)) * n, int(rng.integers(0, (h - n) // n + 1)) * n
        decoded = rng.random((h // 4, w // 4)) < rng.choice([0.0, 0.1, 0.5, 0.9, 1.0])
        order = np.where(decoded, 0, -1).astype(np.int32)
        got = oracle_lib.intra_build_neighbours(recon.reshape(-1), w, w, h, order.reshape(-1), 1, x, y, n)
        want = intra_frame_py.neighbours(recon, decoded, w, h, x, y, n)
        idx = list(range(64 - 2 * n, 64 + 2 * n)) + [128]
        assert np.array_equal(got[idx], want[idx]), (trial, n, x, y)


def test_decoding_the_encoders_levels_gives_the_encoders_reconstruction(oracle_lib):
    c = cases._intra_frame_case(200, 136, 64, 21, "mixed", strong=1)
    rec, lev, cbf = _oracle_frame(oracle_lib, c)
    rec2, lev2, _ = oracle_lib.intra_reconstruct_frame(None, c["stride"], lev, c["tus"], c["qp"], c["w"], c["h"], c["stride"], c["ctu"], c["cidx"], c["strong"])
    assert np.array_equal(rec, rec2) and np.array_equal(lev, lev2)
    # cbf is the OR of the TU's levels
    tus = c["tus"].view(wl.INTRA_TU_DTYPE).reshape(-1)
    for i in (0, 7, len(tus) - 1):
        n = 1 << int(tus[i]["log2"])
        at = intra_frame_py.z_index((int(tus[i]["x"]) % 64) // 4, (int(tus[i]["y"]) % 64) // 4) * 16 + ((int(tus[i]["y"]) // 64) * 4 + int(tus[i]["x"]) // 64) * 4096
        assert cbf[i] == np.bitwise_or.reduce(lev[at:at + n * n].astype(np.int32))


def test_properties_of_a_reconstructed_frame(oracle_lib):
    """A flat frame predicts itself from the 128 of "nothing available" onwards: zero levels, exact reconstruction; finer quantisation never hurts much;
    the prediction plane is what the reconstruction is when every level is zeroed."""
    c = cases._intra_frame_case(128, 72, 64, 31, "mixed", kind="flat")
    rec, lev, cbf = _oracle_frame(oracle_lib, c)
    assert not lev.any() and not cbf.any() and (rec.reshape(72, -1)[:, :128] == 128).all()
    c = cases._intra_frame_case(128, 72, 64, 32, "mixed")
    psnr = []
    for qp in (10, 22, 34, 46):
        c["qp"] = wl.intra_qp(qp)
        rec = _oracle_frame(oracle_lib, c)[0].reshape(72, -1)[:, :128].astype(float)
        psnr.append(10 * np.log10(255 ** 2 / np.mean((rec - c["src"].reshape(72, -1)[:, :128]) ** 2)))
    assert psnr[0] > psnr[1] > psnr[2] > psnr[3] and psnr[0] > 45 and psnr[3] > 20, psnr
    rec, lev, cbf, pred = _oracle_frame(oracle_lib, c, want_pred=True)
    zero = oracle_lib.intra_reconstruct_frame(None, c["stride"], np.zeros_like(lev), c["tus"], c["qp"], c["w"], c["h"], c["stride"], c["ctu"], c["cidx"], c["strong"], want_pred=True)
    assert np.array_equal(zero[0], zero[3])      # no residual: reconstruction == prediction


def test_oracle_refuses_lists_that_do_not_tile(oracle_lib):
    c = cases._intra_frame_case(64, 64, 64, 41, "dc")
    tus = c["tus"].view(wl.INTRA_TU_DTYPE).reshape(-1).copy()
    for broken in (tus[:-1], np.concatenate([tus, tus[-1:]])):
        with pytest.raises(ValueError):
            oracle_lib.intra_reconstruct_frame(c["src"], c["stride"], None, broken, c["qp"], 64, 64, c["stride"], 64, 0, 0)
    bad = tus.copy(); bad[0]["mode"] = 35
    with pytest.raises(ValueError):
        oracle_lib.intra_reconstruct_frame(c["src"], c["stride"], None, bad, c["qp"], 64, 64, c["stride"], 64, 0, 0)


def test_reference_composition_refuses_modes_its_table_does_not_hold(refproc):
    """refdrv OP_INTRA_FRAME goes through hevcasm_get_pred_intra: the reference wires INTRA_DC only (pred_intra.c:113-139), so a planar TU is an error there --
    which is exactly why mixed-mode frames are 'unpinned'."""
    from oracle import pyref
    c = cases._intra_frame_case(64, 64, 64, 51, "planar")
    with pyref.RefProc() as p:
        with pytest.raises(RuntimeError):
            p.intra_reconstruct_frame(c["src"], c["stride"], None, c["tus"], c["qp"], 64, 64, c["stride"], 64, 0)


def test_workspace_size_and_argument_checks_without_a_gpu():
    L = capi.lib()
    assert L.hevcasm_hip_intra_frames_workspace_bytes(1920, 1080, 64, 1) >= 64 + 510 * (256 + 4)
    assert L.hevcasm_hip_intra_frames_workspace_bytes(1920, 1080, 48, 1) == 0          # CTU size
    assert L.hevcasm_hip_intra_frames_workspace_bytes(1922, 1080, 64, 1) == 0          # not a multiple of 4
    assert L.hevcasm_hip_intra_frames_workspace_bytes(1920, 1080, 64, 5000) == 0       # frames per call
    if not L.hevcasm_hip_available():
        assert L.hevcasm_hip_intra_reconstruct_frames(None, 0, 0, None, 0, 0, None, None, None, None, None, None, 64, 64, 64, 0, 0, 1, None, None) == capi.ENODEV


# ------------------------------------------------------------------------------------------------------------------------------------------- GPU

def _frames_job(cs, **kw):
    from hevcasm_amd import batch as B
    c0 = cs[0]
    tus = [c["tus"].view(wl.INTRA_TU_DTYPE).reshape(-1) for c in cs]
    return B.IntraFrames(c0["w"], c0["h"], c0["ctu"], tus, [c["ctu_first"] for c in cs], c0["qp"], c_idx=c0["cidx"], strong=c0["strong"], stride=c0["stride"], **kw)


def _check_frames(job, cs, oracle_lib, what):
    import torch
    torch.cuda.synchronize()
    job.status()
    plane = cs[0]["stride"] * cs[0]["h"]
    rec, lev, cbf = job.recon.cpu().numpy(), job.levels.cpu().numpy(), job.cbf.cpu().numpy()
    per_frame = lev.size // len(cs)
    t0 = 0
    for f, c in enumerate(cs):
        want = _oracle_frame(oracle_lib, c)
        cases.assert_same("%s frame %d recon" % (what, f), (rec[f * plane:(f + 1) * plane],), (want[0],))
        cases.assert_same("%s frame %d levels" % (what, f), (lev[f * per_frame:(f + 1) * per_frame],), (want[1],))
        if c["src"].size:
            cases.assert_same("%s frame %d cbf" % (what, f), (cbf[t0:t0 + len(c["tus"])],), (want[2],))
        t0 += len(c["tus"])


@pytest.mark.gpu
def test_hip_1080p_all_dc_frame_equals_the_reference_composition(oracle_lib):
    """VERDICT round 5, 'done' (a): an all-DC 1080p frame reconstructs bit-identically to the composition of the reference's own C functions -- the committed
    golden hash was made by refdrv OP_INTRA_FRAME; the oracle (checked against that hash on the CPU) gives the arrays to compare with."""
    import torch
    name, family, c = [x for x in cases.intra_frame_cases() if x[0] == "iframe_dc_1920x1080_ctu64"][0]
    got = cases.run_hip(family, c)
    assert golden_io.sha(got) == golden_io.hashes()[name]["outputs"], "1080p all-DC frame differs from the reference's composition"
    cases.assert_same(name, got, cases.run_oracle(family, c))


@pytest.mark.gpu
def test_hip_1080p_mixed_frame_and_its_chroma_planes_equal_the_oracle(oracle_lib):
    """'done' (b): a mixed-mode frame (unpinned) at 1080p, luma with strong smoothing, and a 960x540 chroma plane at CTU 32; prediction plane included."""
    import torch
    from hevcasm_amd import batch as B
    c = cases._intra_frame_case(1920, 1080, 64, 61, "mixed", strong=1, qp=30)
    job = _frames_job([c], want_pred=True)
    job.run(B.u8(c["src"]))
    _check_frames(job, [c], oracle_lib, "1080p mixed")
    want_pred = _oracle_frame(oracle_lib, c, want_pred=True)[3]
    assert np.array_equal(job.pred.cpu().numpy(), want_pred)
    cc = cases._intra_frame_case(960, 540, 32, 62, "mixed", cidx=1, qp=32)
    job = _frames_job([cc])
    job.run(B.u8(cc["src"]))
    _check_frames(job, [cc], oracle_lib, "540p chroma")


@pytest.mark.gpu
def test_hip_batches_of_frames_uneven_load_and_relaunch_on_one_workspace(oracle_lib):
    """Frames of one call are independent and interleaved inside every wavefront step.  24 frames whose CTUs cost wildly different amounts (all-4x4 CTUs next to
    single 32x32 TUs, per frame), three launches on the SAME workspace with different content (stale granules of the previous launch must not count: the memset
    node), then the decoder on the levels just produced.  Every byte of every frame against the oracle."""
    import torch
    from hevcasm_amd import batch as B
    w, h, ctu = 320, 192, 64
    rng = np.random.default_rng(7)
    for launch in range(3):
        cs = []
        for f in range(24):
            split = [(1.0, 1.0, 1.0, 1.0), (1.0, 0.0), (1.0, 0.9, 0.1, 0.9), (1.0, 0.5, 0.5, 0.5)][int(rng.integers(0, 4))]
            cs.append(cases._intra_frame_case(w, h, ctu, 1000 * launch + 7 * f, "mixed", split=split, qp=27, strong=1,
                                              kind=["video", "noise"][f % 2]))
        if launch == 0:
            job = _frames_job(cs)
        else:      # same geometry, new TU lists: rebuild the descriptors but keep the workspace (and dirty it with the last launch's tags)
            ws = job.workspace
            job = _frames_job(cs)
            job.workspace = ws
        src = torch.cat([B.u8(c["src"]) for c in cs])
        job.run(src)
        _check_frames(job, cs, oracle_lib, "launch %d" % launch)
    enc_rec = job.recon.clone()
    job.recon.zero_()
    job.run(None)
    torch.cuda.synchronize()
    job.status()
    assert torch.equal(job.recon, enc_rec)


@pytest.mark.gpu
def test_hip_hundreds_of_frames_take_the_two_waves_per_simd_build(oracle_lib):
    """The loop exists twice (intra_frame.hip: intra_frames_kernel at one wave per SIMD, intra_frames_dense_kernel held to 256 registers for two); the host takes the
    second when the call's concurrency -- frames x min(rows, ceil(cols / 2)) -- exceeds 4 x CUs.  Every other test stays below that, so: 560 frames of 256x128
    (2 CTUs of a frame at a time: 1120 > 1024 on a 256-CU device), eight distinct frames repeated, encoder then decoder, every frame against the oracle."""
    import torch
    from hevcasm_amd import batch as B
    w, h, ctu, F = 256, 128, 64, 560
    kinds = [cases._intra_frame_case(w, h, ctu, 3000 + k, "mixed", strong=1, qp=24, kind=["video", "noise"][k % 2],      # (one QP and smoothing flag per call)
                                     split=[(1.0, 0.7, 0.5, 0.35), (1.0, 1.0, 1.0, 0.5), (1.0, 0.3, 0.9, 0.9), (1.0, 1.0, 0.2, 0.2)][k % 4]) for k in range(8)]
    cs = [kinds[(f * 5 + f // 8) % 8] for f in range(F)]
    job = _frames_job(cs)
    assert F * min(job.rows, (job.cols + 1) // 2) > 4 * torch.cuda.get_device_properties(0).multi_processor_count
    src = torch.cat([B.u8(k["src"]) for k in kinds])
    plane = kinds[0]["stride"] * h
    src_all = torch.cat([src[((f * 5 + f // 8) % 8) * plane:((f * 5 + f // 8) % 8 + 1) * plane] for f in range(F)])
    job.run(src_all)
    torch.cuda.synchronize()
    job.status()
    want = [_oracle_frame(oracle_lib, k) for k in kinds]
    rec, lev, cbf = job.recon.cpu().numpy(), job.levels.cpu().numpy(), job.cbf.cpu().numpy()
    per_frame = lev.size // F
    t0 = 0
    for f in range(F):
        k = (f * 5 + f // 8) % 8
        assert np.array_equal(rec[f * plane:(f + 1) * plane], want[k][0]), "frame %d (kind %d): reconstruction" % (f, k)
        assert np.array_equal(lev[f * per_frame:(f + 1) * per_frame], want[k][1]), "frame %d: levels" % f
        n = len(kinds[k]["tus"])
        assert np.array_equal(cbf[t0:t0 + n], want[k][2]), "frame %d: cbf" % f
        t0 += n
    enc = job.recon.clone()
    job.recon.zero_()
    job.run(None)                 # the decoder's build of the same loop, on the levels just written
    torch.cuda.synchronize()
    job.status()
    assert torch.equal(job.recon, enc)


@pytest.mark.gpu
def test_hip_intra_frames_replay_as_a_hip_graph(oracle_lib):
    """The call only enqueues (memset node + two launches): captured once, replayed on new content of the resident planes."""
    import torch
    from hevcasm_amd import batch as B
    cs = [cases._intra_frame_case(256, 128, 64, 71 + f, "mixed") for f in range(4)]
    job = _frames_job(cs)
    src = torch.cat([B.u8(c["src"]) for c in cs])
    job.run(src)      # warm-up outside capture (module load)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        job.run(src)
    for round_ in range(2):
        new = [dict(c, src=wl.frame_2d(256, 128, c["stride"], 900 + 10 * round_ + f)) for f, c in enumerate(cs)]
        src.copy_(torch.cat([B.u8(c["src"]) for c in new]))
        job.recon.zero_(); job.levels.zero_()
        g.replay()
        _check_frames(job, new, oracle_lib, "replay %d" % round_)


@pytest.mark.gpu
def test_hip_intra_frames_flag_malformed_lists_and_refuse_bad_arguments():
    import torch
    from hevcasm_amd import batch as B
    L = capi.lib()
    c = cases._intra_frame_case(128, 64, 64, 81, "dc")
    tus = c["tus"].view(wl.INTRA_TU_DTYPE).reshape(-1).copy()
    tus[3]["mode"] = 77
    tus[5]["x"] += 2
    bad = dict(c, tus=tus.view(np.uint8).reshape(-1, 8))
    job = _frames_job([bad])
    job.run(B.u8(c["src"]))
    with pytest.raises(capi.HevcasmError, match="malformed TU"):
        job.status()
    good = _frames_job([c])
    good.run(B.u8(c["src"]))
    good.status()
    p = good.recon.data_ptr()
    qp = good.qp.copy(); qp[0, 1] = 28
    args = lambda **k: [k.get("recon", p), 128, 128 * 64, p, 128, 128 * 64, good.levels.data_ptr(), None, None, good.tus.data_ptr(), good.ctu_first.data_ptr(),
                        k.get("qp", good.qp).ctypes.data, k.get("w", 128), 64, k.get("ctu", 64), 0, 0, 1, k.get("ws", good.workspace.data_ptr()), None]
    f = L.hevcasm_hip_intra_reconstruct_frames
    assert f(*args()) == 0
    assert f(*args(ctu=48)) == capi.EINVAL and f(*args(w=130)) == capi.EINVAL and f(*args(qp=qp)) == capi.EINVAL
    assert f(*args(ws=good.workspace.data_ptr() + 4)) == capi.EINVAL and f(*args(recon=None)) == capi.EINVAL
    torch.cuda.synchronize()
