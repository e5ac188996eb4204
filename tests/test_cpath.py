"""The library's own C path (HEVCASM_C_REF / C_OPT table entries, hevcasm_amd/csrc/host/cpath.c)
against the oracle, through the reference's calling pattern: populate a table, fetch the slot,
call the function pointer per block.  CPU only."""
import ctypes as C

import numpy as np
import pytest

import cases
from hevcasm_amd import capi, workload as wl

PD = C.c_ssize_t
VP = C.c_void_p


def _slot(populate, nslots, index, mask=capi.C_REF, extra=()):
    buf = (C.c_void_p * nslots)()
    getattr(capi.lib(), populate)(buf, mask, *extra)
    assert buf[index], "NULL table entry"
    return buf[index]


def _ptr(a, byte_off=0):
    return a.ctypes.data + int(byte_off)


def _sad_slot_index(w, h):
    order = [(64, 64), (64, 32), (32, 64), (32, 32), (32, 16), (16, 32), (16, 16), (16, 8), (8, 16), (8, 8), (8, 4)]
    return order.index((w, h)) if (w, h) in order else 11


def test_sad_and_multiref_and_ssd(oracle_lib):
    for name, family, c in cases.sad_cases(n=12) + cases.ssd_cases(n=12):
        w, h, n = c["w"], c["h"], c["n"]
        src, ref = np.ascontiguousarray(c["src"]), np.ascontiguousarray(c["ref"])
        want = cases.run_oracle(family, c)[0]
        if family == "sad":
            f = C.CFUNCTYPE(C.c_int, VP, PD, VP, PD, C.c_uint32)(_slot("hevcasm_populate_sad", 12, _sad_slot_index(w, h)))
            got = [f(_ptr(src, c["src_off"][i]), c["stride_src"], _ptr(ref, c["ref_off"][i]), c["stride_ref"], capi.rect(w, h)) for i in range(n)]
            assert np.array_equal(np.array(got), want), name
        elif family == "sad4":
            f = C.CFUNCTYPE(None, VP, PD, VP, PD, VP, C.c_uint32)(_slot("hevcasm_populate_sad_multiref", 257, ((w >> 2) - 1) * 16 + (h >> 2) - 1))
            for i in range(n):
                refs = (C.c_void_p * 4)(*[_ptr(ref, c["ref_off"][4 * i + k]) for k in range(4)])
                out = (C.c_int * 4)()
                f(_ptr(src, c["src_off"][i]), c["stride_src"], refs, c["stride_ref"], out, capi.rect(w, h))
                assert list(out) == list(want[i]), name
        else:
            if w != h:
                continue
            f = C.CFUNCTYPE(C.c_int, VP, PD, VP, PD, C.c_int, C.c_int)(_slot("hevcasm_populate_ssd", 5, int(np.log2(w)) - 2))
            got = [f(_ptr(src, c["src_off"][i]), c["stride_src"], _ptr(ref, c["ref_off"][i]), c["stride_ref"], w, h) for i in range(n)]
            assert np.array_equal(np.array(got), want), name


def test_sad_x86_simd_entries(oracle_lib):
    """The SSE2 / AVX2 table entries (csrc/host/sad_x86.c: the reference's x86 kernels written with intrinsics) against the
    oracle, selected with single-bit masks as the reference's self-test does; every size and, for the 4-way AVX2 kernel, every
    multiple-of-4 width; misaligned planes; NULL where the reference's table has no entry for the bit."""
    L = capi.lib()
    have = L.hevcasm_instruction_set_support()
    if not (have & capi.SSE2):
        pytest.skip("not an x86 host")
    named = [(64, 64), (64, 32), (32, 64), (32, 32), (32, 16), (16, 32), (16, 16), (16, 8), (8, 16), (8, 8), (8, 4)]
    sizes = wl.SAD_PARTITIONS + [(w, 20) for w in range(4, 68, 4)]
    for (w, h) in sizes:
        c = wl.sad_case(w, h, 9, 7000 + 64 * w + h, ways=4)
        src, ref = np.ascontiguousarray(c["src"]), np.ascontiguousarray(c["ref"])
        want4 = cases.run_oracle("sad4", c)[0]
        ro = c["ref_off"].reshape(-1, 4)
        for mask in (capi.SSE2, capi.AVX2):
            if not (have & mask):
                continue
            # four references
            buf = (C.c_void_p * 257)()
            L.hevcasm_populate_sad_multiref(buf, mask)
            slot = buf[((w >> 2) - 1) * 16 + (h >> 2) - 1]
            expect_entry = (w, h) in named if mask == capi.SSE2 else True
            assert bool(slot) == expect_entry, (w, h, mask)
            if slot:
                f = C.CFUNCTYPE(None, VP, PD, VP, PD, VP, C.c_uint32)(slot)
                for i in range(c["n"]):
                    refs = (C.c_void_p * 4)(*[_ptr(ref, ro[i, k]) for k in range(4)])
                    out = (C.c_int * 4)()
                    f(_ptr(src, c["src_off"][i]), c["stride_src"], refs, c["stride_ref"], out, capi.rect(w, h))
                    assert list(out) == list(want4[i]), (w, h, mask, i)
            # one reference
            buf1 = (C.c_void_p * 12)()
            L.hevcasm_populate_sad(buf1, mask)
            if (w, h) in named:
                slot1 = buf1[_sad_slot_index(w, h)]
                assert bool(slot1) == (mask == capi.SSE2 or w >= 32), (w, h, mask)
                if slot1:
                    f1 = C.CFUNCTYPE(C.c_int, VP, PD, VP, PD, C.c_uint32)(slot1)
                    got = [f1(_ptr(src, c["src_off"][i]), c["stride_src"], _ptr(ref, ro[i, 0]), c["stride_ref"], capi.rect(w, h)) for i in range(c["n"])]
                    assert got == [int(v) for v in want4[:, 0]], (w, h, mask)
            else:
                assert not buf1[11], "sadGeneric must be NULL for an x86-only mask"   # reference sad.c:80-85: only a C bit fills it


def test_hadamard_satd(oracle_lib):
    for name, family, c in cases.satd_cases(n=20):
        size = c["size"]
        f = C.CFUNCTYPE(C.c_int, VP, PD, VP, PD)(_slot("hevcasm_populate_hadamard_satd", 3, int(np.log2(size)) - 1))
        src, ref = np.ascontiguousarray(c["src"]), np.ascontiguousarray(c["ref"])
        got = [f(_ptr(src, c["src_off"][i]), c["stride_src"], _ptr(ref, c["ref_off"][i]), c["stride_ref"]) for i in range(c["n"])]
        assert np.array_equal(np.array(got), cases.run_oracle(family, c)[0]), name


def test_pred_uni_and_bi(oracle_lib):
    _check_pred(capi.C_REF)


def _check_pred(mask):
    for name, family, c in cases.pred_uni_cases(n=6) + cases.pred_bi_cases(n=4):
        taps, w, h, n = c["taps"], c["w"], c["h"], c["n"]
        want = cases.run_oracle(family, c)[0]
        dst = np.zeros(c["dst_bytes"], dtype=np.uint8)
        ref0 = np.ascontiguousarray(c["ref0"])
        for i in range(n):
            x0, y0 = int(c["frac0"][i]) & 15, int(c["frac0"][i]) >> 4
            if family == "pred_uni":
                idx = (((taps // 4 - 1) * 9 + (w + taps - 1) // taps) * 2 + (1 if x0 else 0)) * 2 + (1 if y0 else 0)
                f = C.CFUNCTYPE(None, VP, PD, VP, PD, C.c_int, C.c_int, C.c_int, C.c_int)(_slot("hevcasm_populate_pred_uni_8to8", 72, idx, mask))
                f(_ptr(dst, c["dst_off"][i]), c["stride_dst"], _ptr(ref0, c["ref0_off"][i]), c["stride_ref"], w, h, x0, y0)
            else:
                ref1 = np.ascontiguousarray(c["ref1"])
                x1, y1 = int(c["frac1"][i]) & 15, int(c["frac1"][i]) >> 4
                idx = ((taps // 4 - 1) * 5 + (w + 2 * taps - 1) // (2 * taps)) * 2 + (1 if (x0 or y0 or x1 or y1) else 0)
                f = C.CFUNCTYPE(None, VP, PD, VP, VP, PD, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)(_slot("hevcasm_populate_pred_bi_8to8", 20, idx, mask))
                f(_ptr(dst, c["dst_off"][i]), c["stride_dst"], _ptr(ref0, c["ref0_off"][i]), _ptr(ref1, c["ref1_off"][i]), c["stride_ref"], w, h, x0, y0, x1, y1)
        got = cases._dst_rows(dst, c["stride_dst"], c["dst_off"], w, h)
        cases.assert_same(name, (got,), (want,))


def test_transforms(oracle_lib):
    assert _check_transforms(capi.C_REF) == (0, 0)


def _check_transforms(mask, encoder=0):
    """Every inverse / forward case through the table entries of `mask`; returns how many (inverse, forward) cases found a NULL slot."""
    null_inv = null_fwd = 0
    for name, family, c in cases.inverse_cases(n=10) + cases.forward_cases(n=10):
        tr, lg, n = c["tr_type"], c["log2n"], c["n"]
        size = 1 << lg
        idx = 0 if tr else lg - 1
        buf = (C.c_void_p * 5)()
        if family == "inverse":
            capi.lib().hevcasm_populate_inverse_transform_add(buf, mask, encoder)
        else:
            capi.lib().hevcasm_populate_transform(buf, mask)
        if not buf[idx]:
            null_inv += family == "inverse"
            null_fwd += family != "inverse"
            continue
        want = cases.run_oracle(family, c)[0]
        if family == "inverse":
            f = C.CFUNCTYPE(None, VP, PD, VP, PD, VP)(buf[idx])
            dst = np.zeros(c["dst_bytes"], dtype=np.uint8)
            pred, coeffs = np.ascontiguousarray(c["pred"]), np.ascontiguousarray(c["coeffs"])
            for i in range(n):
                f(_ptr(dst, c["dst_off"][i]), c["stride_dst"], _ptr(pred, c["pred_off"][i]), c["stride_pred"], _ptr(coeffs, 2 * size * size * i))
            cases.assert_same(name, (cases._dst_rows(dst, c["stride_dst"], c["dst_off"], size, size),), (want,))
        else:
            f = C.CFUNCTYPE(None, VP, VP, PD)(buf[idx])
            src = np.ascontiguousarray(c["src"])
            out = np.zeros(n * size * size, dtype=np.int16)
            for i in range(n):
                f(_ptr(out, 2 * size * size * i), _ptr(src, 2 * int(c["src_off"][i])), c["src_stride"])
            cases.assert_same(name, (out,), (want,))
    return null_inv, null_fwd


def test_quant_family(oracle_lib):
    _check_quant(capi.C_REF)


def _check_quant(mask):
    for name, family, c in cases.quant_cases():
        want = cases.run_oracle(family, c)
        if family == "dequant":
            f = C.CFUNCTYPE(None, VP, VP, C.c_int, C.c_int, C.c_int)(_slot("hevcasm_populate_quantize_inverse", 1, 0, mask))
            src = np.ascontiguousarray(c["src"])
            out = np.zeros_like(src)
            f(_ptr(out), _ptr(src), c["scale"], c["shift"], src.size)
            cases.assert_same(name, (out,), want)
        elif family == "quant":
            f = C.CFUNCTYPE(C.c_int, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int)(_slot("hevcasm_populate_quantize", 1, 0, mask))
            src = np.ascontiguousarray(c["src"])
            out = np.zeros_like(src)
            npb = c["n_per_block"]
            cbf = [f(_ptr(out, 2 * npb * i), _ptr(src, 2 * npb * i), c["scale"], c["shift"], c["offset"], npb) for i in range(src.size // npb)]
            cases.assert_same(name, (out, np.array(cbf, dtype=np.int32)), want)
        else:
            size = c["ncbs"]
            f = C.CFUNCTYPE(None, VP, PD, VP, PD, VP, C.c_int)(_slot("hevcasm_populate_quantize_reconstruct", 4, int(np.log2(size)) - 2, mask))
            rec = np.zeros(c["rec_bytes"], dtype=np.uint8)
            pred, res = np.ascontiguousarray(c["pred"]), np.ascontiguousarray(c["res"])
            for i in range(len(c["rec_off"])):
                f(_ptr(rec, c["rec_off"][i]), c["stride_rec"], _ptr(pred, c["pred_off"][i]), c["stride_pred"], _ptr(res, 2 * size * size * i), size)
            cases.assert_same(name, (cases._dst_rows(rec, c["stride_rec"], c["rec_off"], size, size),), want)


def _x86(bits):
    have = capi.lib().hevcasm_instruction_set_support()
    if (have & bits) != bits:
        pytest.skip("host lacks the instruction sets %#x" % bits)


def test_pred_x86_simd_entries(oracle_lib):
    """The SSE2 / SSE4.1 prediction entries (csrc/host/pred_x86.c: the reference's x86 kernels written with intrinsics, pred_inter_a.asm:174-600)
    against the oracle on every prediction case -- all PU sizes, every fraction pair, both clip rails, 2-wide chroma PUs -- and the table shape of
    single-bit masks: SSE2 alone fills only the integer-vector slots, SSE4.1 alone only the fractional ones (reference pred_inter.c:296-346, :570-593)."""
    _x86(capi.SSE2 | capi.SSE41)
    _check_pred(capi.SSE2 | capi.SSE41)
    L = capi.lib()
    for mask, want_copy, want_filter in ((capi.SSE2, True, False), (capi.SSE41, False, True)):
        uni, bi = (C.c_void_p * 72)(), (C.c_void_p * 20)()
        L.hevcasm_populate_pred_uni_8to8(uni, mask)
        L.hevcasm_populate_pred_bi_8to8(bi, mask)
        for t in range(2):
            for wc in range(9):
                for xf in range(2):
                    for yf in range(2):
                        assert bool(uni[((t * 9 + wc) * 2 + xf) * 2 + yf]) == (want_filter if (xf or yf) else want_copy), (mask, t, wc, xf, yf)
            for wc in range(5):
                for fr in range(2):
                    assert bool(bi[(t * 5 + wc) * 2 + fr]) == (want_filter if fr else want_copy), (mask, t, wc, fr)


def test_pred_x86_truncation_of_the_14_bit_intermediate(oracle_lib):
    """Bi-prediction keeps 14-bit intermediates in int16 BY TRUNCATION (reference pred_inter.c:125 stores an int into an int16_t): a pattern
    whose second-pass sum exceeds 32767 -- 255 under every positive tap, 0 under every negative one, in both directions -- must wrap exactly as
    the C path does, not saturate."""
    _x86(capi.SSE2 | capi.SSE41)
    from oracle import pyoracle
    taps, w, h, stride = 8, 8, 8, 64
    pos = [k for k in range(8) if [-1, 4, -11, 40, 40, -11, 4, -1][k] > 0]
    ref = np.zeros((h + 24) * stride, dtype=np.uint8).reshape(-1, stride)
    for y in range(ref.shape[0]):
        for x in range(stride):
            # pixel (x, y) is 255 when the taps that see it from output (4, 4)+origin are positive in both directions, complementary rows give the negative lobes
            ref[y, x] = 255 if ((x - 1) % 8 in pos) == ((y - 1) % 8 in pos) else 0
    flat = np.ascontiguousarray(ref.reshape(-1))
    off = np.array([8 * stride + 8], dtype=np.uint32)
    frac = np.array([2 | (2 << 4)], dtype=np.uint8)
    want = pyoracle.pred_bi_batch(taps, w * h, w, flat, flat, stride, np.array([0], dtype=np.uint32), off, off, frac, frac, w, h)
    buf = (C.c_void_p * 20)()
    capi.lib().hevcasm_populate_pred_bi_8to8(buf, capi.SSE41)
    f = C.CFUNCTYPE(None, VP, PD, VP, VP, PD, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)(buf[(1 * 5 + 1) * 2 + 1])
    dst = np.zeros(w * h, dtype=np.uint8)
    f(_ptr(dst), w, _ptr(flat, off[0]), _ptr(flat, off[0]), stride, w, h, 2, 2, 2, 2)
    assert np.array_equal(dst, want)


def test_transform_x86_simd_entries(oracle_lib):
    """The SSSE3 / AVX2 transform entries (csrc/host/transform_x86.c) against the oracle on every transform case (full-range, +-32767, alternating,
    DC-only coefficients; 9-bit and full-range residuals), and the table shape of single-bit masks: SSSE3 = inverse 8x8 / 16x16 and forward 16x16,
    AVX2 = every inverse size (reference residual_decode.c:475-497, :918-923)."""
    _x86(capi.SSSE3 | capi.AVX2)
    n_inv = len(cases.inverse_cases(n=10))
    per_size_inv = n_inv // 5
    null_inv, null_fwd = _check_transforms(capi.SSSE3)
    assert null_inv == 3 * per_size_inv and null_fwd == 4 * (len(cases.forward_cases(n=10)) // 5)
    assert _check_transforms(capi.AVX2)[0] == 0
    assert _check_transforms(capi.AVX2, encoder=1)[0] == 0
    assert _check_transforms(capi.C_REF | capi.SSSE3 | capi.AVX2) == (0, 0)


def test_quant_and_ssd_x86_simd_entries(oracle_lib):
    """The SSE4.1 (de)quantiser / reconstruction entries and the AVX SSD entries (csrc/host/quant_x86.c) against the oracle: reference parameters,
    HEVC-style sweeps, src = -32768 (where the reference's own pabsw assembly differs from its C: SURVEY appendix A), full-range residuals."""
    _x86(capi.SSE41 | capi.AVX)
    _check_quant(capi.SSE41)
    # zero inputs under a rounding offset >= 1.0: C path and SSE4.1 entry agree (both give +1 for x == 0 at offset 70000, shift 20)
    src = np.zeros(64, dtype=np.int16)
    src[1::2] = np.arange(-16, 16, dtype=np.int16)
    outs = []
    for mask in (capi.C_REF, capi.SSE41):
        f = C.CFUNCTYPE(C.c_int, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int)(_slot("hevcasm_populate_quantize", 1, 0, mask))
        out = np.zeros_like(src)
        cbf = f(_ptr(out), _ptr(src), 16384, 20, 70000, 64)
        outs.append((out, cbf))
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1] and outs[0][0][0] == 1
    buf = (C.c_void_p * 5)()
    capi.lib().hevcasm_populate_ssd(buf, capi.AVX)
    assert [bool(v) for v in buf] == [False, False, True, True, True]        # reference ssd.c:80-85
    for name, family, c in cases.ssd_cases(n=12):
        w, h = c["w"], c["h"]
        if w != h or w < 16:
            continue
        src, ref = np.ascontiguousarray(c["src"]), np.ascontiguousarray(c["ref"])
        f = C.CFUNCTYPE(C.c_int, VP, PD, VP, PD, C.c_int, C.c_int)(buf[int(np.log2(w)) - 2])
        got = [f(_ptr(src, c["src_off"][i]), c["stride_src"], _ptr(ref, c["ref_off"][i]), c["stride_ref"], w, h) for i in range(c["n"])]
        assert np.array_equal(np.array(got), cases.run_oracle(family, c)[0]), name
