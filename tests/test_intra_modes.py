"""Intra prediction beyond DC -- PARITY UNPINNED: the reference has no C for planar / angular prediction nor for the neighbour smoothing (only f265 assembly for
4x4 / 8x8 that cannot be assembled here, f265/intra.asm:216-1394) and holds no vectors.  What can be checked is checked:
  CPU: the oracle's restatement of H.265 8.4.4.2.3-6 == a second, independent restatement (tests/intra_spec.py) == the library's host C entries;
       mode 1 through the new all-mode path == the reference-pinned DC function; structural properties the text implies.
  GPU: the HIP kernels == the oracle on every case."""
import ctypes as C

import numpy as np
import pytest

import cases
import intra_spec
from hevcasm_amd import capi, workload as wl
from oracle import pyoracle


def _packings():
    for log2 in (2, 3, 4, 5):
        for edge in (0, 1):
            if not (edge and log2 == 5):
                yield log2, edge, (8 << 8) | (log2 << 1) | edge


def test_oracle_equals_the_second_restatement(oracle_lib):
    rng = np.random.default_rng(7)
    for log2, edge, packed in _packings():
        size = 1 << log2
        for kind in ("noise", "extremes"):
            nb = rng.integers(0, 256, 129 * 35, dtype=np.uint8) if kind == "noise" else rng.choice(np.array([0, 255], dtype=np.uint8), 129 * 35)
            got = pyoracle.pred_intra_modes_batch(nb, 129, np.arange(35, dtype=np.uint8), packed).reshape(35, size, size)
            for mode in range(35):
                assert np.array_equal(got[mode], intra_spec.predict(nb[129 * mode: 129 * mode + 129], mode, log2, edge)), (log2, edge, mode, kind)


def test_mode_1_of_the_all_mode_path_is_the_pinned_dc(oracle_lib):
    """The one pinned mode: the generic entry with mode 1 must give what orc_pred_intra_dc (== the reference's C, golden-tested) gives."""
    for name, family, c in cases.intra_cases():
        want = cases.run_oracle(family, c)[0]
        got = pyoracle.pred_intra_modes_batch(c["neighbours"], c["neighbour_stride"], np.ones(c["n"], dtype=np.uint8), c["packed"]) if c["neighbour_stride"] >= 129 else None
        if got is not None:
            assert np.array_equal(got, want), name


def test_structural_properties(oracle_lib):
    """What the text implies whatever the sample values: pure vertical / horizontal replicate an edge, the three pure diagonals are shifts, planar of a
    constant record is that constant, the boundary smoothing of modes 10 / 26 touches the first row / column only."""
    rng = np.random.default_rng(11)
    for log2, edge, packed in _packings():
        n = 1 << log2
        nb = rng.integers(0, 256, 129, dtype=np.uint8)
        top, left, corner = nb[64:64 + 2 * n].astype(int), nb[63::-1][:2 * n].astype(int), int(nb[128])
        pred = lambda mode: pyoracle.pred_intra_modes_batch(nb, 129, np.array([mode], dtype=np.uint8), packed).reshape(n, n).astype(int)
        v, h = pred(26), pred(10)
        assert np.array_equal(v[:, 1:], np.tile(top[1:n], (n, 1))) and np.array_equal(h[1:, :], np.tile(left[1:n, None], (1, n)))
        if edge:
            assert np.array_equal(v[:, 0], np.clip(top[0] + ((left[:n] - corner) >> 1), 0, 255)) and np.array_equal(h[0, :], np.clip(left[0] + ((top[:n] - corner) >> 1), 0, 255))
        else:
            assert np.array_equal(v[:, 0], np.full(n, top[0])) and np.array_equal(h[0, :], np.full(n, left[0]))
        y, x = np.mgrid[0:n, 0:n]
        assert np.array_equal(pred(34), top[x + y + 1]) and np.array_equal(pred(2), left[x + y + 1])
        d = pred(18)
        assert np.array_equal(d, np.where(x > y, top[np.maximum(x - y - 1, 0)], np.where(x == y, corner, left[np.maximum(y - x - 1, 0)])))
        flat = np.full(129, 77, dtype=np.uint8)
        for mode in (0, 1, 5, 14, 22, 30):
            assert (pyoracle.pred_intra_modes_batch(flat, 129, np.array([mode], dtype=np.uint8), packed) == 77).all()


def test_transpose_symmetry_of_the_mode_set(oracle_lib):
    """The mode set is symmetric about the diagonal: with the two edges swapped, mode m predicts the transpose of what mode 36 - m predicts (2 <-> 34, 10 <-> 26,
    18 <-> 18), and planar / DC predict their own transposes.  The oracle restates the text's two branches (vertical, horizontal) separately, so this checks
    one against the other; the boundary smoothing is included (it swaps with the modes)."""
    rng = np.random.default_rng(13)
    for log2, edge, packed in _packings():
        n = 1 << log2
        nb = rng.integers(0, 256, 129, dtype=np.uint8)
        swapped = nb.copy()
        swapped[64:64 + 2 * n] = nb[63::-1][:2 * n]            # new top = old left
        swapped[63::-1][:2 * n] = nb[64:64 + 2 * n]            # new left = old top
        for mode in range(35):
            mirror = mode if mode < 2 else 36 - mode
            a = pyoracle.pred_intra_modes_batch(nb, 129, np.array([mode], dtype=np.uint8), packed).reshape(n, n)
            b = pyoracle.pred_intra_modes_batch(swapped, 129, np.array([mirror], dtype=np.uint8), packed).reshape(n, n)
            assert np.array_equal(a, b.T), (log2, edge, mode)


def test_neighbour_filter_oracle_equals_the_second_restatement(oracle_lib):
    for name, family, c in cases.intra_filter_cases():
        got = pyoracle.intra_filter_neighbours_batch(c["neighbours"], c["neighbour_stride"], c["log2"], c["strong"], c["n"], out_stride=129).reshape(c["n"], 129)
        n = 1 << c["log2"]
        bilinear = 0
        for i in range(c["n"]):
            rec = c["neighbours"][i * c["neighbour_stride"]: i * c["neighbour_stride"] + 129]
            left, top, corner = intra_spec.filter_record(rec, c["log2"], c["strong"])
            assert list(got[i, 64:64 + 2 * n]) == top and list(got[i, 63::-1][:2 * n]) == left and got[i, 128] == corner, (name, i)
            r = rec.astype(int)
            bilinear += bool(c["strong"]) and n == 32 and abs(r[128] + r[127] - 2 * r[95]) < 8 and abs(r[128] + r[0] - 2 * r[32]) < 8
        if c["log2"] == 5 and c["strong"]:
            assert bilinear == 10, "the ten flat-edged records (and only they) meet the text's condition for the bi-linear form"
            # ... and that form differs from [1 2 1] on them somewhere, so the case does tell the two apart
            plain = pyoracle.intra_filter_neighbours_batch(c["neighbours"], c["neighbour_stride"], 5, 0, c["n"], out_stride=129).reshape(c["n"], 129)
            assert (plain[:10] != got[:10]).any() and np.array_equal(plain[10:], got[10:])


def test_plain_populate_keeps_the_reference_table_dc_only():
    """hevcasm_populate_pred_intra fills exactly what the reference fills (pred_intra.c:113-139): mode 1 of every size, plain and edge-filtered; every other
    slot NULL -- a non-NULL slot means "the reference's function" (round 6)."""
    L = capi.lib()
    table = (C.c_void_p * (4 * 35 + 3))()
    L.hevcasm_populate_pred_intra(table, capi.C_REF)
    filled = {i for i in range(4 * 35 + 3) if table[i]}
    assert filled == {k * 35 + 1 for k in range(4)} | {4 * 35 + k for k in range(3)}


def test_host_table_fills_every_mode_and_its_c_entries_equal_the_oracle(oracle_lib):
    """hevcasm_populate_pred_intra_all_modes (the opt-in) under the C bits: p[log2 - 2][mode] for all 35 modes, each equal to the oracle."""
    L = capi.lib()
    table = (C.c_void_p * (4 * 35 + 3))()
    L.hevcasm_populate_pred_intra_all_modes(table, capi.C_REF)
    assert all(table[i] for i in range(4 * 35 + 3))
    fn = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.c_int)
    rng = np.random.default_rng(3)
    for log2, edge, packed in _packings():
        size = 1 << log2
        nb = np.ascontiguousarray(rng.integers(0, 256, 144, dtype=np.uint8))
        want = pyoracle.pred_intra_modes_batch(np.tile(nb, 35), 144, np.arange(35, dtype=np.uint8), packed).reshape(35, size * size)
        for mode in range(35):
            slot = 4 * 35 + (log2 - 2) if (mode == 1 and edge) else (log2 - 2) * 35 + mode
            out = np.zeros(size * size, dtype=np.uint8)
            fn(table[slot])(out.ctypes.data, nb.ctypes.data, mode, packed)
            assert np.array_equal(out, want[mode]), (log2, edge, mode)


@pytest.mark.gpu
@pytest.mark.parametrize("group", sorted(cases.UNPINNED_CASE_LISTS))
def test_hip_equals_oracle_unpinned(group):
    for name, family, c in cases.UNPINNED_CASE_LISTS[group]():
        cases.assert_same(name, cases.run_hip(family, c), cases.run_oracle(family, c))


@pytest.mark.gpu
def test_hip_per_block_table_entry_for_an_angular_mode():
    """The HIP bit's table entry for a non-DC mode (per-block call through the staging arena) equals the oracle."""
    L = capi.lib()
    table = (C.c_void_p * (4 * 35 + 3))()
    L.hevcasm_populate_pred_intra_all_modes(table, capi.HIP)
    fn = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.c_int)
    nb = np.ascontiguousarray(wl.fill_u8(144, 5))
    for log2, mode in ((2, 0), (3, 7), (4, 26), (5, 19)):
        packed = (8 << 8) | (log2 << 1) | (1 if log2 < 5 else 0)
        out = np.zeros(1 << (2 * log2), dtype=np.uint8)
        fn(table[(log2 - 2) * 35 + mode])(out.ctypes.data, nb.ctypes.data, mode, packed)
        assert np.array_equal(out, pyoracle.pred_intra_modes_batch(nb, 144, np.array([mode], dtype=np.uint8), packed)), (log2, mode)


@pytest.mark.gpu
def test_hip_intra_full_size_against_the_definition():
    """2^24 output pixels per size with per-block modes: first and last 300 blocks against the oracle; every block of modes 26 / 10 / 34 / 2 against the
    replication / shift the text implies, computed by torch on the device."""
    import torch
    from hevcasm_amd import batch as B
    for log2 in (2, 3, 4, 5):
        size = 1 << log2
        n = (1 << 24) >> (2 * log2)
        packed = (8 << 8) | (log2 << 1)
        nb = wl.fill_u8(144 * n + 64, 40 + log2, device="cuda")
        modes_h = np.random.default_rng(log2).integers(0, 35, n).astype(np.uint8)
        modes = B.u8(modes_h)
        dst = torch.zeros(n * size * size, dtype=torch.uint8, device="cuda")
        B.pred_intra_modes(dst, nb, 144, modes, packed)
        torch.cuda.synchronize()
        got = dst.view(n, size, size)
        rec = nb[: 144 * n].view(n, 144)
        top, left = rec[:, 64:64 + 2 * size], torch.flip(rec[:, 0:64], dims=[1])[:, :2 * size]
        m = modes.to(torch.int64)
        assert torch.equal(got[m == 26], top[m == 26][:, None, :size].expand(-1, size, -1))
        assert torch.equal(got[m == 10], left[m == 10][:, :size, None].expand(-1, -1, size))
        idx = (torch.arange(size, device="cuda")[:, None] + torch.arange(size, device="cuda")[None, :] + 1).reshape(-1)
        assert torch.equal(got[m == 34].reshape(-1, size * size), top[m == 34][:, idx]) and torch.equal(got[m == 2].reshape(-1, size * size), left[m == 2][:, idx])
        host_nb, host = nb.cpu().numpy(), dst.cpu().numpy()
        for sl in (slice(0, 300), slice(n - 300, n)):
            want = pyoracle.pred_intra_modes_batch(host_nb[144 * sl.start: 144 * sl.stop + 64], 144, modes_h[sl], packed)
            assert np.array_equal(host[sl.start * size * size: sl.stop * size * size], want), (log2, sl)
        del nb, dst, modes
        torch.cuda.empty_cache()


def _compact(nb, stride, log2, n):
    """The 4N + 1 samples of n reference-layout records, as compact records 4N + 1 bytes apart (+ 3 of padding)."""
    size = 1 << log2
    cs = 4 * size + 1 + 3
    out = np.zeros(cs * n + 8, dtype=np.uint8)
    rec = np.asarray(nb)[: stride * (n - 1) + 129]
    for i in range(n):
        out[i * cs: i * cs + 4 * size] = nb[i * stride + 64 - 2 * size: i * stride + 64 + 2 * size]
        out[i * cs + 4 * size] = nb[i * stride + 128]
    return out, cs


@pytest.mark.gpu
def test_hip_compact_records_equal_reference_layout_records():
    """HEVCASM_HIP_INTRA_COMPACT (a batch-API extension: the corner directly behind the top row, records of 4N + 1 bytes): prediction with a mode per block, one
    mode per batch (DC included) and the neighbour smoothing give what the same samples give in the reference's 129-byte layout."""
    import torch
    from hevcasm_amd import batch as B
    rng = np.random.default_rng(5)
    for log2, edge, packed in _packings():
        size, n = 1 << log2, 333
        nb = wl.fill_u8(144 * n + 64, 70 + log2)
        modes = rng.integers(0, 35, n).astype(np.uint8)
        comp, cs = _compact(nb, 144, log2, n)
        want = pyoracle.pred_intra_modes_batch(nb, 144, modes, packed)
        dst = torch.zeros(n * size * size, dtype=torch.uint8, device="cuda")
        B.pred_intra_modes(dst, B.u8(comp), cs, B.u8(modes), packed | B.INTRA_COMPACT)
        torch.cuda.synchronize()
        assert np.array_equal(dst.cpu().numpy(), want), (log2, edge)
        for mode in (1, 0, 26, 13):
            B.pred_intra(dst, B.u8(comp), cs, mode, packed | B.INTRA_COMPACT, n)
            torch.cuda.synchronize()
            assert np.array_equal(dst.cpu().numpy(), pyoracle.pred_intra_modes_batch(nb, 144, np.full(n, mode, dtype=np.uint8), packed)), (log2, edge, mode)
        f = B.intra_filter_neighbours(B.u8(comp), cs, log2 | B.INTRA_COMPACT, 1, n, out_stride=cs)
        torch.cuda.synchronize()
        got, _ = f.cpu().numpy(), None
        ref = pyoracle.intra_filter_neighbours_batch(nb, 144, log2, 1, n, out_stride=144)
        want_c, _ = _compact(ref, 144, log2, n)
        assert np.array_equal(got[: cs * n].reshape(n, cs)[:, : 4 * size + 1], want_c[: cs * n].reshape(n, cs)[:, : 4 * size + 1]), log2
