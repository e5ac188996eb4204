"""Memory-footprint probe for the per-block prediction entries (run as a subprocess by test_footprint.py).

The reference reads the filter margins of a prediction block only in a direction whose fraction is non-zero
(reference pred_inter.c:141-228): W x H bytes for an integer motion vector, (W + T - 1) x H horizontal-only,
W x (H + T - 1) vertical-only.  A caller may therefore hand a drop-in replacement a block whose window ends on the
last byte of an allocation, or starts on its first.  This probe builds exactly that: the window the reference reads,
row stride = window width, flush against an inaccessible page (PROT_NONE) after it and, in a second placement, before
it.  Any byte read outside the reference's footprint is a segmentation fault, which the parent test sees as a
non-zero exit status.  Results are compared with the oracle run on a copy of the window embedded in a large plane.

usage: footprint_probe.py c_ref|hip
"""
import ctypes as C
import mmap
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from hevcasm_amd import capi  # noqa: E402
from oracle import pyoracle  # noqa: E402

PAGE = mmap.PAGESIZE
libc = C.CDLL(None, use_errno=True)
libc.mprotect.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
libc.mprotect.restype = C.c_int
PD, VP = C.c_ssize_t, C.c_void_p


class Guarded:
    """`pages` accessible pages between two inaccessible ones."""

    def __init__(self, pages):
        self.map = mmap.mmap(-1, (pages + 2) * PAGE)
        self.buf = np.frombuffer(self.map, dtype=np.uint8)
        self.base = self.buf.ctypes.data
        assert self.base % PAGE == 0
        for at in (self.base, self.base + (pages + 1) * PAGE):
            if libc.mprotect(at, PAGE, 0) != 0:
                raise OSError(C.get_errno(), "mprotect")
        self.first = PAGE                   # offset of the first accessible byte
        self.end = (pages + 1) * PAGE       # offset one past the last accessible byte


def table_slot(populate, nslots, index, mask):
    buf = (C.c_void_p * nslots)()
    getattr(capi.lib(), populate)(buf, mask)
    assert buf[index], "NULL table entry"
    return buf[index]


def margins(taps, fx, fy):
    before, after = taps // 2 - 1, taps // 2
    return (before if fx else 0, after if fx else 0, before if fy else 0, after if fy else 0)


def place(g, rng, taps, w, h, fx, fy, at_end):
    """Random window of exactly the reference's footprint, flush at the end (or the start) of the accessible
    range; returns (offset of the block's top-left sample, row stride, the window as a 2-D array)."""
    bx, ax, by, ay = margins(taps, fx, fy)
    ww, wh = w + bx + ax, h + by + ay
    win = rng.integers(0, 256, size=(wh, ww), dtype=np.uint8)
    start = g.end - ww * wh if at_end else g.first
    g.buf[start:start + ww * wh] = win.reshape(-1)
    return start + by * ww + bx, ww, win


def expect(taps, w, h, wins, fracs):
    """Oracle on windows embedded in a large random plane (stride 256, 16 samples of random margin all round):
    with the filter margins present but arbitrary, a correct result cannot depend on what lies outside the footprint."""
    stride, origin = 256, 16 * 256 + 16
    rng = np.random.default_rng(7)
    planes, offs = [], []
    for (fx, fy), win in zip(fracs, wins):
        bx, ax, by, ay = margins(taps, fx, fy)
        p = rng.integers(0, 256, size=(16 + 64 + 16 + 16) * stride, dtype=np.uint8)
        for y in range(win.shape[0]):
            at = origin + (y - by) * stride - bx
            p[at:at + win.shape[1]] = win[y]
        planes.append(p)
        offs.append(origin)
    frac = [np.array([fx | (fy << 4)], dtype=np.uint8) for fx, fy in fracs]
    off0 = np.array([0], dtype=np.uint32)
    if len(wins) == 1:
        d = pyoracle.pred_uni_batch(taps, 64 * 64, 64, planes[0], stride, off0, np.array([offs[0]], dtype=np.uint32), frac[0], w, h)
    else:
        d = pyoracle.pred_bi_batch(taps, 64 * 64, 64, planes[0], planes[1], stride, off0, np.array([offs[0]], dtype=np.uint32),
                                   np.array([offs[1]], dtype=np.uint32), frac[0], frac[1], w, h)
    return d.reshape(64, 64)[:h, :w]


def main():
    mask = {"c_ref": capi.C_REF, "hip": capi.HIP}[sys.argv[1]]
    rng = np.random.default_rng(11)
    uni_t = C.CFUNCTYPE(None, VP, PD, VP, PD, C.c_int, C.c_int, C.c_int, C.c_int)
    bi_t = C.CFUNCTYPE(None, VP, PD, VP, VP, PD, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)
    checked = 0
    for taps, sizes in ((8, [(16, 16), (8, 4), (4, 8), (64, 64), (12, 16)]), (4, [(8, 8), (2, 4), (4, 2), (32, 32), (6, 8)])):
        hi = 3 if taps == 8 else 7
        for (w, h) in sizes:
            for (fx, fy) in ((0, 0), (hi, 0), (0, hi), (2, 1)):
                for at_end in (True, False):
                    # uni-prediction
                    g = Guarded(3)
                    off, stride, win = place(g, rng, taps, w, h, fx, fy, at_end)
                    idx = (((taps // 4 - 1) * 9 + (w + taps - 1) // taps) * 2 + (1 if fx else 0)) * 2 + (1 if fy else 0)
                    f = uni_t(table_slot("hevcasm_populate_pred_uni_8to8", 72, idx, mask))
                    dst = np.zeros((h, w), dtype=np.uint8)
                    f(dst.ctypes.data, w, g.base + off, stride, w, h, fx, fy)
                    want = expect(taps, w, h, [win], [(fx, fy)])
                    assert np.array_equal(dst, want), ("uni", taps, w, h, fx, fy, at_end)
                    checked += 1
            # bi-prediction: the two references have their own fractions, hence their own footprints; both windows get the same
            # stride (the function takes one stride_ref), so each is laid out at the wider window's width
            if w < 4:
                continue
            for f0, f1 in (((0, 0), (0, 0)), ((0, 0), (hi, 1)), ((hi, 0), (0, 2)), ((1, 2), (0, 0))):
                for at_end in (True, False):
                    stride = w + (taps - 1 if (f0[0] or f1[0]) else 0)
                    gs, offs, wins = [], [], []
                    for (fx, fy) in (f0, f1):
                        bx, ax, by, ay = margins(taps, fx, fy)
                        wh = h + by + ay
                        win = rng.integers(0, 256, size=(wh, w + bx + ax), dtype=np.uint8)
                        # rows `stride` apart; the first row starts (or the last row ends) flush with the accessible range
                        g = Guarded(3)
                        span = (wh - 1) * stride + win.shape[1]
                        start = g.end - span if at_end else g.first
                        for y in range(wh):
                            g.buf[start + y * stride:start + y * stride + win.shape[1]] = win[y]
                        gs.append(g); wins.append(win); offs.append(start + by * stride + bx)
                    idx = ((taps // 4 - 1) * 5 + (w + 2 * taps - 1) // (2 * taps)) * 2 + (1 if (f0[0] or f0[1] or f1[0] or f1[1]) else 0)
                    f = bi_t(table_slot("hevcasm_populate_pred_bi_8to8", 20, idx, mask))
                    dst = np.zeros((h, w), dtype=np.uint8)
                    f(dst.ctypes.data, w, gs[0].base + offs[0], gs[1].base + offs[1], stride, w, h, f0[0], f0[1], f1[0], f1[1])
                    want = expect(taps, w, h, wins, [f0, f1])
                    assert np.array_equal(dst, want), ("bi", taps, w, h, f0, f1, at_end)
                    checked += 1
    print("footprint probe OK: %d calls flush against inaccessible pages (%s)" % (checked, sys.argv[1]))


if __name__ == "__main__":
    main()
