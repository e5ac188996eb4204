"""What the prediction entry points READ: the reference's footprint (reference pred_inter.c:141-228) -- the filter
margins only in a direction whose fraction is non-zero; byte-exact but for the 4-tap strip kernel of narrow PUs, which loads
whole aligned words between the lowest and the highest word of one wavefront's windows (include/hevcasm_hip.h).  The per-block table entries are probed with blocks flush
against inaccessible pages (tests/footprint_probe.py, in a subprocess: an over-read is a segmentation fault); the batch
kernels with blocks flush at both ends of exactly-sized device allocations."""
import os
import subprocess
import sys

import numpy as np
import pytest

import cases
from hevcasm_amd import workload as wl

HERE = os.path.dirname(os.path.abspath(__file__))


def _probe(which):
    r = subprocess.run([sys.executable, os.path.join(HERE, "footprint_probe.py"), which], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "footprint probe (%s) died with status %d\n%s\n%s" % (which, r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert "footprint probe OK" in r.stdout


def test_c_path_reads_only_the_reference_footprint(oracle_lib):
    _probe("c_ref")


@pytest.mark.gpu
def test_hip_per_block_entries_read_only_the_reference_footprint(oracle_lib):
    _probe("hip")


def _edge_case(taps, w, h, fracs, bi, n):
    """n PUs whose reference footprints tile an exactly-sized plane: the first PU's footprint starts on byte 0 of the
    allocation, the last one's ends on its last byte.  Row stride = footprint width, PUs stacked vertically."""
    before, after = taps // 2 - 1, taps // 2
    c = dict(taps=taps, w=w, h=h, n=n, stride_dst=w, dst_bytes=w * h * n, dst_off=(np.arange(n) * w * h).astype(np.uint32))
    any_fx = any(fx for fx, _ in fracs)
    stride = w + (before + after if any_fx else 0)
    c["stride_ref"] = stride
    for r, (fx, fy) in enumerate(fracs):
        bx, ax = (before, after) if fx else (0, 0)
        by, ay = (before, after) if fy else (0, 0)
        wh = h + by + ay
        # PU i's window occupies rows [i * wh, (i + 1) * wh) of a plane that is exactly n * wh rows of `stride` bytes, minus
        # the unused tail of the very last row when this reference's window is narrower than the stride
        size = (n * wh - 1) * stride + (w + bx + ax)
        c["ref%d" % r] = wl.fill_u8(size, 4000 + 17 * r + w + h + fx * 8 + fy, "noise", None)
        c["ref%d_off" % r] = (np.arange(n) * wh * stride + by * stride + bx).astype(np.uint32)
        c["frac%d" % r] = np.full(n, fx | (fy << 4), dtype=np.uint8)
    return c


def _padded(c, nrefs, pad=8 * 4096):
    """The same case with `pad` bytes of noise before and after every reference plane: what the oracle (which, like the
    reference's C bi-prediction, walks the full window and multiplies the margins by zero) may read freely."""
    p = dict(c)
    for r in range(nrefs):
        noise = wl.fill_u8(2 * pad, 99 + r, "noise", None)
        p["ref%d" % r] = np.concatenate([noise[:pad], c["ref%d" % r], noise[pad:]])
        p["ref%d_off" % r] = (c["ref%d_off" % r].astype(np.int64) + pad).astype(np.uint32)
    return p


@pytest.mark.gpu
def test_batch_kernels_at_the_ends_of_exact_allocations(oracle_lib):
    for taps, sizes in ((8, [(16, 16), (8, 4), (4, 8), (64, 64), (12, 16), (24, 32), (5, 3), (3, 7), (1, 2)]),
                        (4, [(8, 8), (2, 4), (4, 2), (32, 32), (6, 8), (2, 2), (3, 5), (1, 1), (12, 16)])):
        hi = 3 if taps == 8 else 7
        for (w, h) in sizes:
            for frac in ((0, 0), (hi, 0), (0, hi), (1, 2)):
                c = _edge_case(taps, w, h, [frac], False, 5)
                cases.assert_same("uni%d_%dx%d_f%d%d_edges" % (taps, w, h, frac[0], frac[1]),
                                  cases.run_hip("pred_uni", c), cases.run_oracle("pred_uni", _padded(c, 1)))
            for f0, f1 in (((0, 0), (0, 0)), ((0, 0), (hi, 1)), ((hi, 0), (0, 2)), ((1, 2), (0, 0)), ((0, 1), (0, hi))):
                c = _edge_case(taps, w, h, [f0, f1], True, 5)
                cases.assert_same("bi%d_%dx%d_%s%s_edges" % (taps, w, h, f0, f1),
                                  cases.run_hip("pred_bi", c), cases.run_oracle("pred_bi", _padded(c, 2)))
