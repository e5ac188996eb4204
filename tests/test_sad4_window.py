"""4-way SAD of wide blocks: the union-window kernel (candidates close together: one pass of the window through LDS) and the
rescan pass that computes every other block with the generic code (sad.hip).  Candidate geometries on both sides of the
qualification limits, mixed batches, candidates flush with the ends of an exactly-sized plane, misaligned plane bases,
strides at the limits of the window path, all against the oracle."""
import numpy as np
import pytest

import cases
from hevcasm_amd import workload as wl

pytestmark = pytest.mark.gpu

WIDE = [(64, 64), (64, 48), (64, 32), (64, 16), (48, 64), (32, 64), (32, 32), (32, 16), (48, 48), (48, 16), (16, 64), (16, 32), (16, 16), (32, 48), (16, 48)]


def _case(w, h, n, seed, disp, stride_pad=64):
    """n blocks; candidate k of block i = centre_i + disp(i, k) where disp is a callable -> (n, 4, 2) array of (dx, dy)."""
    rng = np.random.default_rng(seed)
    d = disp(rng, n)
    reach_x, reach_y = int(np.abs(d[:, :, 0]).max()), int(np.abs(d[:, :, 1]).max())
    stride = w + 2 * reach_x + stride_pad      # the window path needs stride % 4 == 0: true here for even pads; odd pads take the generic path
    rows = h + 2 * reach_y
    plane = rows * stride
    centre = np.arange(n, dtype=np.int64) * plane + reach_y * stride + reach_x
    ref_off = centre[:, None] + d[:, :, 1] * stride + d[:, :, 0]
    src = wl.fill_u8(n * w * h, seed + 1)
    return dict(src=src, stride_src=w, src_off=(np.arange(n) * w * h).astype(np.uint32),
                ref=wl.fill_u8(n * plane, seed + 2), stride_ref=stride, ref_off=ref_off.reshape(-1).astype(np.uint32), w=w, h=h, n=n, ways=4)


def _uniform(r):
    return lambda rng, n: rng.integers(-r, r + 1, size=(n, 4, 2))


def _corners(sx, sy):
    """the four candidates at the corners of an sx x sy box (the widest spread a span allows), order shuffled per block"""
    def f(rng, n):
        box = np.array([[0, 0], [sx, 0], [0, sy], [sx, sy]])
        return np.stack([box[rng.permutation(4)] for _ in range(n)]) - np.array([sx // 2, sy // 2])
    return f


@pytest.mark.parametrize("w,h", WIDE)
def test_close_far_and_mixed(w, h):
    n = 70
    for name, disp in (("close7", _uniform(7)), ("close8", _uniform(8)), ("far40", _uniform(40)), ("identical", _uniform(0)),
                       ("box16x16", _corners(16, 16)), ("box17x16", _corners(17, 16)), ("box16x17", _corners(16, 17)),
                       ("box16x0", _corners(16, 0)), ("box0x16", _corners(0, 16)), ("box3x1", _corners(3, 1))):
        c = _case(w, h, n, 5000 + w + h, disp)
        cases.assert_same("sad4 %dx%d %s" % (w, h, name), cases.run_hip("sad4", c), cases.run_oracle("sad4", c))

    def mixed(rng, n):
        d = rng.integers(-7, 8, size=(n, 4, 2))
        far = rng.random(n) < 0.3
        d[far] = rng.integers(-60, 61, size=(int(far.sum()), 4, 2))
        return d
    c = _case(w, h, 333, 5100 + w + h, mixed)
    cases.assert_same("sad4 %dx%d mixed" % (w, h), cases.run_hip("sad4", c), cases.run_oracle("sad4", c))
    # one candidate a whole row below and far to the LEFT of the others: a small offset difference that is not a small displacement
    def wrapped(rng, n):
        d = rng.integers(-3, 4, size=(n, 4, 2))
        d[:, 1, 0] -= 40
        d[:, 1, 1] += 1
        return d
    c = _case(w, h, 50, 5200 + w + h, wrapped)
    cases.assert_same("sad4 %dx%d row-wrapped" % (w, h), cases.run_hip("sad4", c), cases.run_oracle("sad4", c))


@pytest.mark.parametrize("w,h", [(64, 64), (48, 64), (64, 16), (32, 32), (32, 64), (16, 16), (16, 64)])
def test_candidates_flush_with_the_plane_ends_and_tight_strides(w, h):
    """Planes that hold exactly the candidates' bounding box: the first block's window starts on byte 0 of the allocation,
    the last one's ends on its last byte; strides from the smallest the window path accepts downwards (generic path)."""
    for pad in (64, 63, 62, 0, 200):
        for (sx, sy) in ((16, 16), (5, 9), (0, 0), (16, 0)):
            def box(rng, n, sx=sx, sy=sy):
                b = np.array([[0, 0], [sx, 0], [0, sy], [sx, sy]]) - np.array([sx // 2, sy // 2])
                return np.stack([b[rng.permutation(4)] for _ in range(n)])
            c = _case(w, h, 9, 5300 + pad + sx, box, stride_pad=pad)
            # make the stride exactly (bounding box width + pad) and the plane exactly n bounding boxes
            bw, bh = w + sx, h + sy
            stride = bw + pad
            n = c["n"]
            plane = (bh - 1) * stride + bw            # the last row is cut right after the box
            rng = np.random.default_rng(42 + pad + sx + sy)
            origin = np.arange(n, dtype=np.int64) * (bh * stride)
            corners = np.array([[0, 0], [sx, 0], [0, sy], [sx, sy]])
            ro = np.stack([origin[i] + (corners[rng.permutation(4)] * np.array([1, stride])).sum(axis=1) for i in range(n)])
            c["stride_ref"] = stride
            c["ref"] = wl.fill_u8((n - 1) * bh * stride + plane, 77 + pad)
            c["ref_off"] = ro.reshape(-1).astype(np.uint32)
            cases.assert_same("sad4 %dx%d tight pad %d box %dx%d" % (w, h, pad, sx, sy), cases.run_hip("sad4", c), cases.run_oracle("sad4", c))


def test_misaligned_base_and_large_batch_equals_four_single_launches():
    import torch
    from hevcasm_amd import batch as B
    for (w, h) in ((64, 64), (48, 64), (64, 32), (32, 32), (16, 32)):
        c = wl.sad_case(w, h, 4000, 900 + w + h, ways=4)
        so, ro4 = B.off32(c["src_off"]), B.off32(c["ref_off"])
        want = np.stack([B.sad(B.u8(c["src"]), c["stride_src"], B.u8(c["ref"]), c["stride_ref"], so, B.off32(c["ref_off"].reshape(-1, 4)[:, k].copy()), w, h).cpu().numpy()
                         for k in range(4)], axis=1)
        for mis in (0, 1, 2, 3):
            big = torch.zeros(c["ref"].size + 8, dtype=torch.uint8, device="cuda")
            big[mis:mis + c["ref"].size] = torch.from_numpy(c["ref"]).cuda()
            got = B.sad_multiref(B.u8(c["src"]), c["stride_src"], big[mis:], c["stride_ref"], so, ro4, w, h).cpu().numpy().reshape(-1, 4)
            assert np.array_equal(got, want), (w, h, mis)
