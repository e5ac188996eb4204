"""Seeded synthetic workloads for the hot path, shared by tests and bench.py.

Data model (what include/hevcasm_hip.h consumes): 8-bit planes stay in frame layout -- a stack
of frames, each ``frame_h`` rows of ``stride`` bytes -- and a block is a u32 byte offset from
the plane's base.  Block ``i`` of a batch is tile ``i`` of the stack in raster order, so
consecutive blocks are horizontal neighbours (what a codec walking a frame produces).

Geometry and offsets are always computed with numpy (cheap, deterministic).  Pixel payloads
come from ``fill_u8`` / ``fill_i16``, either numpy (tests: same bytes go to the oracle and the
GPU) or torch on the device (bench: tens of GB are generated in HBM, never on the host).
"""
import numpy as np

# reference sad.c:231-240
SAD_PARTITIONS = [(64, 64), (64, 48), (64, 32), (64, 16), (48, 64), (32, 64), (32, 32), (32, 24), (32, 16), (32, 8),
                  (24, 32), (16, 64), (16, 32), (16, 16), (16, 12), (16, 8), (16, 4), (12, 16),
                  (8, 32), (8, 16), (8, 8), (8, 4), (4, 8)]
# BASELINE.json config #2: both dimensions >= 8
SAD_BENCH_SIZES = [p for p in SAD_PARTITIONS if p[0] >= 8 and p[1] >= 8]
# reference pred_inter.c:438-444 (luma; chroma = half)
PU_SIZES = [(8, 4), (8, 8), (4, 8), (16, 4), (16, 8), (16, 12), (16, 16), (12, 16), (8, 16), (4, 16),
            (32, 8), (32, 16), (32, 24), (32, 32), (24, 32), (16, 32), (8, 32),
            (64, 16), (64, 32), (64, 48), (64, 64), (48, 64), (32, 64), (16, 64)]
PU_SIZES_BI = [p for p in PU_SIZES if p not in ((8, 4), (4, 8))]
TRANSFORMS = [(1, 2), (0, 2), (0, 3), (0, 4), (0, 5)]  # (trType, log2TrafoSize)

# A frame every hot-path block size tiles exactly: 1920 = lcm-friendly width, 1152 = 6 * 192 rows.
FRAME_W, FRAME_H = 1920, 1152
# Small frame with the same property, for CPU-sized tests.
SMALL_W, SMALL_H = 192, 192


class Plane:
    """Geometry of a frame stack: ``frames`` frames of (frame_h + 2*border_y) rows of ``stride`` bytes."""

    def __init__(self, frame_w, frame_h, frames, border_x=0, border_y=0):
        self.frame_w, self.frame_h, self.frames = frame_w, frame_h, frames
        self.border_x, self.border_y = border_x, border_y
        self.stride = frame_w + 2 * border_x
        self.rows_per_frame = frame_h + 2 * border_y
        self.size = self.stride * self.rows_per_frame * frames

    def tile_offsets(self, n, w, h):
        """Offsets of the first n w x h tiles in raster order (frame after frame)."""
        per_row, per_col = self.frame_w // w, self.frame_h // h
        per_frame = per_row * per_col
        i = np.arange(n, dtype=np.int64)
        f, r = i // per_frame, i % per_frame
        by, bx = r // per_row, r % per_row
        off = (f * self.rows_per_frame + self.border_y + by * h) * self.stride + self.border_x + bx * w
        assert off.max(initial=0) < 2 ** 32, "plane exceeds the 4 GiB offset window: split the batch"
        return off

    @staticmethod
    def frames_for(n, w, h, frame_w, frame_h):
        per_frame = (frame_w // w) * (frame_h // h)
        return max(1, -(-n // per_frame))


def fill_u8(size, seed, kind="noise", device=None):
    """``size`` bytes.  kind: 'noise' (uniform 0..255), 'zeros', 'max' (all 255), 'video' (a smooth gradient plus noise of
    sigma ~8, the second variant SURVEY.md 8(d) asks for config #2: neighbouring samples correlate as in real frames)."""
    if device is None:
        if kind == "zeros":
            return np.zeros(size, dtype=np.uint8)
        if kind == "max":
            return np.full(size, 255, dtype=np.uint8)
        if kind == "video":
            i = np.arange(size, dtype=np.int64)
            base = 128 + 90 * np.sin(i / 97.0) * np.cos(i / 4099.0)
            noise = np.random.default_rng(seed).normal(0.0, 8.0, size)
            return np.clip(np.rint(base + noise), 0, 255).astype(np.uint8)
        return np.random.default_rng(seed).integers(0, 256, size=size, dtype=np.uint8)
    import torch
    if kind == "zeros":
        return torch.zeros(size, dtype=torch.uint8, device=device)
    if kind == "max":
        return torch.full((size,), 255, dtype=torch.uint8, device=device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty(size, dtype=torch.uint8, device=device)
    step = 1 << 30  # bounded temporaries
    for s in range(0, size, step):
        e = min(size, s + step)
        out[s:e] = torch.randint(0, 256, (e - s,), generator=g, device=device, dtype=torch.uint8)
    return out


def fill_i16(size, seed, kind="full", device=None):
    """``size`` int16 values.  kind: 'full' (uniform over all 16 bits, as reference
    residual_decode.c:574), 'residual9' (9-bit, reference :1000), 'natural' (sparse Laplacian-like
    levels), 'neg256' (reference quantize.c:145), 'min' / 'max' (all -32768 / +32767)."""
    rng = np.random.default_rng(seed)
    if kind == "full":
        a = rng.integers(-32768, 32768, size=size, dtype=np.int32)
    elif kind == "residual9":
        a = rng.integers(-256, 256, size=size, dtype=np.int32)
    elif kind == "neg256":
        a = rng.integers(-256, 0, size=size, dtype=np.int32)
    elif kind == "natural":
        mag = rng.geometric(0.25, size=size).astype(np.int32)
        a = np.where(rng.random(size) < 0.7, 0, mag * rng.choice(np.array([-1, 1], dtype=np.int32), size=size))
    elif kind == "min":
        a = np.full(size, -32768, dtype=np.int32)
    elif kind == "max":
        a = np.full(size, 32767, dtype=np.int32)
    elif kind == "alternating":
        a = np.where(np.arange(size) & 1, -32767, 32767).astype(np.int32)
    else:
        raise ValueError(kind)
    a = a.astype(np.int16)
    if device is None:
        return a
    import torch
    return torch.from_numpy(a).to(device)


def fill_i16_device(size, seed, device, lo=-32768, hi=32768):
    """Uniform int16 generated on the device (bench-sized buffers)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return torch.randint(lo, hi, (size,), generator=g, device=device, dtype=torch.int16)


# ---------------------------------------------------------------------------------------------
# cases: dicts of numpy arrays + scalars; keys match the argument names of the batch API


def sad_case(w, h, n, seed, ways=1, max_disp=7, frame=(SMALL_W, SMALL_H), kind="noise", device=None):
    """Config #2: src block i = tile i; ref block i = same tile in a second stack displaced by
    (dx, dy) uniform in [-max_disp, max_disp]^2 (so byte-misaligned, as reference sad.c:214)."""
    fw, fh = frame
    frames = Plane.frames_for(n, w, h, fw, fh)
    sp = Plane(fw, fh, frames)
    rp = Plane(fw, fh, frames, border_x=16, border_y=8)
    assert max_disp <= 8
    rng = np.random.default_rng(seed + 1)
    src_off = sp.tile_offsets(n, w, h)
    base = rp.tile_offsets(n, w, h)
    d = rng.integers(-max_disp, max_disp + 1, size=(n, ways, 2))
    ref_off = base[:, None] + d[:, :, 1] * rp.stride + d[:, :, 0]
    return dict(
        src=fill_u8(sp.size, seed, kind, device), stride_src=sp.stride,
        ref=fill_u8(rp.size, seed + 2, kind if kind != "max" else "zeros", device), stride_ref=rp.stride,
        src_off=src_off.astype(np.uint32), ref_off=ref_off.reshape(-1).astype(np.uint32),
        w=w, h=h, n=n, ways=ways)


def pred_case(taps, w, h, n, seed, bi=False, frame=(SMALL_W, SMALL_H), kind="noise", fracs=None, device=None, coherent_mv=False, mv_region=0):
    """Config #3: dst tiled by w x h PUs; reference origin = same position displaced by an integer
    motion vector in [-8, 8]^2; fractions uniform over all phases (4 luma / 8 chroma per axis)
    unless ``fracs`` = (xFrac, yFrac) pins them.  ``mv_region`` = R > 0: one vector and fraction pair per R x R pixel region
    of a frame (all PUs of a CTU move together: measurement only) instead of one per PU."""
    fw, fh = frame
    frames = Plane.frames_for(n, w, h, fw, fh)
    dp = Plane(fw, fh, frames)
    rp = Plane(fw, fh, frames, border_x=16, border_y=16)
    rng = np.random.default_rng(seed + 1)
    dst_off = dp.tile_offsets(n, w, h)
    base = rp.tile_offsets(n, w, h)
    phases = 4 if taps == 8 else 8
    out = dict(taps=taps, w=w, h=h, n=n, stride_dst=dp.stride, stride_ref=rp.stride, dst_bytes=dp.size,
               dst_off=dst_off.astype(np.uint32))
    for r in range(2 if bi else 1):
        mv = rng.integers(-8, 9, size=(n, 2))
        if coherent_mv:               # one motion vector for the whole batch (exploration only)
            mv[:] = mv[0]
        if fracs is None:
            fx, fy = rng.integers(0, phases, size=n), rng.integers(0, phases, size=n)
        else:
            fx, fy = np.full(n, fracs[0]), np.full(n, fracs[1])
        if mv_region:
            # the region a PU's top-left pixel lies in, per frame
            per_row, per_col = fw // w, fh // h
            i = np.arange(n, dtype=np.int64)
            f, rr = i // (per_row * per_col), i % (per_row * per_col)
            ry, rx = (rr // per_row) * h // mv_region, (rr % per_row) * w // mv_region
            regions_x, regions_y = -(-fw // mv_region), -(-fh // mv_region)
            region = (f * regions_y + ry) * regions_x + rx
            n_regions = int(region.max()) + 1
            mv = rng.integers(-8, 9, size=(n_regions, 2))[region]
            if fracs is None:
                fx, fy = rng.integers(0, phases, size=n_regions)[region], rng.integers(0, phases, size=n_regions)[region]
        off = base + mv[:, 1] * rp.stride + mv[:, 0]
        out["ref%d" % r] = fill_u8(rp.size, seed + 2 + r, kind, device)
        out["ref%d_off" % r] = off.astype(np.uint32)
        out["frac%d" % r] = (fx | (fy << 4)).astype(np.uint8)
    return out


def inverse_case(tr_type, log2n, n, seed, coeff_kind="full", frame=(SMALL_W, SMALL_H), device=None):
    """Config #4: n coefficient blocks (contiguous), prediction and destination planes tiled by N x N."""
    size = 1 << log2n
    fw, fh = frame
    frames = Plane.frames_for(n, size, size, fw, fh)
    pp = Plane(fw, fh, frames)
    dp = Plane(fw, fh, frames, border_x=16)  # a different stride from the prediction plane
    return dict(
        tr_type=tr_type, log2n=log2n, n=n,
        pred=fill_u8(pp.size, seed, "noise", device), stride_pred=pp.stride, pred_off=pp.tile_offsets(n, size, size).astype(np.uint32),
        stride_dst=dp.stride, dst_bytes=dp.size, dst_off=dp.tile_offsets(n, size, size).astype(np.uint32),
        coeffs=fill_i16(n * size * size, seed + 1, coeff_kind, device))


def forward_case(tr_type, log2n, n, seed, kind="residual9", frame=(SMALL_W, SMALL_H), device=None):
    """Residual plane (int16, frame layout) tiled by N x N; offsets in int16 elements."""
    size = 1 << log2n
    fw, fh = frame
    frames = Plane.frames_for(n, size, size, fw, fh)
    p = Plane(fw, fh, frames)
    return dict(tr_type=tr_type, log2n=log2n, n=n, src=fill_i16(p.size, seed, kind, device), src_stride=p.stride,
                src_off=p.tile_offsets(n, size, size).astype(np.uint32))


# ---------------------------------------------------------------------------------------------
# intra frames: a TU map in decoding order + quantiser parameters, for hevcasm_hip_intra_reconstruct_frames

INTRA_TU_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("log2", "u1"), ("mode", "u1"), ("reserved", "<u2")])      # = hevcasm_hip_intra_tu, 8 bytes


def frame_2d(w, h, stride, seed, kind="video"):
    """An h x stride plane (bytes beyond w in a row are zero).  'video': smooth two-dimensional shading + a few straight edges + noise of sigma ~4 --
    directional structure, so that intra modes and block sizes matter; 'noise': uniform bytes; 'flat': 128; 'max' / 'zeros'."""
    out = np.zeros((h, stride), dtype=np.uint8)
    rng = np.random.default_rng(seed)
    if kind == "noise":
        out[:, :w] = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    elif kind == "flat":
        out[:, :w] = 128
    elif kind == "max":
        out[:, :w] = 255
    elif kind == "zeros":
        pass
    else:
        y, x = np.mgrid[0:h, 0:w].astype(np.float64)
        img = 128 + 60 * np.sin(x / 37.0 + y / 91.0) + 40 * np.cos(y / 23.0 - x / 150.0)
        for k in range(6):      # straight edges at assorted angles
            a, b, c = rng.normal(size=3)
            img += 35 * np.tanh((a * (x - w / 2) + b * (y - h / 2) + 40 * c) / 2.0) * (1 if k & 1 else -1)
        img += rng.normal(0.0, 4.0, size=(h, w))
        out[:, :w] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    return out.reshape(-1)


def intra_tu_map(w, h, ctu, seed, modes="mixed", split=(1.0, 0.7, 0.5, 0.35), min_log2=2, max_log2=5):
    """A random quad-tree partition of a w x h plane (multiples of 4) into TUs of 4x4 ... 32x32, CTU after CTU in raster order and in z-scan order inside a
    CTU -- the decoding order the availability rules of H.265 6.4.1 / 8.4.4.2.2 presume.  split[d]: probability that a square at depth d (0 = the CTU) is
    split further (squares larger than 32, or crossing the picture boundary, always are).  modes: 'dc' (all INTRA_DC: the pinned subset), 'mixed' (uniform
    0..34), 'planar', 'angular' (2..34), or an int.  Returns (tus: INTRA_TU_DTYPE array, ctu_first: uint32 [n_ctus + 1])."""
    assert w % 4 == 0 and h % 4 == 0 and ctu in (16, 32, 64)
    rng = np.random.default_rng(seed)
    tus, first = [], [0]

    def mode():
        if isinstance(modes, int):
            return modes
        if modes == "dc":
            return 1
        if modes == "planar":
            return 0
        if modes == "angular":
            return int(rng.integers(2, 35))
        return int(rng.integers(0, 35))

    def walk(x, y, s, depth):
        if x >= w or y >= h:
            return
        log2 = s.bit_length() - 1
        must = log2 > max_log2 or x + s > w or y + s > h
        p = split[min(depth, len(split) - 1)]
        if log2 > min_log2 and (must or rng.random() < p):
            half = s // 2
            for (dx, dy) in ((0, 0), (half, 0), (0, half), (half, half)):      # z-scan
                walk(x + dx, y + dy, half, depth + 1)
        else:
            tus.append((x, y, log2, mode(), 0))

    for cy in range(0, h, ctu):
        for cx in range(0, w, ctu):
            walk(cx, cy, ctu, 0)
            first.append(len(tus))
    return np.array(tus, dtype=INTRA_TU_DTYPE), np.array(first, dtype=np.uint32)


def intra_qp(qp=27):
    """The five arguments of hevcasm_quantize / hevcasm_quantize_inverse per TU size (rows: log2 - 2) that an HEVC encoder uses at this QP for 8-bit
    video with flat scaling lists, int32 [4][5] = { q_scale, q_shift, q_offset, dq_scale, dq_shift }:
    forward: quantScales[qp % 6], shift 14 + qp / 6 + (15 - 8 - log2N), rounding 171 / 512 (intra) -- as the reference's offset << (shift - 16)
    (quantize.c:160-186) that is offset 171 << 7; inverse: levelScale[qp % 6] << (qp / 6), shift 6 - (15 - 8 - log2N) = log2N - 1 (H.265 8.6.4.2 with
    m = 16 folded into the shift).  All within the reference's asserts (quantize.c:162-168).  The reference's self-test parameters (quantize.c:149-150,
    :277-279) are INTRA_QP_REFERENCE."""
    assert 0 <= qp <= 51
    quant_scales = (26214, 23302, 20560, 18396, 16384, 14564)
    level_scale = (40, 45, 51, 57, 64, 72)
    return np.array([(quant_scales[qp % 6], 21 - log2 + qp // 6, 171 << 7, level_scale[qp % 6] << (qp // 6), log2 - 1) for log2 in (2, 3, 4, 5)], dtype=np.int32)


INTRA_QP_REFERENCE = np.array([[51, 20, 14, 51, 14]] * 4, dtype=np.int32)


# ---------------------------------------------------------------------------------------------
# a frame's inter partition: every PU size of the reference's list (pred_inter.c:438-444), three planes, one launch (hevcasm_hip_pred_*_runs)

PRED_RUN_DTYPE = np.dtype([("first", "<u4"), ("count", "<u4"), ("taps", "u1"), ("w", "u1"), ("h", "u1"), ("plane", "u1")])      # = hevcasm_hip_pred_run, 12 bytes


def inter_partition(w, h, seed, bi=False, split=(0.75, 0.6, 0.5, 0.4)):
    """Cuts a w x h luma plane (multiples of 8) into coding units by a random quad-tree (64 ... 8) and every CU into prediction units by a random HEVC part
    mode: 2Nx2N, 2NxN, Nx2N, and the asymmetric 2NxnU / 2NxnD / nLx2N / nRx2N (CUs of 16 and up) -- which between them produce exactly the 24 sizes of
    pred_inter.c:438-444.  8x4 / 4x8 are uni-prediction only (HEVC forbids them for bi; the reference lists 22 sizes for bi, pred_inter.c:687-693).
    -> list of (x, y, pw, ph) in coding order."""
    rng = np.random.default_rng(seed)
    pus = []

    def cu(x, y, s):
        modes = ["2Nx2N", "2NxN", "Nx2N"]
        if s >= 16:
            modes += ["2NxnU", "2NxnD", "nLx2N", "nRx2N"]
        if s == 8 and bi:
            modes = ["2Nx2N"]
        m = modes[int(rng.integers(0, len(modes)))]
        q = s // 4
        parts = {"2Nx2N": [(0, 0, s, s)], "2NxN": [(0, 0, s, s // 2), (0, s // 2, s, s // 2)], "Nx2N": [(0, 0, s // 2, s), (s // 2, 0, s // 2, s)],
                 "2NxnU": [(0, 0, s, q), (0, q, s, s - q)], "2NxnD": [(0, 0, s, s - q), (0, s - q, s, q)],
                 "nLx2N": [(0, 0, q, s), (q, 0, s - q, s)], "nRx2N": [(0, 0, s - q, s), (s - q, 0, q, s)]}[m]
        for (dx, dy, pw, ph) in parts:
            pus.append((x + dx, y + dy, pw, ph))

    def walk(x, y, s, depth):
        if x >= w or y >= h:
            return
        must = x + s > w or y + s > h
        if s > 8 and (must or rng.random() < split[min(depth, len(split) - 1)]):
            for (dx, dy) in ((0, 0), (s // 2, 0), (0, s // 2), (s // 2, s // 2)):
                walk(x + dx, y + dy, s // 2, depth + 1)
        else:
            cu(x, y, s)

    for cy in range(0, h, 64):
        for cx in range(0, w, 64):
            walk(cx, cy, 64, 0)
    return pus


def pred_frame_case(w, h, seed, bi=False, border=80, kind="noise", chroma=True):
    """One 4:2:0 frame's motion compensation as RUNS for hevcasm_hip_pred_uni_8to8_runs / _bi_: luma (8 taps, quarter-pel) on plane 0, Cb and Cr (4 taps,
    eighth-pel, the same entries) on planes 1 and 2; PUs sorted by (plane, size).  Integer vectors in +-8 luma samples, all fractional phases.
    Keys: planes = [dict(stride_dst, dst_bytes, stride_ref, ref0, ref1)], runs (PRED_RUN_DTYPE), dst_off, ref0_off, ref1_off, frac0, frac1."""
    assert w % 8 == 0 and h % 8 == 0
    rng = np.random.default_rng(seed + 1)
    pus = np.array(inter_partition(w, h, seed, bi=bi), dtype=np.int64)
    nref = 2 if bi else 1
    mv = rng.integers(-32, 33, size=(len(pus), nref, 2))      # quarter-pel units, +-8 integer samples
    planes, runs = [], []
    arrays = dict(dst_off=[], ref0_off=[], ref1_off=[], frac0=[], frac1=[])
    at = 0
    for chroma_pass in ((False, True) if chroma else (False,)):
        sh = 1 if chroma_pass else 0
        pw_, ph_ = w >> sh, h >> sh
        b = border >> sh
        stride_dst, stride_ref = pw_, pw_ + 2 * b
        x, y, bw, bh = pus[:, 0] >> sh, pus[:, 1] >> sh, pus[:, 2] >> sh, pus[:, 3] >> sh
        order = np.lexsort((np.arange(len(pus)), bh, bw))      # by (width, height), coding order inside a size
        plane_ids = (1, 2) if chroma_pass else (0,)
        for pid in plane_ids:
            planes.append(dict(stride_dst=stride_dst, dst_bytes=stride_dst * ph_, stride_ref=stride_ref,
                               ref0=fill_u8(stride_ref * (ph_ + 2 * b), seed + 10 + 2 * pid, kind), ref1=fill_u8(stride_ref * (ph_ + 2 * b), seed + 11 + 2 * pid, kind)))
        arrays["dst_off"].append((y * stride_dst + x)[order])
        for r in range(2):
            m = mv[:, min(r, nref - 1)]
            # luma: quarter-pel vector -> integer part m >> 2, fraction m & 3; chroma (4:2:0): the same vector in eighth-pel units of the half-size plane
            ix, iy = (m[:, 0] >> (2 + sh)), (m[:, 1] >> (2 + sh))
            fx, fy = (m[:, 0] & (3 if not sh else 7)), (m[:, 1] & (3 if not sh else 7))
            arrays["ref%d_off" % r].append(((y + b + iy) * stride_ref + x + b + ix)[order])
            arrays["frac%d" % r].append((fx | (fy << 4))[order])
        sizes = np.stack([bw[order], bh[order]], axis=1)
        start = 0
        for i in range(1, len(order) + 1):
            if i == len(order) or (sizes[i] != sizes[start]).any():
                for pid in plane_ids:
                    runs.append((at + start, i - start, 4 if chroma_pass else 8, sizes[start][0], sizes[start][1], pid))
                start = i
        at += len(order)
    out = dict(w=w, h=h, bi=bi, planes=planes, runs=np.array(runs, dtype=PRED_RUN_DTYPE), n_pus=len(pus))
    out["dst_off"] = np.concatenate(arrays["dst_off"]).astype(np.uint32)
    for r in range(2):
        out["ref%d_off" % r] = np.concatenate(arrays["ref%d_off" % r]).astype(np.uint32)
        out["frac%d" % r] = np.concatenate(arrays["frac%d" % r]).astype(np.uint8)
    return out


# the same partition for the motion search's block costs: one launch (hevcasm_hip_sad_runs / _sad_multiref_runs / _ssd_runs)

COST_RUN_DTYPE = np.dtype([("first", "<u4"), ("count", "<u4"), ("rect", "<u4")])      # = hevcasm_hip_cost_run, 12 bytes


def cost_frame_case(w, h, seed, border=80, kind="noise", search=7):
    """One frame's PUs (inter_partition: all 24 sizes of pred_inter.c:438-444) as RUNS for the block-cost entry points: the source plane w x h, a reference plane
    with a `border`, per PU one candidate (ref_off) and four (ref_off4) within +-search pixels.  PUs sorted by size, coding order inside a size.
    Keys: src, stride_src, ref, stride_ref, src_off, ref_off, ref_off4, runs (COST_RUN_DTYPE), sizes [(w, h, first, count)]."""
    assert w % 8 == 0 and h % 8 == 0
    rng = np.random.default_rng(seed + 3)
    pus = np.array(inter_partition(w, h, seed), dtype=np.int64)
    order = np.lexsort((np.arange(len(pus)), pus[:, 3], pus[:, 2]))
    pus = pus[order]
    stride_src, stride_ref = w, w + 2 * border
    d = rng.integers(-search, search + 1, size=(len(pus), 4, 2))
    src_off = pus[:, 1] * stride_src + pus[:, 0]
    ref_off4 = (pus[:, 1, None] + border + d[:, :, 1]) * stride_ref + pus[:, 0, None] + border + d[:, :, 0]
    runs, sizes, start = [], [], 0
    for i in range(1, len(pus) + 1):
        if i == len(pus) or (pus[i, 2:] != pus[start, 2:]).any():
            pw, ph = int(pus[start, 2]), int(pus[start, 3])
            runs.append((start, i - start, (pw << 8) | ph))
            sizes.append((pw, ph, start, i - start))
            start = i
    return dict(w=w, h=h, src=fill_u8(stride_src * h, seed + 20, kind), stride_src=stride_src, ref=fill_u8(stride_ref * (h + 2 * border), seed + 21, kind), stride_ref=stride_ref,
                src_off=src_off.astype(np.uint32), ref_off=np.ascontiguousarray(ref_off4[:, 0]).astype(np.uint32), ref_off4=np.ascontiguousarray(ref_off4).astype(np.uint32).reshape(-1),
                runs=np.array(runs, dtype=COST_RUN_DTYPE), sizes=sizes, n_pus=len(pus))


# ---------------------------------------------------------------------------------------------
# a frame's transform units: the four sizes of the reference's transform tables (residual_decode.h:54, :82), one launch per stage (hevcasm_hip_*_runs over
# hevcasm_hip_tu_run: transform, quantize, inverse_transform_add, dequant_inverse_transform_add)

TU_RUN_DTYPE = np.dtype([("first", "<u4"), ("count", "<u4"), ("coeff_first", "<u4"), ("log2", "u1"), ("tr_type", "u1"), ("plane", "<u2"),
                         ("scale", "<i4"), ("shift", "<i4"), ("offset", "<i4")])      # = hevcasm_hip_tu_run, 28 bytes


def tu_partition(w, h, seed, split=(0.8, 0.6, 0.45), dst_share=0.3):
    """Cuts a w x h plane (multiples of 4) into transform units by a random quad-tree from 32x32 down to 4x4 (an inter CU's residual quad-tree; forced splits at
    the plane's edges); a share of the 4x4 units takes the DST (HEVC: intra luma 4x4).  -> list of (x, y, log2, tr_type) in z-scan order."""
    assert w % 4 == 0 and h % 4 == 0
    rng = np.random.default_rng(seed)
    tus = []

    def walk(x, y, s, depth):
        if x >= w or y >= h:
            return
        must = x + s > w or y + s > h
        if s > 4 and (must or rng.random() < split[min(depth, len(split) - 1)]):
            for (dx, dy) in ((0, 0), (s // 2, 0), (0, s // 2), (s // 2, s // 2)):
                walk(x + dx, y + dy, s // 2, depth + 1)
        else:
            tus.append((x, y, s.bit_length() - 1, int(s == 4 and rng.random() < dst_share)))

    for ty in range(0, h, 32):
        for tx in range(0, w, 32):
            walk(tx, ty, 32, 0)
    return tus


def tu_frame_case(w, h, seed, qp=27, kind="video", residual="residual9", pad=0, src_kind=None, raster=False):
    """One plane's TUs (tu_partition) as RUNS for the four transform-stage entry points: an int16 residual plane and 8-bit prediction / destination planes (stride
    w + pad; offsets in elements of each), the TUs sorted by (size, type) with z-scan order inside a run, coefficients of the runs laid end to end (w x h in all),
    every run with the (de)quantiser parameters HEVC uses for its size at `qp` (intra_qp).  Keys: residual, pred, src, stride, off (uint32, shared by all planes),
    runs_q / runs_dq (TU_RUN_DTYPE with the quantiser's / the dequantiser's parameters), sizes [(log2, tr_type, first, count, coeff_first)], n_tus, n_coeffs."""
    tus = np.array(tu_partition(w, h, seed), dtype=np.int64)
    order = np.lexsort((np.arange(len(tus)), -tus[:, 3], tus[:, 2]))      # 4x4 DST, 4x4 DCT, 8x8, 16x16, 32x32
    if raster:      # inside a run by rows of the plane instead of z-scan: neighbouring lanes then read neighbouring pieces of a row wherever the units are neighbours
        order = np.lexsort((tus[:, 0], tus[:, 1], -tus[:, 3], tus[:, 2]))
    tus = tus[order]
    stride = w + pad
    off = (tus[:, 1] * stride + tus[:, 0]).astype(np.uint32)
    q = intra_qp(qp)
    runs_q, runs_dq, sizes, start, coeff = [], [], [], 0, 0
    for i in range(1, len(tus) + 1):
        if i == len(tus) or (tus[i, 2:] != tus[start, 2:]).any():
            log2, tr = int(tus[start, 2]), int(tus[start, 3])
            qs, qsh, qo, ds, dsh = (int(v) for v in q[log2 - 2])
            runs_q.append((start, i - start, coeff, log2, tr, 0, qs, qsh, qo))
            runs_dq.append((start, i - start, coeff, log2, tr, 0, ds, dsh, 0))
            sizes.append((log2, tr, start, i - start, coeff))
            coeff += (i - start) << (2 * log2)
            start = i
    assert coeff == w * h
    return dict(w=w, h=h, stride=stride, residual=fill_i16(stride * h, seed + 30, residual), pred=fill_u8(stride * h, seed + 31, kind), off=off,
                src=fill_u8(stride * h, seed + 32, kind if src_kind is None else src_kind),      # a source plane for the fused residual loop (src - pred is its residual)
                runs_q=np.array(runs_q, dtype=TU_RUN_DTYPE), runs_dq=np.array(runs_dq, dtype=TU_RUN_DTYPE), sizes=sizes, n_tus=len(tus), n_coeffs=coeff, tus=tus)
