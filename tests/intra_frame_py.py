"""A whole intra frame in plain Python / numpy (TEST INFRASTRUCTURE): the frame loop of oracle/hevc_oracle.c: orc_intra_reconstruct_frame written a second
time, on different building blocks -- tests/intra_spec.py for prediction and smoothing (per-sample formulas from the text), numpy matrix products for the
transforms, a sample-by-sample scan for H.265 8.4.4.2.2 -- so that the oracle's frame logic (availability, substitution, the smoothing decision, the DST / edge
filter conditions, the level layout) is checked by something that shares no code with it.  Small frames only: this is slow."""
import numpy as np

import intra_spec

DST = np.array([[29, 55, 74, 84], [74, 74, 0, -74], [84, -29, -74, 55], [55, -84, 74, -29]], dtype=np.int64)


def dct(n):
    """The HEVC core transform matrix from its definition: round(64 sqrt(2) cos((2i + 1) k pi / 2N)) with HEVC's hand-tuned entries (H.265 8.6.4.2, transMatrix):
    the 31 magnitudes are typed here from the standard's table, not taken from the oracle."""
    mag = [64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64, 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0]
    m = np.zeros((32, 32), dtype=np.int64)
    for k in range(32):
        for i in range(32):
            a = (k * (2 * i + 1)) % 128
            a = 128 - a if a > 64 else a
            m[k, i] = 64 if k == 0 else (-mag[64 - a] if a > 32 else mag[a])
    return m[:: 32 // n, :n]


def _i16(a):
    return ((a + 32768) % 65536) - 32768


def forward(res, tr_dst, log2):
    """hevcasm_transform (residual_decode.c:855-892): rows, then columns; shifts log2 - 1 and log2 + 6; the stores truncate to 16 bits."""
    m = DST if tr_dst else dct(1 << log2)
    s1, s2 = log2 - 1, log2 + 6
    t = _i16((res.astype(np.int64) @ m.T + (1 << (s1 - 1))) >> s1)         # t[r][k] = sum_i M[k][i] res[r][i]
    return _i16((m @ t + (1 << (s2 - 1))) >> s2)                            # c[k2][k1] = sum_r M[k2][r] t[r][k1]


def inverse(coef, tr_dst, log2):
    """the two passes of hevcasm_inverse_transform_add (residual_decode.c:371-413): columns with shift 7, rows with shift 12, each clipped to int16."""
    m = DST if tr_dst else dct(1 << log2)
    t = np.clip((m.T @ coef.astype(np.int64) + 64) >> 7, -32768, 32767)     # t[y][j] = sum_r M[r][y] c[r][j]
    return np.clip((t @ m + 2048) >> 12, -32768, 32767)                     # res[y][x] = sum_j t[y][j] M[j][x]


def quantize(coef, scale, shift, offset):
    a = np.abs(coef.astype(np.int64))
    q = (a * scale + (offset << (shift - 16))) >> shift
    return np.clip(np.where(coef < 0, -q, q), -32768, 32767)


def dequantize(lev, scale, shift):
    return np.clip((lev.astype(np.int64) * scale + (1 << (shift - 1))) >> shift, -32768, 32767)


def neighbours(recon, decoded, w, h, x0, y0, n):
    """8.4.4.2.2 by scanning samples: -> record (left at [63 - y], top at [64 + x], corner at [128])."""
    scan = [(x0 - 1, y0 + y) for y in range(2 * n - 1, -2, -1)] + [(x0 + x, y0 - 1) for x in range(2 * n)]
    ok = [0 <= x < w and 0 <= y < h and bool(decoded[y // 4, x // 4]) for (x, y) in scan]
    val = [int(recon[y, x]) if o else None for (x, y), o in zip(scan, ok)]
    if not any(ok):
        val = [128] * len(scan)
    else:
        first = ok.index(True)
        for i in range(first):
            val[i] = val[first]
        for i in range(first + 1, len(scan)):
            if val[i] is None:
                val[i] = val[i - 1]
    rec = np.zeros(160, dtype=np.uint8)
    for i in range(2 * n):
        rec[63 - (2 * n - 1 - i)] = val[i]
        rec[64 + i] = val[2 * n + 1 + i]
    rec[128] = val[2 * n]
    return rec


def z_index(x, y):
    z = 0
    for b in range(4):
        z |= ((x >> b) & 1) << (2 * b) | ((y >> b) & 1) << (2 * b + 1)
    return z


def reconstruct(src, tus, qp, w, h, ctu, cidx, strong, levels_in=None):
    """src: h x w array or None (decode).  tus: records with x, y, log2, mode in decoding order.  -> (recon h x w, levels, cbf)."""
    cols, rows = -(-w // ctu), -(-h // ctu)
    recon = np.zeros((h, w), dtype=np.uint8)
    decoded = np.zeros(((h + 3) // 4, (w + 3) // 4), dtype=bool)
    levels = np.zeros(cols * rows * ctu * ctu, dtype=np.int16) if src is not None else np.array(levels_in, dtype=np.int16)
    cbf = np.zeros(len(tus), dtype=np.int32)
    for i, t in enumerate(tus):
        x, y, log2, mode = int(t["x"]), int(t["y"]), int(t["log2"]), int(t["mode"])
        n = 1 << log2
        rec = neighbours(recon, decoded, w, h, x, y, n)
        if cidx == 0 and mode != 1 and n != 4 and min(abs(mode - 26), abs(mode - 10)) > {8: 7, 16: 1, 32: 0}[n]:
            left, top, corner = intra_spec.filter_record(rec, log2, strong)
            rec = rec.copy()
            for k in range(2 * n):
                rec[63 - k], rec[64 + k] = left[k], top[k]
            rec[128] = corner
        pred = intra_spec.predict(rec, mode, log2, cidx == 0 and log2 < 5)
        tr_dst = cidx == 0 and log2 == 2
        q = qp[log2 - 2]
        at = ((y // ctu) * cols + x // ctu) * ctu * ctu + 16 * z_index((x % ctu) // 4, (y % ctu) // 4)
        if src is not None:
            res = src[y:y + n, x:x + n].astype(np.int64) - pred
            lev = quantize(forward(res, tr_dst, log2), int(q[0]), int(q[1]), int(q[2]))
            levels[at:at + n * n] = lev.reshape(-1)
            cbf[i] = np.bitwise_or.reduce(lev.reshape(-1).astype(np.int32))
        lev = levels[at:at + n * n].reshape(n, n)
        res = inverse(dequantize(lev, int(q[3]), int(q[4])), tr_dst, log2)
        recon[y:y + n, x:x + n] = np.clip(_i16(pred + res), 0, 255)
        decoded[y // 4:(y + n) // 4, x // 4:(x + n) // 4] = True
    return recon, levels, cbf
