"""A second, independent restatement of H.265 8.4.4.2.3 / .4 / .6 in plain Python (TEST INFRASTRUCTURE): per-sample lookups straight from the text's
formulas, written separately from oracle/hevc_oracle.c (which builds the text's ref[] arrays) so that the two can check each other.  The reference has no C
for these modes: this is what "parity unpinned" rests on -- two restatements of the published text that agree, plus structural properties."""
import numpy as np

ANGLE = [0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32]
INVERSE = {11: -4096, 12: -1638, 13: -910, 14: -630, 15: -482, 16: -390, 17: -315, 18: -256, 19: -315, 20: -390, 21: -482, 22: -630, 23: -910, 24: -1638, 25: -4096}


def predict(nb, mode, log2, edge):
    """nb: one record (left column up to nb[63], top row from nb[64], corner nb[128]) -> N x N array."""
    n = 1 << log2
    top = lambda x: int(nb[128]) if x < 0 else int(nb[64 + x])
    left = lambda y: int(nb[128]) if y < 0 else int(nb[63 - y])
    out = np.zeros((n, n), dtype=np.int64)
    if mode == 0:
        for y in range(n):
            for x in range(n):
                out[y, x] = ((n - 1 - x) * left(y) + (x + 1) * top(n) + (n - 1 - y) * top(x) + (y + 1) * left(n) + n) >> (log2 + 1)
        return out
    if mode == 1:
        dc = (sum(top(i) + left(i) for i in range(n)) + n) >> (log2 + 1)
        out[:] = dc
        if edge:
            out[0, 0] = (left(0) + 2 * dc + top(0) + 2) >> 2
            for i in range(1, n):
                out[0, i] = (top(i) + 3 * dc + 2) >> 2
                out[i, 0] = (left(i) + 3 * dc + 2) >> 2
        return out
    a = ANGLE[mode]
    along, across = (top, left) if mode >= 18 else (left, top)
    ref = lambda i: along(i - 1) if i >= 0 else across(-1 + ((i * INVERSE[mode] + 128) >> 8))
    for v in range(n):
        idx, f = ((v + 1) * a) >> 5, ((v + 1) * a) & 31
        for u in range(n):
            val = ((32 - f) * ref(u + idx + 1) + f * ref(u + idx + 2) + 16) >> 5 if f else ref(u + idx + 1)
            if mode >= 18:
                out[v, u] = val
            else:
                out[u, v] = val
    if edge and mode == 26:
        for y in range(n):
            out[y, 0] = min(255, max(0, top(0) + ((left(y) - top(-1)) >> 1)))
    if edge and mode == 10:
        for x in range(n):
            out[0, x] = min(255, max(0, left(0) + ((top(x) - top(-1)) >> 1)))
    return out


def filter_record(nb, log2, strong):
    """-> (left[0..2N-1], top[0..2N-1], corner) filtered."""
    n = 1 << log2
    c = int(nb[128])
    top = [int(nb[64 + i]) for i in range(2 * n)]
    left = [int(nb[63 - i]) for i in range(2 * n)]
    bi = bool(strong) and n == 32 and abs(c + top[63] - 2 * top[31]) < 8 and abs(c + left[63] - 2 * left[31]) < 8
    def edge(e):
        if bi:
            return [((63 - i) * c + (i + 1) * e[63] + 32) >> 6 for i in range(63)] + [e[63]]
        return [((e[i + 1] + 2 * e[i] + (e[i - 1] if i else c) + 2) >> 2) for i in range(2 * n - 1)] + [e[2 * n - 1]]
    return edge(left), edge(top), c if bi else (left[0] + 2 * c + top[0] + 2) >> 2
