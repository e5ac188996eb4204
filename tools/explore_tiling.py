#!/usr/bin/env python3
"""Sweep the SAD lane mapping (exploration build of sad.hip, -DHEVCASM_EXPLORE_TILING) over
chunk bytes C and rows-in-flight R, on the bench's +-7 displacement pattern and on coherent MVs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from hevcasm_amd import batch as B, capi
from explore_sad import variant

def timeit(w, h, n, mode, ways, reps=4):
    from hevcasm_amd import workload as wl
    if ways == 4 and mode == "both":
        cs = [wl.sad_case(w, h, n, 100 + i, ways=4, frame=(wl.FRAME_W, wl.FRAME_H), device="cuda") for i in range(reps + 1)]
    else:
        cs = [variant(w, h, n, mode, 100 + i) for i in range(reps + 1)]
    ts = []
    ref_check = None
    for i, c in enumerate(cs):
        so, ro = B.off32(c["src_off"]), B.off32(c["ref_off"])
        if ways == 4 and ro.numel() == so.numel():
            ro = torch.stack([ro, ro + 1, ro + c["stride_ref"], ro + c["stride_ref"] + 1], dim=1).contiguous().view(-1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f = B.sad_multiref if ways == 4 else B.sad
        e0.record(); out = f(c["src"], c["stride_src"], c["ref"], c["stride_ref"], so, ro, w, h); e1.record()
        torch.cuda.synchronize()
        if i: ts.append(e0.elapsed_time(e1))
        if i == 1: ref_check = int(out.sum().item())
    ms = float(np.median(ts))
    return ms, ((1 + ways) * w * h + 4 * ways + 4 * (1 + ways)) * n / (ms * 1e-3) / 1e9, ref_check

if __name__ == "__main__":
    n = 1 << 20
    sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1].split(",")]
    ways = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    for (w, h) in sizes:
        nn = n if w * h * (1 + ways) < 8192 else n // 2
        for mode in ("both", "coherent"):
            base = None
            for (c, r) in ((16, 1), (16, 2), (16, 4), (16, 8), (8, 1), (8, 2), (8, 4), (4, 2)):
                for g in (1, 2, 4):
                    if w % c or h % r or 64 % ((w // c) * r): continue
                    os.environ["HEVCASM_SAD_TILING"] = "%d,%d,%d" % (c, r, g)
                    try:
                        ms, gb, chk = timeit(w, h, nn, mode, ways)
                    except capi.HevcasmError as e:
                        continue
                    base = chk if base is None else base
                    print("%2dx%-2d ways%d %-8s C=%2d R=%2d G=%d  %8.4f ms %7.1f GB/s %5.1f%%  %s" % (w, h, ways, mode, c, r, g, ms, gb, gb / 80, "" if chk == base else "CHECKSUM MISMATCH"), flush=True)
