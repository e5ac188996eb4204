#!/usr/bin/env python3
"""4-way SAD: does 16-byte alignment of the candidate rows matter?  (decides whether aligned loads +
register funnel shift are worth their VALU cost)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hevcasm_amd import batch as B, workload as wl

def run(w, h, n, mode):
    ts = []
    for i in range(5):
        c = wl.sad_case(w, h, n, 100 + i, ways=4, frame=(wl.FRAME_W, wl.FRAME_H), device="cuda")
        rp = wl.Plane(wl.FRAME_W, wl.FRAME_H, wl.Plane.frames_for(n, w, h, wl.FRAME_W, wl.FRAME_H), border_x=16, border_y=8)
        base = rp.tile_offsets(n, w, h)
        rng = np.random.default_rng(i)
        dy = rng.integers(-7, 8, (n, 4)); dx = rng.integers(-7, 8, (n, 4))
        if mode == "aligned16": dx = rng.integers(-1, 2, (n, 4)) * 16
        elif mode == "aligned4": dx = (dx // 4) * 4
        elif mode == "dy0": dy[:] = 0
        ro = B.off32((base[:, None] + dy * rp.stride + dx).reshape(-1).astype(np.uint32))
        so = B.off32(c["src_off"])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); B.sad_multiref(c["src"], c["stride_src"], c["ref"], c["stride_ref"], so, ro, w, h); e1.record(); torch.cuda.synchronize()
        if i: ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    print("%2dx%-2d x4 %-10s %8.4f ms  %7.1f Mblocks/s" % (w, h, mode, ms, n / ms / 1e3), flush=True)

for (w, h) in ((64, 64), (32, 32), (16, 16)):
    n = (1 << 19) if w * h >= 4096 else (1 << 20)
    for mode in ("random", "aligned4", "aligned16", "dy0"):
        run(w, h, n, mode)
