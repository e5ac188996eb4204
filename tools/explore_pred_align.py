#!/usr/bin/env python3
"""Prediction: how much would dword-aligned window loads buy?  Same batches with the window origin
(ref_off - 3 for luma) forced to a multiple of 4, against arbitrary alignment."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hevcasm_amd import batch as B, workload as wl

def run(taps, w, h, bi, aligned):
    fw, fh = (1920, 1080) if taps == 8 else (960, 540)
    n = (fw // w) * (fh // h) * 64
    ts = []
    for i in range(4):
        c = wl.pred_case(taps, w, h, n, 60 + i, bi=bi, frame=(fw, fh), device="cuda")
        before = taps // 2 - 1
        offs = []
        for r in range(2 if bi else 1):
            o = c["ref%d_off" % r].astype(np.int64)
            if aligned:
                o = ((o - before) // 4) * 4 + before
            offs.append(B.off32(o.astype(np.uint32)))
        dst = torch.zeros(c["dst_bytes"], dtype=torch.uint8, device="cuda")
        do = B.off32(c["dst_off"]); f0 = B.u8(c["frac0"])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if bi:
            B.pred_bi(dst, c["stride_dst"], c["ref0"], c["ref1"], c["stride_ref"], do, offs[0], offs[1], f0, B.u8(c["frac1"]), taps, w, h)
        else:
            B.pred_uni(dst, c["stride_dst"], c["ref0"], c["stride_ref"], do, offs[0], f0, taps, w, h)
        e1.record(); torch.cuda.synchronize()
        if i: ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))

for (taps, w, h, bi) in ((8, 16, 16, False), (8, 64, 64, False), (8, 8, 8, False), (8, 32, 32, True), (4, 8, 8, False), (4, 16, 16, True)):
    a, b = run(taps, w, h, bi, False), run(taps, w, h, bi, True)
    print("taps %d %2dx%-2d %s: arbitrary %.4f ms, dword-aligned window %.4f ms  (x%.2f)" % (taps, w, h, "bi " if bi else "uni", a, b, a / b), flush=True)
