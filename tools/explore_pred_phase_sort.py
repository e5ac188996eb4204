#!/usr/bin/env python3
"""What sorting a luma bi-prediction batch by phase class COSTS before any pass is skipped (round-4 verdict, next #5: bucket the PUs by
(xFrac == 0, yFrac == 0) per reference so that waves could skip a filter pass).  The shipped kernels skip nothing, so the time of the sorted batch against
the raster-ordered one is the price of the lost locality alone: a kernel that skipped passes would have to win that back first.  Also: the share of
references a class sort could spare (uniform phases)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hevcasm_amd import batch as B, workload as wl


def timed(fn, reps=6):
    ts = []
    for i in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if i: ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


for (w, h) in ((16, 16), (32, 32), (8, 8), (64, 64)):
    n = (1920 // w) * (1080 // h) * 64
    c = wl.pred_case(8, w, h, n, 77, bi=True, frame=(1920, 1080), device="cuda")
    f0, f1 = c["frac0"], c["frac1"]
    # class of a PU: for each reference (xFrac != 0, yFrac != 0) -> 16 classes
    cls = ((f0 & 15) != 0).astype(int) | (((f0 >> 4) != 0).astype(int) << 1) | (((f1 & 15) != 0).astype(int) << 2) | (((f1 >> 4) != 0).astype(int) << 3)
    order = np.argsort(cls, kind="stable")               # raster order kept inside a class
    dst = torch.zeros(c["dst_bytes"], dtype=torch.uint8, device="cuda")
    def dev(o):
        return dict(dst_off=B.off32(c["dst_off"][o]), r0=B.off32(c["ref0_off"][o]), r1=B.off32(c["ref1_off"][o]), f0=B.u8(f0[o]), f1=B.u8(f1[o]))
    a, s = dev(np.arange(n)), dev(order)
    run = lambda d: B.pred_bi(dst, c["stride_dst"], c["ref0"], c["ref1"], c["stride_ref"], d["dst_off"], d["r0"], d["r1"], d["f0"], d["f1"], 8, w, h)
    t_raster, t_sorted = timed(lambda: run(a)), timed(lambda: run(s))
    passes = 2 * 2 * n                                      # horizontal + vertical per reference
    spared = int(((f0 & 15) == 0).sum() + ((f0 >> 4) == 0).sum() + ((f1 & 15) == 0).sum() + ((f1 >> 4) == 0).sum())
    print("pred_bi_8tap %2dx%-2d 64 frames: raster order %.4f ms, sorted by phase class %.4f ms (%+.1f %%); filter passes a perfect skip would spare: %.1f %%"
          % (w, h, t_raster, t_sorted, 100 * (t_sorted / t_raster - 1), 100.0 * spared / passes), flush=True)
    del c, dst, a, s
    torch.cuda.empty_cache()
