#!/usr/bin/env python3
"""Prediction: how much of the kernel's time is the scatter of the motion vectors?  Same PU tilings with one random
integer vector per PU (the bench workload, [-8, 8]^2) against one vector for the whole batch: neighbouring PUs then read
neighbouring windows, the lines a wave's row load touches are shared by its PUs, and the vector L1 sends a fraction of
the requests to L2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hevcasm_amd import batch as B, workload as wl


def run(taps, w, h, bi, coherent, fracs=None):
    fw, fh = (1920, 1080) if taps == 8 else (960, 540)
    n = (fw // w) * (fh // h) * 64
    ts = []
    for i in range(4):
        c = wl.pred_case(taps, w, h, n, 60 + i, bi=bi, frame=(fw, fh), device="cuda", coherent_mv=coherent, fracs=fracs)
        dst = torch.zeros(c["dst_bytes"], dtype=torch.uint8, device="cuda")
        do = B.off32(c["dst_off"]); f0 = B.u8(c["frac0"]); o0 = B.off32(c["ref0_off"])
        if bi:
            o1 = B.off32(c["ref1_off"]); f1 = B.u8(c["frac1"])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if bi:
            B.pred_bi(dst, c["stride_dst"], c["ref0"], c["ref1"], c["stride_ref"], do, o0, o1, f0, f1, taps, w, h)
        else:
            B.pred_uni(dst, c["stride_dst"], c["ref0"], c["stride_ref"], do, o0, f0, taps, w, h)
        e1.record(); torch.cuda.synchronize()
        if i: ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


if __name__ == "__main__":
    only = sys.argv[1:]
    for (taps, w, h, bi) in ((8, 16, 16, False), (8, 16, 16, True), (8, 8, 8, False), (8, 32, 32, False), (8, 64, 64, False), (8, 64, 64, True), (4, 8, 8, False)):
        a = run(taps, w, h, bi, False)
        b = run(taps, w, h, bi, True)
        c = run(taps, w, h, bi, True, fracs=(0, 0))
        d = run(taps, w, h, bi, False, fracs=(0, 0))
        print("taps %d %2dx%-2d %s: per-PU vectors %.4f ms | one vector %.4f (x%.2f) | one vector, integer %.4f | per-PU vectors, integer %.4f" %
              (taps, w, h, "bi " if bi else "uni", a, b, a / b, c, d), flush=True)
