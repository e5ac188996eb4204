#!/usr/bin/env python3
"""Is the prediction kernel limited by its memory pattern or by instruction issue?  Same PU count,
random vs coherent integer motion, random vs zero fractions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hevcasm_amd import batch as B, workload as wl

def run(taps, w, h, coherent, fracs, frames=64):
    fw, fh = (1920, 1080) if taps == 8 else (960, 540)
    n = (fw // w) * (fh // h) * frames
    cs = [wl.pred_case(taps, w, h, n, 50 + i, frame=(fw, fh), device="cuda", coherent_mv=coherent, fracs=fracs) for i in range(3)]
    devs = [dict(dst=torch.zeros(c["dst_bytes"], dtype=torch.uint8, device="cuda"), dst_off=B.off32(c["dst_off"]), ref=c["ref0"], ref_off=B.off32(c["ref0_off"]), frac=B.u8(c["frac0"])) for c in cs]
    ts = []
    for i in range(6):
        d = devs[i % 3]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); B.pred_uni(d["dst"], cs[0]["stride_dst"], d["ref"], cs[0]["stride_ref"], d["dst_off"], d["ref_off"], d["frac"], taps, w, h); e1.record()
        torch.cuda.synchronize()
        if i: ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    by = (w + taps - 1) * (h + taps - 1) + w * h + 9
    print("taps %d %2dx%-2d coherent=%d fracs=%-6s %8.4f ms %8.1f Mblocks/s %7.1f GB/s" % (taps, w, h, coherent, fracs, ms, n / ms / 1e3, n * by / ms / 1e6), flush=True)

for taps, w, h in ((8, 16, 16), (4, 16, 16), (8, 64, 64), (8, 8, 8)):
    for coherent in (0, 1):
        for fracs in (None, (0, 0)):
            run(taps, w, h, coherent, fracs)
