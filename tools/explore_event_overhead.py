import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from hevcasm_amd import batch as B, workload as wl

def mk(frames, i):
    taps, w, h = 8, 16, 16
    n = (1920 // w) * (1080 // h) * frames
    c = wl.pred_case(taps, w, h, n, 50 + i, bi=False, frame=(1920, 1080), device="cuda")
    d = dict(dst=torch.zeros(c["dst_bytes"], dtype=torch.uint8, device="cuda"), dst_off=B.off32(c["dst_off"]), ref0=c["ref0"], ref0_off=B.off32(c["ref0_off"]), frac0=B.u8(c["frac0"]), c=c)
    return d

def launch(d):
    c = d["c"]
    B.pred_uni(d["dst"], c["stride_dst"], d["ref0"], c["stride_ref"], d["dst_off"], d["ref0_off"], d["frac0"], 8, 16, 16)

for frames in (4, 16, 64):
    sets = [mk(frames, i) for i in range(4)]
    for s in sets: launch(s)
    torch.cuda.synchronize()
    for k in (1, 2, 4, 8):
        ts = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for q in range(k): launch(sets[(rep + q) % 4])
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / k)
        print("frames %3d: %d launches between the events: %.4f ms per launch (median of 5)" % (frames, k, float(np.median(ts))), flush=True)
