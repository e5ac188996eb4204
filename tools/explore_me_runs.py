#!/usr/bin/env python3
"""Where the one-launch search of a 1080p partition spends its time: the launch over subsets of the runs (by block area), HIP events, +-8 and +-16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hevcasm_amd import batch as B, workload as wl

c = wl.cost_frame_case(1920, 1080, 950, border=80)
d = dict(src=B.u8(c["src"]), ref=B.u8(c["ref"]), so=B.off32(c["src_off"]), ro=B.off32(c["ref_off"]))
res = torch.zeros((c["n_pus"], 3), dtype=torch.int32, device="cuda")
runs = c["runs"]
area = (runs["rect"] >> 8) * (runs["rect"] & 0xff)


def timed(sel, rng, reps=20):
    r = np.ascontiguousarray(runs[sel])
    if len(r) == 0:
        return 0.0, 0
    for _ in range(3):
        B.me_full_search_runs(d["src"], c["stride_src"], d["ref"], c["stride_ref"], d["so"], d["ro"], r, rng, out=res)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); B.me_full_search_runs(d["src"], c["stride_src"], d["ref"], c["stride_ref"], d["so"], d["ro"], r, rng, out=res); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)) * 1e3, int(r["count"].sum())


for rng in (8, 16):
    print("range +-%d" % rng)
    for name, sel in (("all", area > 0), ("area >= 2048", area >= 2048), ("512 <= area < 2048", (area >= 512) & (area < 2048)), ("128 <= area < 512", (area >= 128) & (area < 512)), ("area < 128", area < 128)):
        us, n = timed(sel, rng)
        px = int((area[sel] * runs["count"][sel]).sum())
        print("  %-22s %6d blocks %9d px  %7.1f us  (%.2f us per block, %.1f ns per kilopixel-candidate)" % (name, n, px, us, us / max(n, 1), us * 1e3 / max(px * (2 * rng + 1) ** 2 / 1e3, 1)))
    for (w, h) in ((64, 64), (32, 32), (16, 16), (8, 8)):
        sel = runs["rect"] == ((w << 8) | h)
        us, n = timed(sel, rng)
        print("  only %2dx%-2d             %6d blocks             %7.1f us  (%.2f us per block)" % (w, h, n, us, us / max(n, 1)))
