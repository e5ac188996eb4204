#!/usr/bin/env python3
"""1-way SAD, narrow blocks: is the kernel bound by the vector L1's access count?  Same batches with the reference
displacement dx forced to a multiple of 4 (a dword-aligned 8-byte lane load is one access where a misaligned one is
two), to 0, and with all blocks' displacements equal (neighbouring blocks' reference rows contiguous)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hevcasm_amd import batch as B, workload as wl

def run(w, h, n, mode):
    ts = []
    for i in range(6):
        c = wl.sad_case(w, h, n, 100 + i, ways=1, frame=(wl.FRAME_W, wl.FRAME_H), device="cuda")
        rp = wl.Plane(wl.FRAME_W, wl.FRAME_H, wl.Plane.frames_for(n, w, h, wl.FRAME_W, wl.FRAME_H), border_x=16, border_y=8)
        base = rp.tile_offsets(n, w, h)
        rng = np.random.default_rng(i)
        dy = rng.integers(-7, 8, n); dx = rng.integers(-7, 8, n)
        if mode == "aligned4": dx = (dx // 4) * 4
        elif mode == "dx0": dx[:] = 0
        elif mode == "same_mv": dx[:] = 3; dy[:] = -2
        elif mode == "zero_mv": dx[:] = 0; dy[:] = 0
        ro = B.off32((base + dy * rp.stride + dx).astype(np.uint32))
        so = B.off32(c["src_off"])
        out = torch.empty(n, dtype=torch.int32, device="cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); B.sad(c["src"], c["stride_src"], c["ref"], c["stride_ref"], so, ro, w, h, out=out); e1.record(); torch.cuda.synchronize()
        if i: ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    print("%2dx%-2d %-10s %8.4f ms  %7.1f Mblocks/s  %6.0f GB/s" % (w, h, mode, ms, n / ms / 1e3, n * (2 * w * h + 12) / ms / 1e6), flush=True)

for (w, h) in ((16, 16), (8, 8), (16, 64), (32, 32)):
    for mode in ("random", "aligned4", "dx0", "same_mv", "zero_mv"):
        run(w, h, 1 << 20, mode)
