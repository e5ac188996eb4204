#!/usr/bin/env python3
"""Does the ORDER in which a batch lists its blocks matter to the 64x64 SAD kernels (DRAM page locality, L2 reuse of window halos)?
The same blocks, the descriptor arrays permuted: raster (config #2), vertical strips 4 / 8 / 2 blocks wide (a workgroup = 4 consecutive
blocks stays one row of 4), column-major, random."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from hevcasm_amd import workload as wl

w, h = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "64x64").split("x"))
blocks = 1 << 19
bw, bh = wl.FRAME_W // w, wl.FRAME_H // h          # blocks per frame row / column
per_frame = bw * bh

def order(kind, n):
    idx = np.arange(n)
    f, r, c = idx // per_frame, (idx % per_frame) // bw, idx % bw
    if kind == "raster": key = idx
    elif kind.startswith("strip"):
        k = int(kind[5:]); key = ((f * (bw // k + 1) + c // k) * bh + r) * k + c % k
    elif kind == "colmajor": key = (f * bw + c) * bh + r
    elif kind == "random": key = np.random.default_rng(3).permutation(n)
    return np.argsort(key, kind="stable")

torch.cuda.set_device(0)
sets = [bench.SizeBatch(w, h, blocks, 1000 + 7 * i, torch.device("cuda", 0)) for i in range(2)]
for kind in ("raster", "strip4", "strip8", "strip2", "colmajor", "random", "raster"):
    perm_sets = []
    for b in sets:
        chunks = []
        for c in b.chunks:
            p = torch.from_numpy(order(kind, c["n"])).cuda()
            d = dict(c)
            d["src_off"] = c["src_off"][p].contiguous()
            d["ref_off4"] = c["ref_off4"].view(-1, 4)[p].contiguous().view(-1)
            d["ref_off1"] = c["ref_off1"][p].contiguous()
            chunks.append(d)
        perm_sets.append(chunks)
    for ways in (1, 4):
        ts = []
        for i in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for c in perm_sets[i % 2]: sets[0].launch(ways, c)
            e1.record(); torch.cuda.synchronize()
            if i: ts.append(e0.elapsed_time(e1))
        print("%dx%d ways %d  %-9s %.4f ms per %d blocks" % (w, h, ways, kind, float(np.median(ts)), blocks), flush=True)
