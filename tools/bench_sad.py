#!/usr/bin/env python3
"""The SAD rows of bench.py as bench_families-style JSON lines (one per kernel and size), for tools/ab.py:
    python tools/bench_sad.py [--ways 4] [--blocks N] [--disp D] 64x64 64x32 ...
Same generator, frame geometry and byte accounting as bench.py (config #2); cold inputs (two input sets alternate)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from hevcasm_amd import workload as wl

args = sys.argv[1:]
ways_list, blocks, disp, far = (1, 4), 1 << 20, None, 0
while args and args[0].startswith("--"):
    if args[0] == "--ways": ways_list = (int(args[1]),)
    if args[0] == "--blocks": blocks = int(args[1])
    if args[0] == "--disp": disp = int(args[1])
    if args[0] == "--far": far = int(args[1])      # candidates 1..3 moved up to this many rows up or down (the generic path of the 4-way entry)
    args = args[2:]
sizes = [tuple(int(v) for v in s.split("x")) for s in args] or wl.SAD_BENCH_SIZES
torch.cuda.set_device(0)
if disp is not None:      # another displacement range than config #2's +-7 (exploration: far-apart candidates take the generic path)
    _orig = wl.sad_case
    def sad_case(*a, **k):
        k["max_disp"] = min(disp, 8)
        c = _orig(*a, **k)
        return c
    wl.sad_case = sad_case
for (w, h) in sizes:
    sets = [bench.SizeBatch(w, h, blocks, 1000 + 7 * i, torch.device("cuda", 0)) for i in range(2)]
    if far:
        for b in sets:
            for c in b.chunks:
                g = torch.Generator(device="cuda"); g.manual_seed(5)
                d = torch.randint(-far, far + 1, (c["n"], 4), generator=g, device="cuda", dtype=torch.int64) * c["stride_ref"]
                d[:, 0] = 0
                ro = (c["ref_off4"].view(-1, 4).to(torch.int64) & 0xffffffff) + d
                hi = c["ref"].numel() - (h * c["stride_ref"] + w) - 64
                ro = ro.clamp_(64, hi)
                c["ref_off4"] = (((ro + 2 ** 31) % 2 ** 32) - 2 ** 31).to(torch.int32).contiguous().view(-1)     # u32 bit patterns
    for ways in ways_list:
        ts = []
        for i in range(7):
            b = sets[i % 2]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for c in b.chunks: b.launch(ways, c)
            e1.record(); torch.cuda.synchronize()
            if i: ts.append(e0.elapsed_time(e1))
        ms = float(np.median(ts))
        by = bench.algorithmic_bytes(w, h, ways)
        print(json.dumps(dict(kernel="sad_multiref_4" if ways == 4 else "sad", size="%dx%d" % (w, h), ms=round(ms, 4), blocks=blocks,
                              mblocks_per_s=round(blocks / ms / 1e3, 1), bytes_per_block=by, algorithmic_gb_per_s=round(blocks * by / ms / 1e6, 1),
                              frac_of_hbm_peak=round(blocks * by / ms / 1e6 / 8000.0, 4))), flush=True)
    del sets
    torch.cuda.empty_cache()
