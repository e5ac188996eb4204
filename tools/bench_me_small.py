#!/usr/bin/env python3
"""Full search over lists of SMALL blocks (below me_parts' 2^23-byte-difference threshold: 16x16 / +-64 = 4.3 M, 32x32 / +-32 = 4.3 M) and of the
threshold geometry (32x32 / +-64 = 17 M), a few block counts each: the A/B row for HEVCASM_ME_PARTS = 1 / 2 / 4 (tools/ab_env.py), i.e. whether cutting
such blocks' candidate rows over several workgroups would pay (the threshold was only measured at 64x64 / +-64)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hevcasm_amd import batch as B, workload as wl

for (w, h, r) in ((16, 16, 64), (32, 32, 32), (32, 32, 64)):
    for n in (64, 256, 1024, 4096):
        fw, fh = 1920, 1152
        per = (fw // w) * (fh // h)
        frames = -(-n // per)
        pad = 64
        sp = wl.Plane(fw, fh, frames)
        rp = wl.Plane(fw, fh, frames, border_x=pad, border_y=pad)
        src, ref = wl.fill_u8(sp.size, 5, device="cuda"), wl.fill_u8(rp.size, 6, device="cuda")
        so, ro = B.off32(sp.tile_offsets(n, w, h).astype(np.uint32)), B.off32(rp.tile_offsets(n, w, h).astype(np.uint32))
        out = torch.empty((n, 3), dtype=torch.int32, device="cuda")
        run = lambda: B.me_full_search(src, sp.stride, ref, rp.stride, so, ro, w, h, r, out=out)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record(); torch.cuda.synchronize()
        print(json.dumps(dict(kernel="me_full_search %dx%d +-%d" % (w, h, r), size="n=%d" % n, ms=round(e0.elapsed_time(e1) / 20, 4))), flush=True)
