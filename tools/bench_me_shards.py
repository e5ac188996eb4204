#!/usr/bin/env python3
"""What one rank's launch of BASELINE config #5 costs when a 3840x2160 frame's CTU rows are split over W ranks (W = 1, 2, 4, 8), measured
on ONE GPU: each rank's rows are searched by their own launch (hevcasm_hip_me_full_search_frame_rows on the rank's slab), timed with HIP
events over `reps` back-to-back launches; the frame's time at W ranks is the slowest rank's.  (The real thing -- W processes on W GPUs --
is `bench.py --workload me5` under torch.distributed.run; this is the single-GPU rehearsal of its per-rank launch sizes.)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hevcasm_amd import batch as B, me_frame, shard

fw, fh, ctu, R, reps = 3840, 2160, 64, 64, 40
f = me_frame.synthetic_frame_pair(fw, fh, R, seed=500)
rows, cols = -(-fh // ctu), -(-fw // ctu)
for unit in ("rows", "ctus"):
    for world in (1, 2, 4, 8):
        worst = 0.0
        per_rank = []
        for rank in range(world):
            if unit == "rows":
                r0, r1 = shard.ctu_row_range(rows, rank, world)
                first, last = r0 * cols, r1 * cols
            else:
                first, last = shard.ctu_range(rows * cols, rank, world)
                r0, r1 = shard.rows_of_ctus(first, last, cols)
            sl = me_frame.Slab(f, ctu, R, r0, r1)
            cur, ref = torch.from_numpy(sl.cur).cuda(), torch.from_numpy(sl.ref).cuda()
            out = torch.empty((last - first, 3), dtype=torch.int32, device="cuda")
            run = lambda: B.me_full_search_frame_ctus(cur, sl.cur_byte0, sl.stride_cur, ref, sl.ref_byte0, sl.stride_ref, fw, fh, ctu, R, first, last - first, out=out)
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            per_rank.append(round(ms, 4))
            worst = max(worst, ms)
        print(json.dumps(dict(kernel="me frame split by %s x%d" % (unit, world), size="4K +-64", ms=round(worst, 4), per_rank_ms=per_rank, frames_per_s=round(1e3 / worst, 1))), flush=True)
