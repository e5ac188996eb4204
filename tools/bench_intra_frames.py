#!/usr/bin/env python3
"""Intra frames (hevcasm_hip_intra_reconstruct_frames): time per 1080p frame alone (the dependency-limited latency) and in batches (throughput),
for several TU-size distributions.  One JSON row per configuration on stdout.
usage: bench_intra_frames.py [--frames 1,4,16,64] [--w 1920 --h 1080] [--reps 5] [--decode]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from hevcasm_amd import batch as B, workload as wl

MAPS = {   # name -> (modes, split probabilities by depth)
    "mixed_quadtree": ("mixed", (1.0, 0.7, 0.5, 0.35)),
    "all_32x32": ("mixed", (1.0, 0.0)),
    "all_16x16": ("mixed", (1.0, 1.0, 0.0)),
    "all_8x8": ("mixed", (1.0, 1.0, 1.0, 0.0)),
    "all_4x4": ("mixed", (1.0, 1.0, 1.0, 1.0)),
    "dc_quadtree": ("dc", (1.0, 0.7, 0.5, 0.35)),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", default="1,4,16,64")
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--ctu", type=int, default=64)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--maps", default=",".join(MAPS))
    ap.add_argument("--decode", action="store_true")
    args = ap.parse_args()
    w, h, ctu = args.w, args.h, args.ctu
    cols, rows = -(-w // ctu), -(-h // ctu)
    steps = cols + 2 * (rows - 1)
    for name in args.maps.split(","):
        modes, split = MAPS[name]
        for F in [int(x) for x in args.frames.split(",")]:
            maps = [wl.intra_tu_map(w, h, ctu, 100 + f, modes=modes, split=split) for f in range(min(F, 4))]
            tus = [maps[f % len(maps)][0] for f in range(F)]
            firsts = [maps[f % len(maps)][1] for f in range(F)]
            job = B.IntraFrames(w, h, ctu, tus, firsts, wl.intra_qp(27), c_idx=0, strong=1, stride=w)
            src = torch.cat([B.u8(wl.frame_2d(w, h, w, 7 + f)) for f in range(min(F, 4))] * ((F + 3) // 4))[: F * w * h].contiguous()
            job.run(src)
            torch.cuda.synchronize()
            job.status()
            times = []
            for _ in range(args.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                job.run(None if args.decode else src)
                e1.record()
                torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1))
            ms = float(np.median(times))
            n_tus = sum(len(t) for t in tus)
            # algorithmic bytes per frame: source + reconstruction (1 byte per pixel each) + levels (2 per pixel) + 8 per TU descriptor + 4 per cbf
            bytes_alg = F * (4 * w * h) + 12 * n_tus
            print(json.dumps(dict(kernel="intra_frames " + name, size="F=%d" % F, family="intra_frames", map=name, decode=args.decode, frame="%dx%d" % (w, h), ctu=ctu, frames=F, tus_per_frame=n_tus // F,
                                  ms=round(ms, 4), ms_min=round(min(times), 4), ms_per_frame=round(ms / F, 4), frames_per_s=round(F / ms * 1e3, 1),
                                  mpixels_per_s=round(F * w * h / ms / 1e3, 1), wavefront_steps=steps,
                                  us_per_step=round(ms * 1e3 / steps, 2) if F == 1 else None,
                                  hbm_gbs=round(bytes_alg / ms / 1e6, 1), hbm_frac=round(bytes_alg / ms / 1e6 / 8000.0, 5))), flush=True)


if __name__ == "__main__":
    main()
