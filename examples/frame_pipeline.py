#!/usr/bin/env python3
"""One inter-coded 1080p frame through the whole library on the GPU (hevcasm_amd/pipeline.py): full search -> quarter-pel refinement -> prediction ->
forward transform -> quantisation -> fused dequantise + inverse transform + add -> SSD / PSNR.  Prints what came out and how long the chain took.

    python examples/frame_pipeline.py [width height block_size search_range]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hevcasm_amd import me_frame, pipeline

w, h, size, rng = (int(v) for v in (sys.argv[1:5] + ["1920", "1088", "16", "16"][len(sys.argv) - 1:]))
frame = me_frame.synthetic_frame_pair(w, h, rng + 8, seed=1, ctu=size, max_motion=rng - 2)
ops = pipeline.HipOps()
pipeline.stages(frame, ops, size=size, search_range=rng)        # warm-up: uploads, first launches
torch.cuda.synchronize()
t0 = time.perf_counter()
r = pipeline.HipOps.to_numpy(pipeline.stages(frame, ops, size=size, search_range=rng))
ms = (time.perf_counter() - t0) * 1e3
moved = int((np.abs(r["mv"][:, :2]).sum(axis=1) > 0).sum())
print("%dx%d frame, %d blocks of %dx%d, search +-%d: %d blocks moved, %d with a fractional vector, %d coded (cbf), PSNR %.2f dB, chain %.2f ms host-to-host"
      % (w, h, r["blocks"], size, size, rng, moved, int((r["qmv"][:, :2] & 3).any(axis=1).sum()), int((r["cbf"] != 0).sum()), r["psnr"], ms))
