#!/usr/bin/env python3
"""One inter-coded 1080p frame through the whole library on the GPU (hevcasm_amd/pipeline.py): full search -> quarter-pel refinement -> prediction ->
forward transform -> quantisation -> fused dequantise + inverse transform + add -> SSD / PSNR.  Runs the chain eagerly (one launch after another from
Python) and as ONE captured HIP graph replayed per frame, and prints what came out and what each way costs.

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
for _ in range(10):
    r = pipeline.stages(frame, ops, size=size, search_range=rng)
torch.cuda.synchronize()
eager_ms = (time.perf_counter() - t0) * 1e2
graph, results = ops.capture(frame, size=size, search_range=rng)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    graph.replay()
e1.record()
r = pipeline.HipOps.to_numpy(results)
graph_ms = e0.elapsed_time(e1) / 20
moved = int((np.abs(r["mv"][:, :2]).sum(axis=1) > 0).sum())
print("%dx%d frame, %d blocks of %dx%d, search +-%d: %d blocks moved, %d with a fractional vector, %d coded (cbf), PSNR %.2f dB"
      % (w, h, r["mv"].shape[0], size, size, rng, moved, int((r["qmv"][:, :2] & 3).any(axis=1).sum()), int((r["cbf"] != 0).sum()), pipeline.psnr(int(r["frame_sse"][0]), w, h)))
print("the chain per frame: %.3f ms launched from Python stage by stage, %.3f ms as one replayed HIP graph (%.0f frames/s)" % (eager_ms, graph_ms, 1e3 / graph_ms))

# the same chain on a REAL partition: PUs of every HEVC size from a random quad-tree; prediction, the PUs' SAD and the PUs' SSD are ONE launch each
from hevcasm_amd import workload
pus = workload.inter_partition(w, h, 2)
graph, results = ops.capture(frame, pus=pus, tu=size, search_range=rng)
e0.record()
for _ in range(20):
    graph.replay()
e1.record()
r = pipeline.HipOps.to_numpy(results)
sizes = sorted({(p[2], p[3]) for p in pus})
print("mixed partition: %d PUs of %d sizes, %dx%d transform units: PSNR %.2f dB; prediction cost == refinement cost on %d of %d PUs; %.3f ms per frame as one HIP graph"
      % (len(pus), len(sizes), size, size, pipeline.psnr(int(r["frame_sse"][0]), w, h), int((r["pu_sad"] == r["qmv"][:, 2]).sum()), len(pus), e0.elapsed_time(e1) / 20))

# ... and with the residual on a quad-tree of transform units as well (4x4 DST ... 32x32, each size with the quantiser parameters HEVC gives it at this QP): first a
# launch per transform stage (hevcasm_hip_*_runs), then the whole residual loop as ONE launch from the current and predicted planes
tus = workload.tu_partition(w, h, 3)
for fused in (False, True):
    graph, results = ops.capture(frame, pus=pus, search_range=rng, tus=tus, qp=27, fused=fused)
    e0.record()
    for _ in range(20):
        graph.replay()
    e1.record()
    r = pipeline.HipOps.to_numpy(results)
    print("mixed partition, %d TUs of 5 kinds, %s: PSNR %.2f dB, %d TUs coded; %.3f ms per frame as one HIP graph"
          % (len(tus), "the residual loop in ONE launch" if fused else "a launch per transform stage", pipeline.psnr(int(r["frame_sse"][0]), w, h), int((r["cbf"] != 0).sum()), e0.elapsed_time(e1) / 20))
