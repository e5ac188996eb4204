#!/usr/bin/env python3
"""Exploration: how the reference displacement pattern moves the SAD kernel (16x16 / 8x8 / 64x64).
Variants: disp=0 (aligned, coalesced), dx only, dy only, +-7 both (bench default), coherent
(one displacement per 64 consecutive blocks).  Each timed over cold, distinct buffers."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hevcasm_amd import batch as B, workload as wl

def variant(w, h, n, mode, seed):
    c = wl.sad_case(w, h, n, seed, ways=1, frame=(wl.FRAME_W, wl.FRAME_H), device="cuda")
    rp = wl.Plane(wl.FRAME_W, wl.FRAME_H, wl.Plane.frames_for(n, w, h, wl.FRAME_W, wl.FRAME_H), border_x=16, border_y=8)
    base = rp.tile_offsets(n, w, h)
    rng = np.random.default_rng(seed)
    dx = rng.integers(-7, 8, n); dy = rng.integers(-7, 8, n)
    if mode == "zero": dx[:] = 0; dy[:] = 0
    elif mode == "dx": dy[:] = 0
    elif mode == "dy": dx[:] = 0
    elif mode == "coherent":
        dx = np.repeat(dx[::64], 64)[:n]; dy = np.repeat(dy[::64], 64)[:n]
    elif mode == "dx4": dy[:] = 0; dx = (dx // 4) * 4
    elif mode == "dy_dx16": dx = rng.integers(-1, 2, n) * 16
    elif mode == "dy_dx4": dx = (dx // 4) * 4
    elif mode == "dy_dx8": dx = rng.integers(-1, 2, n) * 8
    c["ref_off"] = (base + dy * rp.stride + dx).astype(np.uint32)
    return c

def timeit(w, h, n, mode, reps=5):
    cs = [variant(w, h, n, mode, 100 + i) for i in range(reps + 1)]
    devs = [(c["src"], c["ref"], B.off32(c["src_off"]), B.off32(c["ref_off"]), torch.empty(n, dtype=torch.int32, device="cuda")) for c in cs]
    c0 = cs[0]
    ts = []
    for i, (s, r, so, ro, out) in enumerate(devs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); B.sad(s, c0["stride_src"], r, c0["stride_ref"], so, ro, w, h, out=out); e1.record()
        torch.cuda.synchronize()
        if i: ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    gb = (2 * w * h + 12) * n / (ms * 1e-3) / 1e9
    return ms, gb

if __name__ == "__main__":
    n = 1 << 20
    sizes = [(16, 16), (8, 8), (32, 32), (64, 64)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split("x")) for a in sys.argv[1].split(",")]
    modes = ("zero", "dx4", "dx", "dy", "both", "coherent") if len(sys.argv) < 3 else sys.argv[2].split(",")
    for (w, h) in sizes:
        nn = n if w * h < 4096 else n // 2
        for mode in modes:
            ms, gb = timeit(w, h, nn, mode)
            print("%2dx%-2d %-9s %8.4f ms  %7.1f GB/s  %5.1f%% of 8 TB/s" % (w, h, mode, ms, gb, gb / 80), flush=True)
