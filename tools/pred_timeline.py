#!/usr/bin/env python3
"""Per-wave timeline of the prediction kernel from in-kernel s_memtime stamps (a library built with -DPRED_STAMPS):
    make -C hevcasm_amd/csrc OUT=../lib_ab/stamps OBJ=build_stamps EXTRA=-DPRED_STAMPS
    HEVCASM_HIP_LIB=$PWD/hevcasm_amd/lib_ab/stamps/libhevcasm_hip.so python tools/pred_timeline.py [uni|bi] TAPS W H
Stamps: 0 entry, 1 descriptors arrived, 2 first window row unpacked, 3 prologue (T-1 rows) unpacked, 4+y output row y stored (y < 24)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hevcasm_amd import batch as B, capi, workload as wl

kind, taps, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
bi = kind == "bi"
fw, fh = (1920, 1080) if taps == 8 else (960, 540)
n = (fw // w) * (fh // h) * 64
L = capi.lib()
L.hevcasm_hip_debug_pred_stamps.argtypes = [C.c_void_p]
stamps = torch.zeros(1 << 22, dtype=torch.int64, device="cuda")
res = []
for i in range(3):
    c = wl.pred_case(taps, w, h, n, 50 + i, bi=bi, frame=(fw, fh), device="cuda")
    dst = torch.zeros(c["dst_bytes"], dtype=torch.uint8, device="cuda")
    stamps.zero_()
    assert L.hevcasm_hip_debug_pred_stamps(stamps.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    if bi:
        B.pred_bi(dst, c["stride_dst"], c["ref0"], c["ref1"], c["stride_ref"], B.off32(c["dst_off"]), B.off32(c["ref0_off"]), B.off32(c["ref1_off"]), B.u8(c["frac0"]), B.u8(c["frac1"]), taps, w, h)
    else:
        B.pred_uni(dst, c["stride_dst"], c["ref0"], c["stride_ref"], B.off32(c["dst_off"]), B.off32(c["ref0_off"]), B.u8(c["frac0"]), taps, w, h)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    s = stamps.cpu().numpy().reshape(-1, 32)
    s = s[s[:, 0] != 0]
    res = (ms, s)
ms, s = res
rows = int((s[0, 4:28] != 0).sum())      # output rows this wave stored (tall PUs of small launches are cut into row segments)
print("%s %d-tap %dx%d: %d waves of %d output rows, %.4f ms with instrumentation (includes the tool's descriptor uploads; shader-clock cycles below)" % (kind, taps, w, h, len(s), rows, ms))
def pct(a): return "med %7d  p10 %7d  p90 %7d" % (np.median(a), np.percentile(a, 10), np.percentile(a, 90))
print("  lifetime (entry -> last row stored)", pct(s[:, 4 + rows - 1] - s[:, 0]))
print("  entry -> descriptors               ", pct(s[:, 1] - s[:, 0]))
print("  descriptors -> first row unpacked  ", pct(s[:, 2] - s[:, 1]))
print("  first row -> prologue done         ", pct(s[:, 3] - s[:, 2]))
print("  prologue done -> row 0 stored      ", pct(s[:, 4] - s[:, 3]))
for y in range(1, rows):
    print("  row %2d                             " % y, pct(s[:, 4 + y] - s[:, 3 + y]))
