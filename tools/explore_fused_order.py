#!/usr/bin/env python3
"""Why does dequant_inverse_transform_add dct32 lose to the unfused kernel at 2^28 coefficients per launch and not at 2^26
?  Same inputs, both orders, interleaved launches, median and min of 9.
usage: explore_fused_order.py [LOG2_COEFFS ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hevcasm_amd import batch as B, workload as wl

torch.cuda.set_device(0)


def t(fn, reps=9):
    out = []
    for i in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(i); e1.record(); torch.cuda.synchronize()
        if i:
            out.append(e0.elapsed_time(e1))
    return float(np.median(out)), float(min(out))


for lgc in [int(a) for a in sys.argv[1:]] or [26, 28]:
    for lg in (4, 5):
        n = (1 << lgc) >> (2 * lg)
        for dist in ("natural", "full"):
            cs = [wl.inverse_case(0, lg, n, 70 + i, dist, frame=(wl.FRAME_W, wl.FRAME_H)) for i in range(2)]
            devs = [dict(pred=B.u8(c["pred"]), pred_off=B.off32(c["pred_off"]), dst=torch.zeros(c["dst_bytes"], dtype=torch.uint8, device="cuda"),
                         dst_off=B.off32(c["dst_off"]), coeffs=B.i16(c["coeffs"])) for c in cs]
            c0 = cs[0]
            plain = lambda i: B.inverse_transform_add(devs[i % 2]["dst"], c0["stride_dst"], devs[i % 2]["pred"], c0["stride_pred"], devs[i % 2]["dst_off"], devs[i % 2]["pred_off"], devs[i % 2]["coeffs"], 0, lg)
            fused = lambda i: B.dequant_inverse_transform_add(devs[i % 2]["dst"], c0["stride_dst"], devs[i % 2]["pred"], c0["stride_pred"], devs[i % 2]["dst_off"], devs[i % 2]["pred_off"], devs[i % 2]["coeffs"], 51, 14, 0, lg)
            fused_id = lambda i: B.dequant_inverse_transform_add(devs[i % 2]["dst"], c0["stride_dst"], devs[i % 2]["pred"], c0["stride_pred"], devs[i % 2]["dst_off"], devs[i % 2]["pred_off"], devs[i % 2]["coeffs"], 1, 0, 0, lg) if False else None
            rows = []
            for name, fn in (("plain", plain), ("fused", fused), ("plain", plain), ("fused", fused)):
                rows.append((name,) + t(fn))
            print("2^%d coeffs dct%d %-8s " % (lgc, 1 << lg, dist) + "  ".join("%s med %.4f min %.4f" % r for r in rows), flush=True)
            del cs, devs
            torch.cuda.empty_cache()
