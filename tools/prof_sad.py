#!/usr/bin/env python3
"""Tiny driver for counter profiling: N launches of one SAD configuration over distinct, cold
buffers.  usage: prof_sad.py W H WAYS MODE [REPS]   (MODE as tools/explore_sad.py)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from hevcasm_amd import batch as B
from explore_sad import variant

w, h, ways, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
n = 1 << 20
cs = [variant(w, h, n, mode, 300 + i) for i in range(reps)]
for c in cs:
    so, ro = B.off32(c["src_off"]), B.off32(c["ref_off"])
    if ways == 4:
        ro = torch.stack([ro, ro + 1, ro + c["stride_ref"], ro + c["stride_ref"] + 1], dim=1).contiguous().view(-1)
        B.sad_multiref(c["src"], c["stride_src"], c["ref"], c["stride_ref"], so, ro, w, h)
    else:
        B.sad(c["src"], c["stride_src"], c["ref"], c["stride_ref"], so, ro, w, h)
torch.cuda.synchronize()
print("done", w, h, ways, mode)
