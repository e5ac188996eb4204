#!/usr/bin/env python3
"""One family / one configuration, a few launches on cold buffers -- the target of rocprofv3 --pmc.
usage: prof_one.py pred_uni TAPS W H | pred_bi TAPS W H | inverse LOG2 [LOG2_COEFFS] | deqinv LOG2 [LOG2_COEFFS] | forward LOG2 [LOG2_COEFFS] | sad W H WAYS | me
       | frame_sse W H FRAMES | intra LOG2 [compact] | intra_dc LOG2
Several configurations in one process: separate them with "+" (pick ones whose kernels have different names: the summary is per kernel name)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hevcasm_amd import batch as B, workload as wl

def one(argv):
    kind = argv[0]
    sys.argv = ["prof_one.py"] + list(argv)
    _one(kind)


def _one(kind):
  if kind in ("pred_uni", "pred_bi"):
      taps, w, h = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
      bi = kind == "pred_bi"
      fw, fh = (1920, 1080) if taps == 8 else (960, 540)
      n = (fw // w) * (fh // h) * 64
      for i in range(3):
          c = wl.pred_case(taps, w, h, n, 50 + i, bi=bi, frame=(fw, fh), device="cuda")
          dst = torch.zeros(c["dst_bytes"], dtype=torch.uint8, device="cuda")
          if bi:
              B.pred_bi(dst, c["stride_dst"], c["ref0"], c["ref1"], c["stride_ref"], B.off32(c["dst_off"]), B.off32(c["ref0_off"]), B.off32(c["ref1_off"]), B.u8(c["frac0"]), B.u8(c["frac1"]), taps, w, h)
          else:
              B.pred_uni(dst, c["stride_dst"], c["ref0"], c["stride_ref"], B.off32(c["dst_off"]), B.off32(c["ref0_off"]), B.u8(c["frac0"]), taps, w, h)
  elif kind in ("inverse", "forward", "deqinv"):
      lg = int(sys.argv[2]); n = (1 << (int(sys.argv[3]) if len(sys.argv) > 3 else 26)) >> (2 * lg)
      for i in range(3):
          if kind == "forward":
              f = wl.forward_case(0, lg, n, 80 + i, frame=(wl.FRAME_W, wl.FRAME_H))
              B.transform(B.i16(f["src"]), f["src_stride"], B.off32(f["src_off"]), 0, lg)
              continue
          c = wl.inverse_case(0, lg, n, 70 + i, "natural", frame=(wl.FRAME_W, wl.FRAME_H))
          args = (torch.zeros(c["dst_bytes"], dtype=torch.uint8, device="cuda"), c["stride_dst"], B.u8(c["pred"]), c["stride_pred"], B.off32(c["dst_off"]), B.off32(c["pred_off"]), B.i16(c["coeffs"]))
          if kind == "inverse":
              B.inverse_transform_add(*args, 0, lg)
          else:
              B.dequant_inverse_transform_add(*args, 51, 14, 0, lg)
  elif kind == "sad":
      w, h, ways = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
      n = (1 << 20) if w * h < 4096 else (1 << 19)
      for i in range(3):
          c = wl.sad_case(w, h, n, 100 + i, ways=4, frame=(wl.FRAME_W, wl.FRAME_H), device="cuda")
          ro4 = B.off32(c["ref_off"])
          if ways == 4:
              B.sad_multiref(c["src"], c["stride_src"], c["ref"], c["stride_ref"], B.off32(c["src_off"]), ro4, w, h)
          else:
              B.sad(c["src"], c["stride_src"], c["ref"], c["stride_ref"], B.off32(c["src_off"]), ro4.view(n, 4)[:, 0].contiguous(), w, h)
  elif kind == "frame_sse":
      fw, fh, frames = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
      stride = fw + 128
      for i in range(3):
          a, b = wl.fill_u8(stride * fh * frames + 64, 510 + i, device="cuda"), wl.fill_u8(stride * fh * frames + 64, 520 + i, device="cuda")
          off = B.off32((np.arange(frames, dtype=np.int64) * stride * fh + 64).astype(np.uint32))
          B.frame_sse(a, stride, b, stride, off, off, fw, fh)
  elif kind in ("intra", "intra_dc"):
      lg = int(sys.argv[2]); size = 1 << lg; nb = (1 << 28) >> (2 * lg)
      compact = len(sys.argv) > 3 and sys.argv[3] == "compact"
      packed = (8 << 8) | (lg << 1) | (1 if lg < 5 else 0)
      for i in range(3):
          dst = torch.empty(nb * size * size, dtype=torch.uint8, device="cuda")
          if kind == "intra_dc":
              B.pred_intra(dst, wl.fill_u8(128 * nb + 128, 400 + i, device="cuda"), 128, 1, packed, nb)
          else:
              stride = 4 * size + 1 if compact else 144
              modes = B.u8(np.random.default_rng(i).integers(0, 35, nb).astype(np.uint8))
              B.pred_intra_modes(dst, wl.fill_u8(stride * nb + 192, 450 + i, device="cuda"), stride, modes, packed | (B.INTRA_COMPACT if compact else 0))
  elif kind == "me":
      # config #5: the whole 3840x2160 frame through the frame-rows entry point, one launch per frame pair
      from hevcasm_amd import me_frame
      for i in range(3):
          f = me_frame.synthetic_frame_pair(3840, 2160, 64, seed=900 + i)
          B.me_full_search_frame_rows(B.u8(f["cur"]), 0, f["stride_cur"], B.u8(f["ref"]), 64 * f["stride_ref"] + 64, f["stride_ref"], 3840, 2160, 64, 64, 0, 34)

cfgs, cur = [], []
for a in sys.argv[1:]:
    if a == "+":
        cfgs.append(cur); cur = []
    else:
        cur.append(a)
cfgs.append(cur)
for c in cfgs:
    one(c)
torch.cuda.synchronize()
print("done")
