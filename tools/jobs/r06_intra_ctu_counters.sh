#!/bin/bash
# round 6: what ONE TU costs, without waiting: 256 frames of one 64x64 CTU each (independent, no neighbours to wait for, one wave per CU) -- time per launch / TUs per CTU,
# and the SQ instruction counters of the same launches (separate --pmc passes, no tracing)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for m in all_4x4 all_8x8 all_16x16 all_32x32; do
  timeout -k 10 120 python tools/bench_intra_frames.py --frames 256 --w 64 --h 64 --maps $m --reps 5 | python -c '
import sys, json
for l in sys.stdin:
    r = json.loads(l); print("%-10s TUs/CTU=%-4d ms=%.4f  us/TU=%.3f" % (r["map"], r["tus_per_frame"], r["ms"], r["ms"] * 1e3 / r["tus_per_frame"]))'
done
for m in all_4x4 all_8x8; do
  bash tools/pmc2.sh r06_intra_ctu_pmc_$m tools/bench_intra_frames.py --frames 256 --w 64 --h 64 --maps $m --reps 2 > gpurun_out/r06_intra_ctu_pmc_$m.txt 2>&1
  grep -A40 "intra_frames_kernel" gpurun_out/r06_intra_ctu_pmc_$m.txt | head -45
done
