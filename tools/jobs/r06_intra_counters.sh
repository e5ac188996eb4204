#!/bin/bash
# round 6: what a TU costs in the intra frame kernel -- SQ instruction counters (separate --pmc passes, no tracing), one 1080p frame, all-4x4 and all-32x32 partitions
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for m in all_4x4 all_32x32 all_8x8; do
  bash tools/pmc2.sh r06_intra_pmc_$m tools/bench_intra_frames.py --frames 1 --maps $m --reps 2 > gpurun_out/r06_intra_pmc_$m.txt 2>&1
  grep -A40 "intra_frames_kernel" gpurun_out/r06_intra_pmc_$m.txt | head -45
done
