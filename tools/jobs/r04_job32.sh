#!/bin/bash
# round 4, GPU job 32: counters of pred_hfirst_bi8_kernel at 16x16 and 32x32 (lane-operations per pixel for bench.py's VALU fraction; what bounds it now)
O=gpurun_out/r4ac; mkdir -p $O
for s in "16 16" "32 32"; do
  t=$(echo $s | tr " " "x")
  bash tools/pmc2.sh r4ac/pmc_$t tools/prof_one.py pred_bi 8 $s > $O/pmc_$t.txt 2>&1
done
echo job32 done
