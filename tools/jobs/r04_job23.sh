#!/bin/bash
# round 4, GPU job 23: counters of pred_hfirst_uni8_kernel (what bounds it: fewer VALU instructions than the stream kernel, yet slower)
O=gpurun_out/r4w; mkdir -p $O
for s in "64 64" "16 16"; do
  t=$(echo $s | tr " " "x")
  bash tools/pmc2.sh r4w/pmc_$t tools/prof_one.py pred_uni 8 $s > $O/pmc_$t.txt 2>&1
done
echo job23 done
