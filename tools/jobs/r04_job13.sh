#!/bin/bash
# round 4, GPU job 13: VALU utilisation of the luma prediction kernels by PU size (the roof that actually bounds them), counters only
O=gpurun_out/r4m; mkdir -p $O
for s in "64 64" "32 32" "8 8"; do
  t=$(echo $s | tr " " "x")
  bash tools/pmc2.sh r4m/pmc_$t tools/prof_one.py pred_uni 8 $s + pred_bi 8 $s > $O/pmc_$t.txt 2>&1
done
echo job13 done
