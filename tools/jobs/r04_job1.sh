#!/bin/bash
# round 4, GPU job 1: counters of the kernels that ship, the fused-vs-unfused dct32 order test, chroma at 256 frames
O=gpurun_out/r4a; mkdir -p $O
python3 tools/explore_fused_order.py 26 28 > $O/fused_order.txt 2>&1
bash tools/pmc2.sh r4a/pmc_g1 tools/prof_one.py pred_uni 4 8 8 + pred_bi 4 8 8 + pred_bi 4 16 32 + pred_bi 8 16 16 + inverse 4 + inverse 5 + forward 4 + forward 5 > $O/pmc_g1.txt 2>&1
bash tools/pmc2.sh r4a/pmc_g2 tools/prof_one.py pred_uni 4 2 8 + pred_bi 8 16 64 + deqinv 5 28 + pred_uni 8 16 16 > $O/pmc_g2.txt 2>&1
bash tools/pmc2.sh r4a/pmc_g3 tools/prof_one.py inverse 5 28 > $O/pmc_g3.txt 2>&1
bash tools/pmc3.sh r4a/pmc3_g1 tools/prof_one.py pred_uni 4 8 8 + pred_bi 4 8 8 + pred_bi 8 16 16 + inverse 5 > $O/pmc3_g1.txt 2>&1
python3 tools/bench_families.py pred4sub256 > $O/pred4_256frames.jsonl 2>&1
python3 tools/bench_families.py pred4sub64 > $O/pred4_64frames.jsonl 2>&1
echo job1 done
