#!/bin/bash
# round 4, GPU job 28: what coherent motion is worth on the shipped prediction kernels -- one vector per 64 x 64 luma region (a CTU) against one per PU
O=gpurun_out/r4z; mkdir -p $O
python3 tools/bench_families.py predctu0 > $O/pred_per_pu.jsonl 2> $O/pred_per_pu.err; echo "rc=$?"
python3 tools/bench_families.py predctu64 > $O/pred_per_ctu.jsonl 2> $O/pred_per_ctu.err; echo "rc=$?"
python3 tools/bench_families.py predctu256 > $O/pred_per_256.jsonl 2> $O/pred_per_256.err; echo "rc=$?"
echo job28 done
