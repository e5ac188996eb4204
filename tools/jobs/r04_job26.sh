#!/bin/bash
# round 4, GPU job 26: the access mix of chroma prediction without arithmetic (tools/ubench/pred_mix.hip chroma)
O=gpurun_out/r4y; mkdir -p $O
timeout -k 10 300 hevcasm_amd/lib_ab/pred_mix chroma > $O/pred_mix_chroma.txt 2>&1; echo "rc=$?"
echo job26 done
