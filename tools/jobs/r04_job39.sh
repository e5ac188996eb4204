#!/bin/bash
# round 4, GPU job 39: the access pattern of the narrow four-reference SAD without arithmetic, per-candidate rows (deal A) against a union window (deal B): tools/ubench/sad4_mix.hip
O=gpurun_out/r4aj; mkdir -p $O
timeout -k 10 300 hevcasm_amd/lib_ab/sad4_mix > $O/sad4_mix.txt 2>&1; echo "rc=$?"
echo job39 done
