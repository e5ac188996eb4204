#!/bin/bash
# round 4, GPU job 25: the access mix of luma uni-prediction without arithmetic (tools/ubench/pred_mix.hip)
O=gpurun_out/r4y; mkdir -p $O
timeout -k 10 300 hevcasm_amd/lib_ab/pred_mix > $O/pred_mix.txt 2>&1; echo "rc=$?"
echo job25 done
