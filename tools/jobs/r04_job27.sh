#!/bin/bash
# round 4, GPU job 27: does the plane stride change the rate of prediction's access pattern?  (tools/ubench/pred_mix.hip s)
O=gpurun_out/r4y; mkdir -p $O
timeout -k 10 300 hevcasm_amd/lib_ab/pred_mix s > $O/pred_mix_stride.txt 2>&1; echo "rc=$?"
echo job27 done
