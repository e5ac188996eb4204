#!/bin/bash
# round 4, GPU job 41: pred_hfirst_bi8_kernel on 64-wide PUs at 256 and 64 frames per launch (blocked lane deal), against the stream kernel
O=gpurun_out/r4al; mkdir -p $O
python3 tools/ab_env.py --env - HEVCASM_PRED_HFIRST_BI_ALL=1 --rounds 3 -- python3 tools/bench_families.py pred8wide256 > $O/hfirst_bi_wide_256.txt 2>&1
python3 tools/ab_env.py --env - HEVCASM_PRED_HFIRST_BI_ALL=1 --rounds 3 -- python3 tools/bench_families.py pred8wide64 > $O/hfirst_bi_wide_64.txt 2>&1
echo job41 done
