#!/bin/bash
# round 4, GPU job 29: the H-first luma kernel under coherent motion (one vector per 64 x 64 region), where the access pattern is cheap
O=gpurun_out/r4z; mkdir -p $O
L0=hevcasm_amd/lib/libhevcasm_hip.so; L1=hevcasm_amd/lib_ab/hfirst/libhevcasm_hip.so
python3 tools/ab.py --libs $L0 $L1 --rounds 3 -- python3 tools/bench_families.py pred8ctu64 > $O/hfirst_ctu64.txt 2>&1
echo job29 done
