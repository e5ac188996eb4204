#!/bin/bash
# round 4, GPU job 24: is the H-first kernel's loss a matter of launch size?  Same box, the shipped library against the H-first build (commit cf83a19),
# luma 16x16 ... 64x64 at 64 and at 256 frames per launch
O=gpurun_out/r4x; mkdir -p $O
L0=hevcasm_amd/lib/libhevcasm_hip.so; L1=hevcasm_amd/lib_ab/hfirst/libhevcasm_hip.so
python3 tools/ab.py --libs $L0 $L1 --rounds 3 -- python3 tools/bench_families.py pred8sub256 > $O/hfirst_256.txt 2>&1
python3 tools/ab.py --libs $L0 $L1 --rounds 3 -- python3 tools/bench_families.py pred8sub64 > $O/hfirst_64.txt 2>&1
echo job24 done
