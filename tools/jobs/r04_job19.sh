#!/bin/bash
# round 4, GPU job 19: luma uni stream kernel held to 96 registers (5 waves per SIMD, 20 bytes of scratch) against 99 (4 waves); luma sizes
O=gpurun_out/r4s; mkdir -p $O
python3 tools/ab.py --libs hevcasm_amd/lib/libhevcasm_hip.so hevcasm_amd/lib_ab/uni5w/libhevcasm_hip.so --rounds 3 -- python3 tools/bench_families.py pred8tap > $O/stream_uni_5waves_ab.txt 2>&1
echo job19 done
