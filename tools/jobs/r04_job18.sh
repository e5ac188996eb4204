#!/bin/bash
# round 4, GPU job 18: chroma bi strip kernel held to 168 registers (3 waves per SIMD, 40 bytes of scratch) against the default (177 registers, 2 waves); chroma sizes, 3 rounds
O=gpurun_out/r4r; mkdir -p $O
python3 tools/ab.py --libs hevcasm_amd/lib/libhevcasm_hip.so hevcasm_amd/lib_ab/strip3w/libhevcasm_hip.so --rounds 3 -- python3 tools/bench_families.py pred4tap > $O/strip_bi_3waves_ab.txt 2>&1
echo job18 done
