#!/bin/bash
# round 5, job 7: reconstruct with K groups per workgroup (K = 4 default build; 1 = the round-4 form; 2; 8): parity, then A/B
set -e
python -m pytest tests/test_hip_parity.py tests/test_golden.py tests/test_hip_fuzz.py -q -x -m gpu -k "quant or reconstruct or recon or golden or in_place or transforms_and or bottom_up or window" 2>&1 | tail -2
for v in rc1 rc2 rc8; do HEVCASM_HIP_LIB=$PWD/hevcasm_amd/lib_ab/$v/libhevcasm_hip.so python -m pytest tests/test_hip_parity.py tests/test_golden.py -q -x -m gpu -k "quant or recon or golden or in_place" 2>&1 | tail -1; done
python tools/ab.py --libs hevcasm_amd/lib_ab/rc1/libhevcasm_hip.so hevcasm_amd/lib_ab/rc2/libhevcasm_hip.so hevcasm_amd/lib/libhevcasm_hip.so hevcasm_amd/lib_ab/rc8/libhevcasm_hip.so --rounds 3 -- python tools/bench_families.py quant > gpurun_out/r05_reconstruct_unroll_ab.txt 2>&1
grep -v "^# round" gpurun_out/r05_reconstruct_unroll_ab.txt
