#!/bin/bash
# round 5, job 8: inverse 16x16 / 32x32 with G consecutive groups of blocks per workgroup (a block loop: fewer, longer workgroups; the verdict's "4-8 block generations"): parity, then A/B
set -e
for v in gen2 gen4 gen8; do HEVCASM_HIP_LIB=$PWD/hevcasm_amd/lib_ab/$v/libhevcasm_hip.so python -m pytest tests/test_hip_parity.py tests/test_golden.py -q -x -m gpu -k "inverse or dequant or config4 or golden or in_place" 2>&1 | tail -1; done
python tools/ab.py --libs hevcasm_amd/lib/libhevcasm_hip.so hevcasm_amd/lib_ab/gen2/libhevcasm_hip.so hevcasm_amd/lib_ab/gen4/libhevcasm_hip.so hevcasm_amd/lib_ab/gen8/libhevcasm_hip.so --rounds 3 -- python tools/bench_transform_large.py 16 32 > gpurun_out/r05_inverse_generations_ab.txt 2>&1
grep -v "^# round" gpurun_out/r05_inverse_generations_ab.txt | grep -v "^transform "
