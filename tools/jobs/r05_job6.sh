#!/bin/bash
# round 5, job 6: INTRA_DC with K groups per workgroup, loads first (K = 4 default build; 1 = the round-4 form; 8): parity, then A/B at 2^28 pixels
set -e
python -m pytest tests/test_hip_parity.py tests/test_golden.py tests/test_hip_fuzz.py tests/test_intra_modes.py tests/test_footprint.py -q -x -m gpu -k "intra or golden or small_families or footprint" 2>&1 | tail -2
for v in dc1 dc8; do HEVCASM_HIP_LIB=$PWD/hevcasm_amd/lib_ab/$v/libhevcasm_hip.so python -m pytest tests/test_hip_parity.py tests/test_golden.py -q -x -m gpu -k "intra or golden" 2>&1 | tail -1; done
python tools/ab.py --libs hevcasm_amd/lib_ab/dc1/libhevcasm_hip.so hevcasm_amd/lib/libhevcasm_hip.so hevcasm_amd/lib_ab/dc8/libhevcasm_hip.so --rounds 2 -- python tools/bench_families.py intra > gpurun_out/r05_intra_dc_unroll_ab.txt 2>&1
grep -v "^# round" gpurun_out/r05_intra_dc_unroll_ab.txt
