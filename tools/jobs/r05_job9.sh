#!/bin/bash
# round 5, job 9: intra modes kernel, third form (one-shot workgroups, ref[] straight from global loads) against the second (lib_ab/old): parity, then A/B
set -e
python -m pytest tests/test_intra_modes.py tests/test_hip_fuzz.py tests/test_hip_parity.py -q -x -m gpu -k "intra or small_families" 2>&1 | tail -2
python tools/ab.py --libs hevcasm_amd/lib_ab/old/libhevcasm_hip.so hevcasm_amd/lib/libhevcasm_hip.so --rounds 3 -- python tools/bench_families.py intramodes > gpurun_out/r05_intra_third_form_ab.txt 2>&1
grep -v "^# round" gpurun_out/r05_intra_third_form_ab.txt
