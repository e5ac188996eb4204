#!/bin/bash
# round 5, job 5: inverse transforms with wave-level hand-overs (default build) against the workgroup-barrier form (lib_ab/wgbar): parity, then A/B
set -e
python -m pytest tests/test_hip_parity.py tests/test_golden.py -q -x -m gpu -k "inverse or forward or dequant or config4 or roundtrip or in_place or golden or bottom_up" 2>&1 | tail -2
python tools/ab.py --libs hevcasm_amd/lib_ab/wgbar/libhevcasm_hip.so hevcasm_amd/lib/libhevcasm_hip.so --rounds 3 -- python tools/bench_transform_large.py 8 16 32 > gpurun_out/r05_inverse_wave_sync_ab.txt 2>&1
grep -v "^# round" gpurun_out/r05_inverse_wave_sync_ab.txt
