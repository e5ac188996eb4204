#!/bin/bash
# round 5, job 2: parity of the two transform A/B builds, then their timing against the shipped library; frame SSE rows; narrowing cases on the GPU
set -e
for v in wide pad; do
  HEVCASM_HIP_LIB=$PWD/hevcasm_amd/lib_ab/$v/libhevcasm_hip.so python -m pytest tests/test_hip_parity.py tests/test_golden.py -q -x -m gpu -k "inverse or forward or dequant or config4 or roundtrip or in_place or golden or bottom_up" > gpurun_out/r05_job2_parity_$v.log 2>&1 || { tail -30 gpurun_out/r05_job2_parity_$v.log; exit 1; }
  tail -2 gpurun_out/r05_job2_parity_$v.log
done
python tools/ab.py --libs hevcasm_amd/lib/libhevcasm_hip.so hevcasm_amd/lib_ab/wide/libhevcasm_hip.so hevcasm_amd/lib_ab/pad/libhevcasm_hip.so --rounds 3 -- python tools/bench_transform_large.py 8 16 32 > gpurun_out/r05_transform_wide_pad_ab.txt 2>&1
cat gpurun_out/r05_transform_wide_pad_ab.txt
python tools/bench_families.py ssdlin > gpurun_out/r05_ssdlin.txt 2>&1
cat gpurun_out/r05_ssdlin.txt
python -m pytest tests/test_hip_parity.py -q -x -m gpu -k "narrowing or frame_sse" 2>&1 | tail -3
