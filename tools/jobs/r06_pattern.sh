#!/bin/bash
# round 6: the prediction floor = the shipped kernels without arithmetic (lib/pattern), same box, alternating with the real library; then the GPU prediction tests on the real one
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for round in 1 2 3; do
  echo "== real $round"; timeout -k 10 200 python tools/bench_families.py pred16 2>> gpurun_out/r06_pattern.err | cut -c1-200
  echo "== pattern $round"; HEVCASM_HIP_LIB=$R/hevcasm_amd/lib/pattern/libhevcasm_hip.so timeout -k 10 200 python tools/bench_families.py pred16 2>> gpurun_out/r06_pattern.err | cut -c1-200
done
