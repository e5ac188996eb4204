#!/bin/bash
# builds hevcasm_amd/lib_ab/<variant>/libhevcasm_hip.so: the library with intra_frame.hip compiled under one knock-out switch each (timing experiments: which step of
# a TU costs what; the results of such a build are wrong by design).  Run here (hipcc cross-compiles), the libraries travel with gpurun.
set -e
cd "$(dirname "$0")/../../hevcasm_amd/csrc"
for v in NB PRED FWD QUANT INV; do
  mkdir -p build_ab ../lib_ab/knock_$v
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -Wall -Wno-unused-result -Wno-unused-variable -Wno-unused-but-set-variable -I../../include -Ihost -DIF_KNOCK_$v -c hip/intra_frame.hip -o build_ab/hip_intra_frame_$v.o &
done
wait
for v in NB PRED FWD QUANT INV; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_ab/knock_$v/libhevcasm_hip.so $(ls build/host_*.o build/hip_*.o | grep -v "hip_intra_frame.o\|pattern") build_ab/hip_intra_frame_$v.o
done
ls -la ../lib_ab/*/libhevcasm_hip.so
