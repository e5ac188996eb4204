#!/bin/bash
# round 6: (1) the intra frame cases still equal the oracle after the ILP / DPP changes; (2) one 1080p frame, per partition, library proper against five knock-out builds
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout -k 10 300 python tools/jobs/r06_intra_first.py > gpurun_out/r06_intra_parity2.log 2>&1 || { tail -20 gpurun_out/r06_intra_parity2.log; exit 1; }
tail -1 gpurun_out/r06_intra_parity2.log
python tools/ab.py --libs hevcasm_amd/lib/libhevcasm_hip.so hevcasm_amd/lib_ab/knock_NB/libhevcasm_hip.so hevcasm_amd/lib_ab/knock_PRED/libhevcasm_hip.so hevcasm_amd/lib_ab/knock_FWD/libhevcasm_hip.so \
  hevcasm_amd/lib_ab/knock_QUANT/libhevcasm_hip.so hevcasm_amd/lib_ab/knock_INV/libhevcasm_hip.so --rounds 1 -- python tools/bench_intra_frames.py --frames 1 --reps 3 --maps all_4x4,all_8x8,all_16x16,all_32x32,mixed_quadtree
