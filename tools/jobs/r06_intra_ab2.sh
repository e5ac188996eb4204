#!/bin/bash
# A/B of an intra-kernel build against the shipped library: tools/jobs/r06_intra_ab2.sh <dir under hevcasm_amd/lib_ab>; parity first, then alternating timings
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
fmt='
import sys, json
for l in sys.stdin:
    r = json.loads(l); print("%-16s F=%-3d ms=%8.3f" % (r["map"], r["frames"], r["ms"]))
'
HEVCASM_HIP_LIB=$R/hevcasm_amd/lib_ab/$1/libhevcasm_hip.so timeout -k 10 300 python tools/jobs/r06_intra_first.py 2>&1 | tail -1
for round in 1 2; do
  for lib in lib lib_ab/$1; do
    echo "== $lib"
    HEVCASM_HIP_LIB=$R/hevcasm_amd/$lib/libhevcasm_hip.so timeout -k 10 300 python tools/bench_intra_frames.py --frames 1,64,256 --maps mixed_quadtree,all_32x32,all_4x4 --reps 3 | python -c "$fmt"
  done
done
