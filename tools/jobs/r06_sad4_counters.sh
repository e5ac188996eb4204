#!/bin/bash
# round 6: the counter passes of the NARROW 4-way SAD kernels re-taken on the final round-6 library (tools/pmc.sh: one group per run, no tracing)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
bash tools/pmc.sh sad4_pmc sad 16 16 4 + sad 8 8 4 + sad 32 32 4 + sad 32 64 4 + sad 12 16 4 + sad 8 32 4 > gpurun_out/r06_sad4_counters_raw.txt 2>&1
sha256sum hevcasm_amd/lib/libhevcasm_hip.so | cut -c1-16
grep -c dispatches gpurun_out/r06_sad4_counters_raw.txt
