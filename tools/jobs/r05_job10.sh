#!/bin/bash
# round 5, job 10: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) of the round's new kernels
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd $R
mkdir -p $R/gpurun_out/pmc_r05_new
for cfg in "frame_sse 3840 2160 32" "intra 2" "intra 2 compact" "intra 4" "intra 5" "intra_dc 5"; do
  tag=$(echo $cfg | tr ' ' '_')
  for g in FETCH_SIZE WRITE_SIZE; do
    d=$R/gpurun_out/pmc_r05_new/${tag}_$g
    rocprofv3 --pmc $g --output-format csv -d $d -- python3 tools/prof_one.py $cfg > $d.log 2>&1 || echo "FAILED $cfg $g"
  done
  echo "## $cfg"
  python3 tools/summarize_prof.py pmc "$cfg" $R/gpurun_out/pmc_r05_new/${tag}_*/*/*_counter_collection.csv | grep -v "torch\|at::\|elementwise\|Memset\|fill" 
done
