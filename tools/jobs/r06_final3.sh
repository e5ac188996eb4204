#!/bin/bash
# round 6, final library: the driver's bench command once more (traffic.json of this library in place), after bench.py's pattern rows became the faster of two child runs
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout -k 10 600 python bench.py --full-out gpurun_out/r06_bench_full_record.json > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err; echo "bench rc=$?"; cat gpurun_out/r06_bench_default.json
