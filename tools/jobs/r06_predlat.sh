#!/bin/bash
# round 6: one launch for a frame's partition -- tests, then the latency rows with and without device-memory kernel arguments
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout -k 10 500 python -m pytest tests/test_pred_runs.py -x -q -m gpu > gpurun_out/r06_predruns_tests.log 2>&1 || { tail -20 gpurun_out/r06_predruns_tests.log; exit 1; }
tail -2 gpurun_out/r06_predruns_tests.log
for round in 1 2; do
  timeout -k 10 200 python tools/bench_families.py predlat 2> gpurun_out/r06_predlat.err | grep "partition, one\|frame graph\|tap\"" | cut -c1-120
  HIP_FORCE_DEV_KERNARG=1 timeout -k 10 200 python tools/bench_families.py predlat 2>> gpurun_out/r06_predlat.err | grep "partition, one\|frame graph\|tap\"" | cut -c1-120 | sed 's/^/DEVKERNARG /'
done
