#!/bin/bash
# round 6: F frames per launch for config #5 -- the GPU test, then the me5 bench line at F = 1 and F = 8 (one GPU: also what the slowest of 8 ranks would take)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout -k 10 600 python -m pytest tests/test_shard.py -x -q -m gpu > gpurun_out/r06_me5_tests.log 2>&1 || { tail -30 gpurun_out/r06_me5_tests.log; exit 1; }
tail -2 gpurun_out/r06_me5_tests.log
for F in 1 8; do
  timeout -k 10 400 python bench.py --workload me5 --steps 10 --warmup 2 --frames $F --cpu-seconds 6 --full-out gpurun_out/r06_me5_F$F.full.json > gpurun_out/r06_me5_F$F.json 2> gpurun_out/r06_me5_F$F.err || tail -5 gpurun_out/r06_me5_F$F.err
  cat gpurun_out/r06_me5_F$F.json
done
