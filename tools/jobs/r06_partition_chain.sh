#!/bin/bash
# round 6: the frame chain on a mixed partition -- tests, then the example (eager / graph timings)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout -k 10 600 python -m pytest tests/test_frame_pipeline.py tests/test_cost_runs.py -x -q -m gpu > gpurun_out/r06_partition_tests.log 2>&1 || { tail -30 gpurun_out/r06_partition_tests.log; exit 1; }
tail -2 gpurun_out/r06_partition_tests.log
timeout -k 10 300 python examples/frame_pipeline.py > gpurun_out/r06_frame_pipeline.txt 2> gpurun_out/r06_frame_pipeline.err || { tail -20 gpurun_out/r06_frame_pipeline.err; exit 1; }
cat gpurun_out/r06_frame_pipeline.txt
