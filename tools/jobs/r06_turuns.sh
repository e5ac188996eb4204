#!/bin/bash
# round 6: one launch per transform stage for a frame's TUs -- tests (incl. the existing transform / quantiser parity: the kernels' bodies moved into .inc files), then the latency rows
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout -k 10 600 python -m pytest tests/test_tu_runs.py tests/test_frame_pipeline.py tests/test_hip_parity.py tests/test_golden.py -x -q -m gpu > gpurun_out/r06_turuns_tests.log 2>&1 || { tail -40 gpurun_out/r06_turuns_tests.log; exit 1; }
tail -2 gpurun_out/r06_turuns_tests.log
timeout -k 10 300 python tools/bench_families.py tulat > gpurun_out/r06_tulat.jsonl 2> gpurun_out/r06_tulat.err || { tail -20 gpurun_out/r06_tulat.err; exit 1; }
cut -c1-200 gpurun_out/r06_tulat.jsonl
