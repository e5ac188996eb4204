#!/bin/bash
# round 6: one launch for the block costs of a frame's partition -- tests (incl. the existing SAD parity, since the block kernel's body moved), then the latency rows
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout -k 10 600 python -m pytest tests/test_cost_runs.py tests/test_hip_parity.py -x -q -m gpu -k "cost or sad or ssd" > gpurun_out/r06_costruns_tests.log 2>&1 || { tail -30 gpurun_out/r06_costruns_tests.log; exit 1; }
tail -2 gpurun_out/r06_costruns_tests.log
timeout -k 10 300 python tools/bench_families.py costlat > gpurun_out/r06_costlat.jsonl 2> gpurun_out/r06_costlat.err || { tail -20 gpurun_out/r06_costlat.err; exit 1; }
cut -c1-170 gpurun_out/r06_costlat.jsonl
