#!/bin/bash
# round 6: the fused residual loop (hevcasm_hip_transform_quantize_reconstruct_runs) -- tests, then the latency / throughput rows and the per-size table
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout -k 10 600 python -m pytest tests/test_tu_runs.py tests/test_frame_pipeline.py tests/test_abi.py -x -q -m gpu > gpurun_out/r06_tufused_tests.log 2>&1 || { tail -40 gpurun_out/r06_tufused_tests.log; exit 1; }
tail -2 gpurun_out/r06_tufused_tests.log
timeout -k 10 300 python tools/bench_families.py tulat > gpurun_out/r06_tulat.jsonl 2> gpurun_out/r06_tulat.err || { tail -20 gpurun_out/r06_tulat.err; exit 1; }
cut -c1-150 gpurun_out/r06_tulat.jsonl | tail -8
timeout -k 10 300 python tools/bench_families.py tufusedsizes > gpurun_out/r06_tufused_sizes.jsonl 2> gpurun_out/r06_tufused_sizes.err || { tail -20 gpurun_out/r06_tufused_sizes.err; exit 1; }
cut -c1-260 gpurun_out/r06_tufused_sizes.jsonl
