#!/bin/bash
# round 5, job 3: frame SSE after the one-atomic-per-workgroup rewrite: parity, then the rows
set -e
python -m pytest tests/test_hip_parity.py -q -x -m gpu -k "frame_sse" 2>&1 | tail -3
python tools/bench_families.py ssdlin > gpurun_out/r05_ssdlin2.txt 2>&1
grep frame_sse gpurun_out/r05_ssdlin2.txt
