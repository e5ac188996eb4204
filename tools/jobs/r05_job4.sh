#!/bin/bash
# round 5, job 4: intra modes kernel, second form: parity, then the rows
set -e
python -m pytest tests/test_intra_modes.py tests/test_hip_parity.py -q -x -m gpu -k "intra" 2>&1 | tail -3
python tools/bench_families.py intramodes > gpurun_out/r05_intramodes2.txt 2>&1
cut -c1-175 gpurun_out/r05_intramodes2.txt | grep kernel
