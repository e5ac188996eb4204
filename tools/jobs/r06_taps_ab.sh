#!/bin/bash
# round 6: chroma taps from literals by arithmetic (no table load behind the fraction byte): parity of the prediction families, then same-box A/B against the library before the change
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_golden.py tests/test_hip_fuzz.py tests/test_footprint.py tests/test_pred_runs.py -x -q -m gpu -k "pred or golden or fuzz or footprint or narrowing" > gpurun_out/r06_taps_tests.log 2>&1 || { tail -20 gpurun_out/r06_taps_tests.log; exit 1; }
tail -2 gpurun_out/r06_taps_tests.log
python tools/ab.py --libs hevcasm_amd/lib_ab/before_taps/libhevcasm_hip.so hevcasm_amd/lib/libhevcasm_hip.so --rounds 3 -- python tools/bench_families.py pred4tap
