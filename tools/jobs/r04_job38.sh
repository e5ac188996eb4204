#!/bin/bash
# round 4, GPU job 38: pred_hfirst_bi8_kernel with 4 rows in flight behind the first 8 (96 registers, 5 waves per SIMD at 16-wide) against 8 (112 registers, 4 waves)
O=gpurun_out/r4ai; mkdir -p $O
L0=hevcasm_amd/lib/libhevcasm_hip.so; L1=hevcasm_amd/lib_ab/ahead4/libhevcasm_hip.so
HEVCASM_HIP_LIB=$PWD/$L1 HEVCASM_PRED_HFIRST_BI_ALL=1 timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "pred or footprint or golden or hip_equals" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
grep -q "rc=0" $O/pytest.log || { echo "parity failed: no timing"; exit 1; }
python3 tools/ab.py --libs $L0 $L1 --rounds 4 -- python3 tools/bench_families.py pred8sub64 > $O/hfirst_bi_ahead_ab.txt 2>&1
echo job38 done
