#!/bin/bash
# round 4, GPU job 34: pred_hfirst_bi8_kernel for 8-wide PUs (two 8-byte loads per row) -- parity with the kernel selected for every size, then A/B against the stream kernel on those sizes
O=gpurun_out/r4ae; mkdir -p $O
HEVCASM_PRED_HFIRST_BI_ALL=1 timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "pred or footprint or golden or fuzz or hip_equals or subpel or abi or selftest or bottom_up or 4gib" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
grep -q "rc=0" $O/pytest.log || { echo "parity failed: no timing"; exit 1; }
python3 tools/ab_env.py --env HEVCASM_PRED_HFIRST_BI_NO8=1 HEVCASM_PRED_HFIRST_BI_ALL=1 --rounds 4 -- python3 tools/bench_families.py pred8w8 > $O/hfirst_bi_w8_ab.txt 2>&1
echo job34 done
