#!/bin/bash
# round 4, GPU job 35: pred_hfirst_bi8_kernel with the blocked lane deal (a PU's strips of reference 0, then of reference 1) -- parity with the kernel selected for every size, then A/B against the interleaved deal
O=gpurun_out/r4af; mkdir -p $O
HEVCASM_PRED_HFIRST_BI_ALL=1 timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "pred or footprint or golden or fuzz or hip_equals or subpel or abi or selftest or bottom_up or 4gib" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
grep -q "rc=0" $O/pytest.log || { echo "parity failed: no timing"; exit 1; }
python3 tools/ab_env.py --env HEVCASM_PRED_HFIRST_INTERLEAVED=1,HEVCASM_PRED_HFIRST_BI_ALL=1 HEVCASM_PRED_HFIRST_BI_ALL=1 HEVCASM_PRED_NOHFIRST_BI=1 --rounds 4 -- python3 tools/bench_families.py pred8sub64 > $O/hfirst_bi_deal_ab.txt 2>&1
echo job35 done
