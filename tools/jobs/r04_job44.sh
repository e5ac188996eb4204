#!/bin/bash
# round 4, GPU job 44: pred_hfirst_bi8_kernel's blocked lane deal for 24- and 48-wide PUs (partner by ds_bpermute) -- parity with the kernel selected for every size, then A/B
O=gpurun_out/r4ao; mkdir -p $O
HEVCASM_PRED_HFIRST_BI_ALL=1 timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "pred or footprint or golden or hip_equals or fuzz" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
grep -q "rc=0" $O/pytest.log || { echo "parity failed: no timing"; exit 1; }
python3 tools/ab_env.py --env HEVCASM_PRED_NOHFIRST_BI=1 HEVCASM_PRED_HFIRST_INTERLEAVED=1,HEVCASM_PRED_HFIRST_BI_ALL=1 HEVCASM_PRED_HFIRST_BI_ALL=1 --rounds 3 -- python3 tools/bench_families.py pred8odd > $O/hfirst_bi_odd_ab.txt 2>&1
echo job44 done
