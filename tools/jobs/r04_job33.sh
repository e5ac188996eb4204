#!/bin/bash
# round 4, GPU job 33: pred_hfirst_bi8_kernel with word-aligned loads -- parity with the kernel selected for EVERY size it can take, then A/B: stream kernel / byte-aligned loads / word-aligned loads
O=gpurun_out/r4ad; mkdir -p $O
HEVCASM_PRED_HFIRST_BI_ALL=1 timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "pred or footprint or golden or fuzz or hip_equals or subpel or abi or selftest or bottom_up or 4gib" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
grep -q "rc=0" $O/pytest.log || { echo "parity failed: no timing"; exit 1; }
python3 tools/ab_env.py --env HEVCASM_PRED_NOHFIRST_BI=1 HEVCASM_PRED_HFIRST_UNALIGNED=1,HEVCASM_PRED_HFIRST_BI_ALL=1 HEVCASM_PRED_HFIRST_BI_ALL=1 --rounds 3 -- python3 tools/bench_families.py pred8tap > $O/hfirst_bi_aligned_ab.txt 2>&1
echo job33 done
