#!/bin/bash
# round 4, GPU job 30: pred_hfirst_bi8_kernel -- parity with it selected for every size, then the word-aligned loads against the byte-aligned ones
O=gpurun_out/r4aa; mkdir -p $O
HEVCASM_PRED_HFIRST_BI=1 timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "pred or footprint or golden or fuzz or hip_equals or subpel or abi or selftest or bottom_up or 4gib" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
grep -q "rc=0" $O/pytest.log || { echo "parity failed: no timing"; exit 1; }
python3 tools/ab_env.py --env - HEVCASM_PRED_HFIRST_BI=1 --rounds 3 -- python3 tools/bench_families.py pred8tap > $O/hfirst_bi_ab.txt 2>&1
echo job30 done
