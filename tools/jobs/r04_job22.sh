#!/bin/bash
# round 4, GPU job 22: pred_hfirst_uni8_kernel -- parity (every prediction test, footprints at guard pages, fuzz, golden), then same-binary A/B against pred_stream_kernel
O=gpurun_out/r4v; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "pred or footprint or golden or fuzz or hip_equals or subpel or abi or selftest" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
grep -q "rc=0" $O/pytest.log || { echo "parity failed: no timing"; exit 1; }
python3 tools/ab_env.py --env HEVCASM_PRED_NOHFIRST=1 - --rounds 3 -- python3 tools/bench_families.py pred8tap > $O/hfirst_ab.txt 2>&1
echo job22 done
