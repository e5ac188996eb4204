#!/bin/bash
# round 4, GPU job 31: the library with pred_hfirst_bi8_kernel adopted for 16- / 32-wide PUs -- whole GPU suite, smoke, then A/B of the adopted selection against the stream kernel
O=gpurun_out/r4ab; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
grep -q "rc=0" $O/pytest.log || { echo "suite failed"; exit 1; }
python3 -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
python3 tools/ab_env.py --env HEVCASM_PRED_NOHFIRST_BI=1 - --rounds 4 -- python3 tools/bench_families.py pred8tap > $O/hfirst_bi_adopted_ab.txt 2>&1
echo job31 done
