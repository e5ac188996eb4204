#!/bin/bash
# round 4, GPU job 36: the library as committed (pred_hfirst_bi8_kernel adopted) -- whole GPU suite, __graft_entry__.smoke(), the driver's bench command, the me5 line, every family
O=gpurun_out/r4ag; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
grep -q "rc=0" $O/pytest.log || { echo "suite failed"; exit 1; }
python3 -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python3 bench.py --workload me5 --steps 20 --warmup 3 > $O/bench_me5.json 2> $O/bench_me5.err; echo "me5 rc=$?"
python3 tools/bench_families.py > $O/families_final.jsonl 2> $O/families_final.err; echo "families rc=$?"
echo job36 done
