#!/bin/bash
# round 4, GPU job 42: the library as committed at the end of the round, built from clean -- whole GPU suite and __graft_entry__.smoke()
O=gpurun_out/r4am; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
python3 -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
python3 bench.py > $O/bench_noflags.json 2> $O/bench_noflags.err; echo "bench (no flags) rc=$? in $SECONDS s"
echo job42 done
