#!/bin/bash
# round 4, GPU job 2: the GPU suite after the launcher / wrapper changes, the default bench line (new scalar record) and the me5 line
O=gpurun_out/r4b; mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
python3 bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python3 bench.py --workload me5 --steps 20 --warmup 3 > $O/bench_me5.json 2> $O/bench_me5.err; echo "me5 rc=$?"
echo job2 done
