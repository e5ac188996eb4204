#!/bin/bash
# round 4, GPU job 21: wall-clock of the driver's commands
O=gpurun_out/r4u; mkdir -p $O
t0=$SECONDS; python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench.py --gpus 1 --steps 20 --warmup 5: rc=$? $((SECONDS - t0)) s" | tee $O/wall.txt
t0=$SECONDS; python3 bench.py > $O/bench_noflags.json 2> $O/bench_noflags.err; echo "bench.py (no flags): rc=$? $((SECONDS - t0)) s" | tee -a $O/wall.txt
echo job21 done
