#!/bin/bash
# round 4, GPU job 6: the library as it stands -- whole GPU suite, the driver's bench command, the me5 line, multi-rank rehearsals on one GPU (gloo), every family, the self-test
O=gpurun_out/r4f; mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python3 bench.py --workload me5 --steps 20 --warmup 3 > $O/bench_me5.json 2> $O/bench_me5.err; echo "me5 rc=$?"
for n in 2 4; do
  HEVCASM_BENCH_ONE_DEVICE=1 HEVCASM_DIST_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n bench.py --gpus $n --steps 3 --warmup 1 --blocks 65536 > $O/rehearsal_sad_x$n.json 2> $O/rehearsal_sad_x$n.err; echo "rehearsal sad x$n rc=$?"
  HEVCASM_BENCH_ONE_DEVICE=1 HEVCASM_DIST_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --workload me5 --gpus $n --steps 5 --warmup 1 > $O/rehearsal_me5_x$n.json 2> $O/rehearsal_me5_x$n.err; echo "rehearsal me5 x$n rc=$?"
done
python3 tools/bench_families.py > $O/families_final.jsonl 2> $O/families_final.err; echo "families rc=$?"
hevcasm_amd/lib/hevcasm > $O/selftest.txt 2>&1; echo "selftest rc=$?"
echo job6 done
