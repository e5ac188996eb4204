#!/bin/bash
# round 4, GPU job 40: final library -- luma prediction at 256 frames per launch, the self-test, multi-rank rehearsals on one GPU (gloo), a longer soak
O=gpurun_out/r4ak; mkdir -p $O
python3 tools/bench_families.py pred8sub256 > $O/pred_luma_256f.jsonl 2> $O/pred_luma_256f.err; echo "pred256 rc=$?"
hevcasm_amd/lib/hevcasm > $O/selftest.txt 2>&1; echo "selftest rc=$?"
for n in 2 4; do
  HEVCASM_BENCH_ONE_DEVICE=1 HEVCASM_DIST_BACKEND=gloo timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n bench.py --gpus $n --steps 3 --warmup 1 --blocks 65536 > $O/rehearsal_sad_x$n.json 2> $O/rehearsal_sad_x$n.err; echo "rehearsal sad x$n rc=$?"
  HEVCASM_BENCH_ONE_DEVICE=1 HEVCASM_DIST_BACKEND=gloo timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --workload me5 --gpus $n --steps 5 --warmup 1 > $O/rehearsal_me5_x$n.json 2> $O/rehearsal_me5_x$n.err; echo "rehearsal me5 x$n rc=$?"
done
timeout -k 10 500 python3 tools/soak.py 2400 60 1 > $O/soak.txt 2>&1; echo "soak x1 rc=$?"
tail -1 $O/soak.txt
echo job40 done
