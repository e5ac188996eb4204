#!/bin/bash
# round 4, GPU job 43: bench.py with the access-pattern probes' figures beside the prediction and inverse-transform rows (<key>_pattern_ms, <key>_vs_pattern)
O=gpurun_out/r4an; mkdir -p $O
hevcasm_amd/lib/ubench_pred_mix live > $O/pred_mix_live.txt 2>&1; echo "probe rc=$?"; cat $O/pred_mix_live.txt
hevcasm_amd/lib/ubench_transform_mix live > $O/transform_mix_live.txt 2>&1; echo "probe rc=$?"; cat $O/transform_mix_live.txt
python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? in $SECONDS s"
python3 -m pytest tests/test_bench_plumbing.py -x -q 2>&1 | tail -2
echo job43 done
