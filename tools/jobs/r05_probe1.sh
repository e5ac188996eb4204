#!/bin/bash
# round 5, job 1: transform access-mix probe with coalesced deals; small-block ME parts A/B
set -e
./hevcasm_amd/lib/ubench_transform_mix > gpurun_out/r05_transform_mix_deals.txt 2>&1
cat gpurun_out/r05_transform_mix_deals.txt
python tools/ab_env.py --env - HEVCASM_ME_PARTS=1 HEVCASM_ME_PARTS=2 HEVCASM_ME_PARTS=4 --rounds 2 -- python tools/bench_me_small.py > gpurun_out/r05_me_small_parts_ab.txt 2>&1
cat gpurun_out/r05_me_small_parts_ab.txt
python tools/bench_me_shards.py > gpurun_out/r05_me_shards.txt 2>&1
cat gpurun_out/r05_me_shards.txt
