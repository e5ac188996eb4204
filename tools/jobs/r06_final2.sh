#!/bin/bash
# round 6, final library: rocprofv3 kernel-trace stats + the PMC passes behind roofline.traffic (tools/profile_round.sh; traffic.json carries the library's sha256),
# then the driver's bench command with that traffic.json in place, and the me5 workload at 1 and 8 frames per launch
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
bash tools/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1; tail -1 gpurun_out/r06_profile_round.log
main=$(ls -S gpurun_out/prof_r06/kt_full/*/*_kernel_stats.csv | head -1)
python3 tools/summarize_prof.py stats $main > gpurun_out/r06_bench_kernel_stats.txt 2>&1; head -6 gpurun_out/r06_bench_kernel_stats.txt | cut -c1-150
cp gpurun_out/prof_r06/traffic.json profiles/traffic.json
timeout -k 10 600 python bench.py --full-out gpurun_out/r06_bench_full_record.json > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err; echo "bench rc=$?"; cat gpurun_out/r06_bench_default.json
for F in 1 8; do
  timeout -k 10 400 python bench.py --workload me5 --steps 10 --warmup 2 --frames $F --cpu-seconds 6 --full-out /dev/null > gpurun_out/r06_bench_me5_F$F.json 2> gpurun_out/r06_me5_F$F.err; echo "me5 F=$F rc=$?"
done
