#!/bin/bash
# round 6: the rocprofv3 evidence on the FINAL library -- kernel-trace stats of the bench command, the FETCH_SIZE / WRITE_SIZE passes behind roofline.traffic
# (traffic.json carries the library's sha256), then the counter passes of the four headline prediction rows on the library proper and on the pattern-only build
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
bash tools/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1; tail -2 gpurun_out/r06_profile_round.log
python3 tools/summarize_prof.py stats gpurun_out/prof_r06/kt_full/*/*_kernel_stats.csv > gpurun_out/r06_bench_kernel_stats.txt 2>&1; head -12 gpurun_out/r06_bench_kernel_stats.txt
cat gpurun_out/prof_r06/traffic.json | head -30
bash tools/pmc2.sh r06_pred_pmc_real tools/bench_families.py pred16 > gpurun_out/r06_pred_pmc_real.txt 2>&1
HEVCASM_HIP_LIB=$R/hevcasm_amd/lib/pattern/libhevcasm_hip.so bash tools/pmc2.sh r06_pred_pmc_pattern tools/bench_families.py pred16 > gpurun_out/r06_pred_pmc_pattern.txt 2>&1
grep -c "" gpurun_out/r06_pred_pmc_real.txt gpurun_out/r06_pred_pmc_pattern.txt
