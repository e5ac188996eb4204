#!/bin/bash
# round 6: the driver's three commands on one box -- the GPU suite, smoke(), the default bench -- plus the self-test executable with its HIP column
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputest1.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06_gputest1.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r06_smoke.log
timeout -k 10 600 python bench.py > gpurun_out/r06_bench1.json 2> gpurun_out/r06_bench1.err; echo "bench rc=$?"; cat gpurun_out/r06_bench1.json
HEVCASM_ENABLE_HIP=1 timeout -k 10 300 hevcasm_amd/lib/hevcasm > gpurun_out/r06_selftest_mi355x.txt 2>&1; echo "selftest rc=$?"; tail -2 gpurun_out/r06_selftest_mi355x.txt
