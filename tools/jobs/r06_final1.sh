#!/bin/bash
# round 6, final library: the GPU suite, smoke(), the frame pipeline example (1080p and 4K: the figures DESIGN / INTEGRATION quote), the self-test with its HIP column
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r06_gputest_final.log 2>&1; echo "pytest rc=$?"; tail -18 gpurun_out/r06_gputest_final.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r06_smoke.log
(timeout -k 10 200 python examples/frame_pipeline.py; timeout -k 10 200 python examples/frame_pipeline.py 3840 2176 16 16) > gpurun_out/r06_frame_pipeline.txt 2>&1; echo "pipeline rc=$?"; cat gpurun_out/r06_frame_pipeline.txt | grep -v amdgpu.ids
HEVCASM_ENABLE_HIP=1 timeout -k 10 300 hevcasm_amd/lib/hevcasm > gpurun_out/r06_selftest_mi355x.txt 2>&1; echo "selftest rc=$?"; tail -1 gpurun_out/r06_selftest_mi355x.txt
gcc -std=gnu11 -Iinclude examples/inter_chain.c -Lhevcasm_amd/lib -lhevcasm_hip -Wl,-rpath,$PWD/hevcasm_amd/lib -o /tmp/inter_chain && /tmp/inter_chain
gcc -std=gnu11 -Iinclude examples/tu_chain.c -Lhevcasm_amd/lib -lhevcasm_hip -Wl,-rpath,$PWD/hevcasm_amd/lib -o /tmp/tu_chain && /tmp/tu_chain
