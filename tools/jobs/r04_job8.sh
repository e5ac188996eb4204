#!/bin/bash
# round 4, GPU job 8: sad4_strip_kernel second form (DPP reductions; one-wave workgroups against four) -- parity, then A/B
O=gpurun_out/r4h; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "sad or golden or fuzz or footprint or hip_equals" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
grep -q "rc=0" $O/pytest.log || { echo "parity failed: no timing"; exit 1; }
python3 tools/ab_env.py --env HEVCASM_SAD4_NOSTRIP=1 - --rounds 3 -- python3 tools/bench_sad.py --ways 4 16x32 16x16 16x12 16x8 8x32 8x16 8x8 > $O/sad4_strip_ab.txt 2>&1
HEVCASM_HIP_LIB=$PWD/hevcasm_amd/lib_ab/strip4w/libhevcasm_hip.so python3 tools/bench_sad.py --ways 4 16x32 16x16 16x8 8x16 8x8 > $O/sad4_strip_4waves.txt 2>&1
python3 tools/ab_env.py --env HEVCASM_SAD4_NOSTRIP=1 - --rounds 2 -- python3 tools/bench_sad.py --ways 4 --far 40 16x16 8x8 16x32 > $O/sad4_strip_far_ab.txt 2>&1
echo job8 done
