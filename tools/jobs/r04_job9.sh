#!/bin/bash
# round 4, GPU job 9: scalar descriptor loads in block_cost_kernel -- parity, then A/B against the vector-descriptor build (both with the strip kernel off), then the strip kernel on top
O=gpurun_out/r4i; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "sad or ssd or golden or fuzz or footprint or hip_equals or abi" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
grep -q "rc=0" $O/pytest.log || { echo "parity failed: no timing"; exit 1; }
HEVCASM_SAD4_NOSTRIP=1 python3 tools/ab.py --libs hevcasm_amd/lib_ab/vecdesc/libhevcasm_hip.so hevcasm_amd/lib/libhevcasm_hip.so --rounds 3 -- python3 tools/bench_sad.py > $O/scalar_desc_ab.txt 2>&1
python3 tools/ab_env.py --env HEVCASM_SAD4_NOSTRIP=1 - --rounds 3 -- python3 tools/bench_sad.py --ways 4 16x32 16x16 16x12 16x8 8x32 8x16 8x8 > $O/sad4_strip_ab.txt 2>&1
echo job9 done
