#!/bin/bash
# round 4, GPU job 11: whole GPU suite; transform offsets by scalar loads (A/B by build); SAD scalar descriptors up to 32 blocks per wave (A/B by build)
O=gpurun_out/r4k; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
grep -q "rc=0" $O/pytest.log || { echo "parity failed: no timing"; exit 1; }
python3 tools/ab.py --libs hevcasm_amd/lib_ab/trvec/libhevcasm_hip.so hevcasm_amd/lib/libhevcasm_hip.so --rounds 4 -- python3 tools/bench_transform_large.py 16 32 > $O/transform_scalar_off_ab.txt 2>&1
python3 tools/ab.py --libs hevcasm_amd/lib/libhevcasm_hip.so hevcasm_amd/lib_ab/desc32/libhevcasm_hip.so --rounds 4 -- python3 tools/bench_sad.py 8x8 8x16 16x8 12x16 > $O/sad_desc32_ab.txt 2>&1
echo job11 done
