#!/bin/bash
# round 4, GPU job 3: transform A/B (round-4 VALU trims against the library as committed before them), the new prediction test, parity of the transform families
O=gpurun_out/r4c; mkdir -p $O
python3 -m pytest tests/test_hip_parity.py tests/test_golden.py tests/test_hip_fuzz.py -m gpu -x -q -k "pred_one_1080p or inverse or transform or dequant or golden or fuzz or bi" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log

python3 tools/ab.py --libs hevcasm_amd/lib_ab/r4base/libhevcasm_hip.so hevcasm_amd/lib/libhevcasm_hip.so --rounds 3 -- python3 tools/bench_transform_large.py > $O/transform_ab_large.txt 2>&1
echo job3 done
