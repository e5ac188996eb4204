#!/bin/bash
# round 4, GPU job 4: the whole GPU suite on the library with the bi-prediction wrap fix, the transform access-mix probe, a second A/B of the 16x16 / 32x32 transforms
O=gpurun_out/r4d; mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -Wno-unused-value -o /tmp/transform_mix tools/ubench/transform_mix.hip > $O/transform_mix.txt 2>&1 && /tmp/transform_mix >> $O/transform_mix.txt 2>&1
hevcasm_amd/lib/ubench_qsad_rate live > $O/qsad_live.txt 2>&1
python3 tools/ab.py --libs hevcasm_amd/lib_ab/r4base/libhevcasm_hip.so hevcasm_amd/lib/libhevcasm_hip.so --rounds 4 -- python3 tools/bench_transform_large.py 16 32 > $O/transform_ab_16_32.txt 2>&1
python3 tools/ab.py --libs hevcasm_amd/lib_ab/r4base/libhevcasm_hip.so hevcasm_amd/lib/libhevcasm_hip.so --rounds 2 -- python3 tools/bench_families.py predq > $O/pred_ab.txt 2>&1
echo job4 done
