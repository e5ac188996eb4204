#!/bin/bash
# round 4, GPU job 17: prediction descriptors by scalar loads (8-byte lanes) -- parity, then A/B by build over every luma PU size and the chroma sizes
O=gpurun_out/r4q; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "pred or footprint or golden or fuzz or hip_equals or subpel" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
grep -q "rc=0" $O/pytest.log || { echo "parity failed: no timing"; exit 1; }
python3 tools/ab.py --libs hevcasm_amd/lib_ab/predvec/libhevcasm_hip.so hevcasm_amd/lib/libhevcasm_hip.so --rounds 3 -- python3 tools/bench_families.py pred8tap > $O/pred_scalar_desc_luma_ab.txt 2>&1
python3 tools/ab.py --libs hevcasm_amd/lib_ab/predvec/libhevcasm_hip.so hevcasm_amd/lib/libhevcasm_hip.so --rounds 3 -- python3 tools/bench_families.py pred4tap > $O/pred_scalar_desc_chroma_ab.txt 2>&1
echo job17 done
