#!/bin/bash
# round 4, GPU job 12: whole GPU suite (incl. the odd-alignment descriptor test) on the library as it stands; reconstruct offsets by scalar loads (A/B by build)
O=gpurun_out/r4l; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
grep -q "rc=0" $O/pytest.log || { echo "parity failed: no timing"; exit 1; }
python3 tools/ab.py --libs hevcasm_amd/lib_ab/reconvec/libhevcasm_hip.so hevcasm_amd/lib/libhevcasm_hip.so --rounds 4 -- python3 tools/bench_families.py quant > $O/recon_scalar_off_ab.txt 2>&1
echo job12 done
