#!/bin/bash
# round 4, GPU job 5: parity after the strip-kernel row segments and the full-search lane deal; A/B of each (segments by switch, lane deal and the
# VOP3P chain starts of luma bi-prediction by library build)
O=gpurun_out/r4e; mkdir -p $O
python3 -m pytest tests -m gpu -x -q -k "pred or footprint or golden or fuzz or full_search or me_ or shard or hip_equals" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
python3 tools/ab_env.py --env HEVCASM_PRED_NOSEGMENTS=1 - --rounds 3 -- python3 tools/bench_families.py pred4tap > $O/strip_segments_ab.txt 2>&1
python3 tools/ab.py --libs hevcasm_amd/lib/libhevcasm_hip.so hevcasm_amd/lib_ab/bilean/libhevcasm_hip.so --rounds 3 -- python3 tools/bench_families.py pred8tap > $O/bi_lean_ab.txt 2>&1
python3 tools/ab.py --libs hevcasm_amd/lib_ab/r4base/libhevcasm_hip.so hevcasm_amd/lib/libhevcasm_hip.so --rounds 4 -- python3 tools/bench_families.py meframe > $O/me_lane_deal_ab.txt 2>&1
bash tools/pmc2.sh r4e/pmc_sad4 tools/prof_one.py sad 16 16 4 + sad 8 8 4 + sad 32 32 4 + sad 32 64 4 + sad 12 16 4 > $O/pmc_sad4.txt 2>&1
bash tools/pmc3.sh r4e/pmc3_sad4 tools/prof_one.py sad 16 16 4 + sad 8 8 4 + sad 32 32 4 > $O/pmc3_sad4.txt 2>&1
echo job5 done
