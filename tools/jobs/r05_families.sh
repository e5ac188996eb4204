#!/bin/bash
# round 5: every family at every size on the final library (profiles/r05_families_final.jsonl + means), and the self-test with the HIP column
python tools/bench_families.py quant transform pred satd intra intramodes ssdlin me subpel > gpurun_out/r05_families_final.jsonl 2> gpurun_out/r05_families.err
tail -2 gpurun_out/r05_families.err
python tools/family_means.py gpurun_out/r05_families_final.jsonl > gpurun_out/r05_families_means.txt
cat gpurun_out/r05_families_means.txt
./hevcasm_amd/lib/hevcasm > gpurun_out/r05_selftest_mi355x.txt 2>&1; echo "selftest exit $?"
tail -3 gpurun_out/r05_selftest_mi355x.txt
