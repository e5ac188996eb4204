#!/bin/bash
# round 4, GPU job 14: final library -- whole GPU suite (with the extreme-content fuzz group), soak of the fuzz groups with other seeds and batch scales
O=gpurun_out/r4n; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
grep -q "rc=0" $O/pytest.log || { echo "parity failed"; exit 1; }
timeout -k 10 500 python3 tools/soak.py 400 30 1 > $O/soak.txt 2>&1; echo "soak x1 rc=$?"
timeout -k 10 300 python3 tools/soak.py 500 6 4 >> $O/soak.txt 2>&1; echo "soak x4 rc=$?"
timeout -k 10 400 python3 tools/soak.py 600 4 16 >> $O/soak.txt 2>&1; echo "soak x16 rc=$?"
tail -3 $O/soak.txt
echo job14 done
