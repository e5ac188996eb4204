#!/bin/bash
# round 4, GPU job 37: rocprofv3 kernel-trace stats of the bench command on the final library, then the soak of the fuzz groups (other seeds, larger batches)
O=gpurun_out/r4ah; mkdir -p $O
bash tools/profile_stats.sh r04 > gpurun_out/profile_stats_r04.log 2>&1; echo "profile_stats rc=$?"
timeout -k 10 400 python3 tools/soak.py 1400 30 1 > $O/soak.txt 2>&1; echo "soak x1 rc=$?"
timeout -k 10 200 python3 tools/soak.py 1500 6 4 >> $O/soak.txt 2>&1; echo "soak x4 rc=$?"
timeout -k 10 300 python3 tools/soak.py 1600 4 16 >> $O/soak.txt 2>&1; echo "soak x16 rc=$?"
tail -2 $O/soak.txt
echo job37 done
