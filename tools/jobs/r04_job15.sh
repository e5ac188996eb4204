#!/bin/bash
# round 4, GPU job 15: rocprofv3 evidence for bench.py on the final library: kernel-trace stats (tools/profile_stats.sh) and the separate PMC passes of the dominant sizes (tools/profile_round.sh)
bash tools/profile_stats.sh r04 > gpurun_out/profile_stats_r04.log 2>&1; echo "profile_stats rc=$?"
bash tools/profile_round.sh r04 > gpurun_out/profile_round_r04.log 2>&1; echo "profile_round rc=$?"
echo job15 done
