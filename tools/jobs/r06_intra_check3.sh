#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout -k 10 300 python tools/jobs/r06_intra_first.py > gpurun_out/r06_intra_parity5.log 2>&1 || { grep -v "^ok" gpurun_out/r06_intra_parity5.log | tail -20; exit 1; }
tail -1 gpurun_out/r06_intra_parity5.log
timeout -k 10 400 python -m pytest tests/test_intra_frames.py tests/test_hip_fuzz.py -x -q -m gpu -k "intra" 2>&1 | tail -2
fmt='
import sys, json
for l in sys.stdin:
    r = json.loads(l); print("%-16s F=%-3d ms=%8.3f ms/frame=%7.3f fps=%8.1f us/step=%s" % (r["map"], r["frames"], r["ms"], r["ms_per_frame"], r["frames_per_s"], r["us_per_step"]))
'
timeout -k 10 400 python tools/bench_intra_frames.py --frames 1,64 --reps 3 | python -c "$fmt"
