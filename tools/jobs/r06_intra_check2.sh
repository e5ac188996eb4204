#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout -k 10 300 python tools/jobs/r06_intra_first.py > gpurun_out/r06_intra_parity4.log 2>&1 || { grep -v "^ok" gpurun_out/r06_intra_parity4.log | tail -20; exit 1; }
tail -1 gpurun_out/r06_intra_parity4.log
timeout -k 10 400 python -m pytest tests/test_intra_modes.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -2
fmt='
import sys, json
for l in sys.stdin:
    r = json.loads(l); print("%-16s F=%-3d ms=%8.3f ms/frame=%7.3f fps=%8.1f us/step=%s" % (r["map"], r["frames"], r["ms"], r["ms_per_frame"], r["frames_per_s"], r["us_per_step"]))
'
timeout -k 10 400 python tools/bench_intra_frames.py --frames 1,64 --reps 3 | python -c "$fmt"
timeout -k 10 400 python tools/bench_intra_frames.py --frames 128,256 --reps 3 --maps mixed_quadtree,all_32x32 | python -c "$fmt"
timeout -k 10 400 python tools/bench_intra_frames.py --frames 1,64 --reps 3 --maps mixed_quadtree --decode | python -c "$fmt"
timeout -k 10 200 python tools/bench_families.py intramodes 2>/dev/null | cut -c1-220
