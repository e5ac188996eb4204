#!/bin/bash
# round 6, final library: the intra frame table DESIGN 4.3 quotes -- parity cases, then 1080p frames alone and in batches, the decoder, per-TU cost on independent one-CTU frames
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
fmt='
import sys, json
for l in sys.stdin:
    r = json.loads(l); print("%-16s %-7s %-9s F=%-3d ms=%8.3f ms/frame=%7.3f fps=%8.1f us/step=%s%s" % (r["map"], r["frame"], "decode" if r["decode"] else "encode", r["frames"], r["ms"], r["ms_per_frame"], r["frames_per_s"], r["us_per_step"], "  us/TU=%.3f" % (r["ms"] * 1e3 / r["tus_per_frame"]) if r["frame"] == "64x64" else ""))
'
timeout -k 10 300 python tools/jobs/r06_intra_first.py > gpurun_out/r06_intra_parity_final.log 2>&1 || { grep -v "^ok" gpurun_out/r06_intra_parity_final.log | tail -20; exit 1; }
tail -1 gpurun_out/r06_intra_parity_final.log
timeout -k 10 400 python tools/bench_intra_frames.py --frames 1,64 --reps 5 | python -c "$fmt"
timeout -k 10 400 python tools/bench_intra_frames.py --frames 128,256 --maps mixed_quadtree,all_32x32 --reps 3 | python -c "$fmt"
timeout -k 10 400 python tools/bench_intra_frames.py --frames 1,64 --maps mixed_quadtree --reps 5 --decode | python -c "$fmt"
timeout -k 10 400 python tools/bench_intra_frames.py --frames 256 --w 64 --h 64 --maps all_4x4,all_8x8,all_16x16,all_32x32 --reps 5 | python -c "$fmt"
