#!/bin/bash
# usage: tools/profile_stats.sh <round tag>      (through gpurun, from the repo root)
# The kernel-trace half of tools/profile_round.sh alone, condensed straight into gpurun_out/<tag>_bench_kernel_stats.txt:
# rocprofv3 --kernel-trace --stats of the bench command (full mix with the headline families; dominant size alone; headline size alone).
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; T=$1; O=$R/gpurun_out/stats_$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
OUT=$R/gpurun_out/${T}_bench_kernel_stats.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/full -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --full-out $O/full.full.json > $O/full.json 2> $O/full.err
{
  echo "# rocprofv3 --kernel-trace --stats of: python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0   (MI355X, $T, final library; the kernels after the SAD mix are the inter-prediction / IDCT headline measurements)"
  echo "# hevc::sad4_window_kernel<W, H> = hevcasm_hip_sad_multiref_batch of 64-/48-wide blocks (sad_multiref_4 64x64 ... in bench.py's detail); template args of hevc::block_cost_kernel: <W, C bytes per lane, R rows in flight, U rows per trip, WAYS, OP(0=SAD), LOOP(several trips), TIGHT>; of hevc::pred_stream_kernel: <TAPS, dwords per lane, BI, rows in flight, NARROW, FAST (branch-free row loop)>"
  python3 tools/summarize_prof.py stats $O/full/*/*_kernel_stats.csv
} > $OUT
for s in 64x64 16x16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$s -- python3 bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-families --sizes $s --full-out /dev/null > $O/$s.json 2> $O/$s.err
  {
    echo "# $s alone: python3 bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-families --sizes $s   (bench line of the same process: roofline.avg_launch_ms $(python3 -c "import json,sys; print(json.loads(open('$O/$s.json').read().strip().splitlines()[-1])['roofline']['avg_launch_ms'])"))"
    python3 tools/summarize_prof.py stats $O/$s/*/*_kernel_stats.csv | grep -v "torch RNG\|copyBuffer"
  } >> $OUT
done
echo profile_stats done
