#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of the bench command (full mix, dominant kernel only, headline kernel only),
#   then separate PMC passes (FETCH_SIZE, WRITE_SIZE, TCP/TCC counters) -- never combined with tracing.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_full -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --full-out $O/bench_full.full.json > $O/bench_full.json 2> $O/kt_full.err
for s in 64x64 16x16 64x48 48x64; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$s -- python3 bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-families --sizes $s --full-out $O/bench_$s.full.json > $O/bench_$s.json 2> $O/kt_$s.err
  for g in "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_BUSY_avr"; do
    d=$O/pmc_${s}_$(echo $g | cut -d" " -f1)
    rocprofv3 --pmc $g --output-format csv -d $d -- python3 bench.py --steps 2 --warmup 0 --cpu-seconds 0 --no-families --sizes $s --full-out /dev/null > $d.json 2> $d.err || echo "FAILED pmc $s $g"
  done
done
# traffic.json with the identity (sha256) of the library these passes ran on; copy it to profiles/traffic.json after the call
python3 tools/make_traffic.py $O $O/traffic.json > $O/traffic.log 2>&1 || echo "FAILED make_traffic"
echo profile_round done
