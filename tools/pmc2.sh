#!/bin/bash
# usage: tools/pmc2.sh <outdir-name> <python script and args...>
# rocprofv3 counter passes (each group in its own run, no tracing: the pool's rule) of an arbitrary bench command; prints
# per-kernel means.  Run through gpurun from the repo root.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$1; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
for g in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD" \
         "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM" \
         "GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  d=$O/$(echo $g | cut -d" " -f1)
  rocprofv3 --pmc $g --output-format csv -d $d -- python3 "$@" > $d.log 2>&1 || echo "FAILED $g"
done
python3 tools/summarize_prof.py pmc "$*" $O/*/*/*_counter_collection.csv
