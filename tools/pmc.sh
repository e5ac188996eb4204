#!/bin/bash
# usage: pmc.sh <outdir-name> <prof_one.py args...>   -- SQ / memory counter passes for one kernel configuration
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$1; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
for g in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr" "FETCH_SIZE" "WRITE_SIZE"; do
  d=$O/$(echo $g | cut -d" " -f1)
  rocprofv3 --pmc $g --output-format csv -d $d -- python3 tools/prof_one.py "$@" > $d.log 2>&1 || echo "FAILED $g"
done
python3 tools/summarize_prof.py pmc "$*" $O/*/*/*_counter_collection.csv
