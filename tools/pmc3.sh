#!/bin/bash
# usage: tools/pmc3.sh <outdir-name> <python script and args...>
# The vector-memory stall counters (TA / TCP FIFOs, address translation), one group per run as tools/pmc2.sh.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$1; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
for g in "GRBM_GUI_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" \
         "TCP_RFIFO_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
         "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
         "TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_LFIFO_FULL_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
         "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  d=$O/$(echo $g | cut -d" " -f2)
  # A group the counter blocks cannot hold makes rocprofv3 abort at the first dispatch and then hang ("error code 38: Request exceeds the capabilities of
  # the hardware to collect": TA_TA_BUSY_sum + 3 TA_*_STALLED_* did, profiles/r02_pred_ab.txt).  The TA / TD / TCA blocks have two counter registers each
  # (rocprofv3 --list-avail: block sizes): refuse such a group here instead of finding out from a timeout on a GPU-holding process.
  for blk in TA_ TD_ TCA_; do
    if [ $(echo $g | tr " " "\n" | grep -c "^$blk") -gt 2 ]; then echo "SKIPPED (more than 2 ${blk}* counters in one pass): $g"; continue 2; fi
  done
  timeout -k 5 120 rocprofv3 --pmc $g --output-format csv -d $d -- python3 "$@" > $d.log 2>&1 || echo "FAILED $g"
  echo "done $g" >> $O/progress.txt
done
python3 tools/summarize_prof.py pmc "$*" $O/*/*/*_counter_collection.csv
