#!/bin/bash
# round 6: SQ counters of the fused residual loop on planes of one TU size (tools/bench_families.py tufusedsizes): instructions, VALU / LDS activity, bank conflicts
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/tufused_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
for g in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"; do
  d=$O/$(echo $g | cut -d" " -f1)
  timeout -k 5 200 rocprofv3 --pmc $g --output-format csv -d $d -- python3 tools/bench_families.py tufusedsizes > $d.log 2>&1 || echo "FAILED $g"
done
python3 tools/summarize_prof.py pmc "fused residual loop, one TU size per plane (32 x 1920x1088), per dispatch" $O/*/*/*_counter_collection.csv > gpurun_out/r06_tufused_counters.txt 2>&1
grep -A18 'fused_tu_runs' gpurun_out/r06_tufused_counters.txt | head -60
