#!/bin/bash
# runs the probe under two counter passes and prints accesses per wave-instruction
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/tcp_model; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 $R/tools/ubench/tcp_access_model.hip -o /tmp/tcp_access_model 2>/dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr --output-format csv -d $O/pmc -- /tmp/tcp_access_model > $O/run.log 2>&1
python3 - <<PY
import csv, glob, re, collections
acc = collections.OrderedDict()
for p in glob.glob("$O/pmc/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        m = re.search(r"probe<(\d+), (\d+), (\d+), (\d+)>", r["Kernel_Name"])
        if m: acc.setdefault(tuple(int(x) for x in m.groups()), {})[r["Counter_Name"]] = float(r["Counter_Value"])
print("bytes/lane lanes/row misalign pitch | TCP accesses per wave-instr | TCC read reqs per wave-instr | TA busy cycles per wave-instr (per CU avg x256/16384)")
for k, v in acc.items():
    n = 16384.0
    print("%9d %9d %8d %5d | %8.2f | %8.2f | %8.2f" % (k[0], k[1], k[2], k[3], v.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / n, v.get("TCP_TCC_READ_REQ_sum", 0) / n, v.get("TA_BUSY_avr", 0) * 256 / n))
PY
