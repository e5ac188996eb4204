#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of one HIP source, as the compiler reports it
(hipcc -Rpass-analysis=kernel-resource-usage, gfx950).  usage: tools/resusage.py hevcasm_amd/csrc/hip/sad.hip [filter]"""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC",
       "-I" + ROOT + "/include", "-I" + ROOT + "/hevcasm_amd/csrc/host", "-I" + ROOT + "/hevcasm_amd/csrc/hip",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/resusage.o"] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
name, vals = None, {}
keys = {"VGPRs": "vgpr", "AGPRs": "agpr", "TotalSGPRs": "sgpr", "ScratchSize [bytes/lane]": "scratch",
        "Occupancy [waves/SIMD]": "occ", "LDS Size [bytes/block]": "lds"}
def flush():
    if name and flt in name:
        print(name[:150].ljust(152), " ".join("%s=%s" % (k, vals.get(k, "?")) for k in ["vgpr", "agpr", "sgpr", "scratch", "lds", "occ"]))
for line in out.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        flush()
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name)
        vals = {}
        continue
    m = re.search(r"remark:\s+([^:]+): (\S+) \[-Rpass", line)
    if m and m.group(1).strip() in keys:
        vals[keys[m.group(1).strip()]] = m.group(2)
    elif "error" in line:
        print(line)
flush()
