#!/usr/bin/env python3
"""profiles/traffic.json from the PMC passes of tools/profile_round.sh.

HBM bytes per launch = 2 * FETCH_SIZE KiB + WRITE_SIZE KiB, the gfx950 correction of
MI355X_MICROARCH.md (FETCH_SIZE counts 128-byte requests as 64 bytes on wide streaming reads),
calibrated here on the aligned SAD case: 2 * 269.3 MB = 539 MB against 545 MB algorithmic.
usage: make_traffic.py <gpurun_out/prof_rNN> [<output.json>, default profiles/traffic.json]
"""
import csv, glob, hashlib, json, os, re, subprocess, sys

prof = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the identity of what was measured: bench.py reports roofline.traffic only when the library it has loaded hashes to this (round 6).
# Run this script where the PMC passes ran (the same tree, the same .so), e.g. as the last step of tools/profile_round.sh.
LIB = os.environ.get("HEVCASM_HIP_LIB") or os.path.join(ROOT, "hevcasm_amd", "lib", "libhevcasm_hip.so")
try:
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except OSError:
    head = None
if not head and os.path.exists(os.path.join(ROOT, "gpurun_out", "git_head.txt")):      # the GPU box has no .git: the caller leaves the hash there
    head = open(os.path.join(ROOT, "gpurun_out", "git_head.txt")).read().strip()
out = {"_measured_on": dict(library_sha256=hashlib.sha256(open(LIB, "rb").read()).hexdigest(), git_head=head, profile_dir=os.path.basename(os.path.normpath(prof)))}
for size in ("64x64", "16x16", "64x48", "48x64"):
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        for p in glob.glob(os.path.join(prof, "pmc_%s_%s" % (size, counter), "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(p)):
                m = re.search(r"hevc::block_cost_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)[,>]", r["Kernel_Name"])
                ways = int(m.group(5)) if m else (4 if "hevc::sad4_window_kernel<" in r["Kernel_Name"] else None)     # the 4-way kernel of 64-/48-wide blocks
                if ways and r["Counter_Name"] == counter:
                    vals.setdefault((ways, counter), []).append(float(r["Counter_Value"]))
    bench = json.load(open(os.path.join(prof, "bench_%s.full.json" % size)))      # the full record (--full-out): the stdout line is compact since round 5
    for ways, name in ((1, "sad"), (4, "sad_multiref_4")):
        f, w = vals.get((ways, "FETCH_SIZE")), vals.get((ways, "WRITE_SIZE"))
        if not f or not w:
            continue
        row = [r for r in bench["detail"] if r["kernel"] == name and r["size"] == size][0]
        blocks = bench["config"]["blocks_per_step_per_gpu"] // 2 // row["launches_per_pass"]
        fetch, write = sum(f) / len(f) * 1024, sum(w) / len(w) * 1024
        out["%s %s" % (name, size)] = dict(
            hbm_bytes_per_launch=int(2 * fetch + write), fetch_size_kib=round(sum(f) / len(f), 1), write_size_kib=round(sum(w) / len(w), 1),
            blocks_per_launch=blocks, algorithmic_bytes_per_launch=row["bytes_per_block"] * blocks,
            formula="2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes), separate --pmc passes", dispatches=len(f))
dest = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "traffic.json")
json.dump(out, open(dest, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
