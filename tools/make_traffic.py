#!/usr/bin/env python3
"""profiles/traffic.json from the PMC passes of tools/profile_round.sh.

HBM bytes per launch = 2 * FETCH_SIZE KiB + WRITE_SIZE KiB, the gfx950 correction of
MI355X_MICROARCH.md (FETCH_SIZE counts 128-byte requests as 64 bytes on wide streaming reads),
calibrated here on the aligned SAD case: 2 * 269.3 MB = 539 MB against 545 MB algorithmic.
usage: make_traffic.py <gpurun_out/prof_rNN> <blocks_per_launch for 64x64> <blocks_per_launch for 16x16>
"""
import csv, glob, json, os, re, sys

prof = sys.argv[1]
out = {}
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
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
