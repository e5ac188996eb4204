#!/usr/bin/env python3
"""Means over the PU sizes (and the range) of the prediction rows of a tools/bench_families.py record, as the verdicts quote them.
usage: family_means.py profiles/r04_families_final.jsonl"""
import collections, json, sys
rows = collections.defaultdict(list)
for l in open(sys.argv[1]):
    if not l.startswith("{"):
        continue
    r = json.loads(l)
    if "frac_of_hbm_peak" in r and r.get("kernel", "").startswith(("pred_", "inverse", "dequant", "transform", "quantize", "hadamard", "ssd_linear", "pred_intra", "frame_sse", "intra_filter")):
        rows[r["kernel"]].append((r["frac_of_hbm_peak"], r.get("size", "")))
print("%-34s %5s %7s %7s %7s   %s" % ("kernel", "sizes", "mean", "min", "max", "slowest size"))
for k, v in rows.items():
    f = [x[0] for x in v]
    print("%-34s %5d %7.3f %7.3f %7.3f   %s" % (k, len(f), sum(f) / len(f), min(f), max(f), min(v)[1]))
