#!/usr/bin/env python3
"""Compare bench_families outputs: tools/cmp_fam.py new.jsonl old.jsonl [old2.jsonl ...] -> per row GB/s and per-kernel means."""
import json, sys, collections
def load(p):
    d = collections.OrderedDict()
    for l in open(p):
        l = l.strip()
        if l.startswith("{"):
            r = json.loads(l); d[(r["kernel"], str(r.get("size", "")))] = r
    return d
files = [load(p) for p in sys.argv[1:] if p != "-q"]
means = [collections.defaultdict(list) for _ in files]
for k in files[0]:
    row = [f.get(k, {}).get("algorithmic_gb_per_s") for f in files]
    for i, v in enumerate(row):
        if v is not None and all(x is not None for x in row): means[i][k[0]].append(v)
    if "-q" not in sys.argv: print(k[0].ljust(30), k[1].ljust(10), "  ".join("%7s" % (v if v is not None else "-") for v in row))
print("means (rows present in all files):")
for kern in means[0]:
    print(kern.ljust(30), "  ".join("%7.0f" % (sum(m[kern]) / len(m[kern])) for m in means))
