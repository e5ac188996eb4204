#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the small text summaries committed under profiles/.

    summarize_prof.py stats  <kernel_stats.csv>                  -> per-kernel table (calls, avg, %)
    summarize_prof.py pmc    <label> <counter_collection.csv>... -> per-kernel mean counter values
Kernel names are shortened; torch's RNG fill kernels are reported as one line.
"""
import collections, csv, re, sys


def short(name):
    if "distribution_" in name or "random_" in name:
        return "torch RNG fill (input generation, outside the timed region)"
    m = re.search(r"hevc::(\w+)<([^>]*)>", name)
    if m:
        return "hevc::%s<%s>" % (m.group(1), m.group(2))
    return name[:100]


def stats(path):
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        k = short(r["Name"])
        d = rows.setdefault(k, [0, 0.0, 0.0])
        d[0] += int(r["Calls"]); d[1] += float(r["TotalDurationNs"]); d[2] += float(r["Percentage"])
    print("%-72s %7s %14s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
    for k, (calls, tot, pct) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print("%-72s %7d %14.3f %12.2f %7.2f" % (k, calls, tot / 1e6, tot / calls / 1e3, pct))


def pmc(label, paths):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in paths:
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            if k.startswith("torch RNG") or "hevc::" not in k:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("# %s" % label)
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            print("    %-38s dispatches=%-4d mean=%-14.6g min=%-14.6g max=%.6g" % (c, len(v), sum(v) / len(v), min(v), max(v)))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3:])
