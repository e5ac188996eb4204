#!/usr/bin/env python3
"""Same-box A/B of environment switches of ONE library build: runs a bench command once per setting per round, alternating, and
prints the per-row medians side by side (first setting = baseline).  "-" = nothing set.

    python tools/ab_env.py --env - HEVCASM_PRED_WINDOW=1 HEVCASM_PRED_WINDOW=2 --rounds 2 -- python tools/bench_families.py pred16
"""
import collections, json, os, subprocess, sys
import numpy as np

args = sys.argv[1:]
cut = args.index("--")
cmd = args[cut + 1:]
opts = args[:cut]
envs = opts[opts.index("--env") + 1:opts.index("--rounds")] if "--rounds" in opts else opts[opts.index("--env") + 1:]
rounds = int(opts[opts.index("--rounds") + 1]) if "--rounds" in opts else 2
ms = [collections.OrderedDict() for _ in envs]
for r in range(rounds):
    for i, e in enumerate(envs):
        env = dict(os.environ)
        if e != "-":
            for kv in e.split(","):
                k, v = kv.split("=", 1)
                env[k] = v
        out = subprocess.run(cmd, env=env, capture_output=True, text=True)
        if out.returncode != 0:
            print("FAILED", e, out.stderr[-2000:]); sys.exit(1)
        for l in out.stdout.splitlines():
            if l.startswith("{"):
                j = json.loads(l)
                ms[i].setdefault((j["kernel"], str(j.get("size", ""))), []).append(j["ms"])
        print("# round %d setting %d done" % (r, i), flush=True)
print("# " + "  ".join("[%d] %s" % (i, l) for i, l in enumerate(envs)))
ratios = collections.defaultdict(list)
for k in ms[0]:
    med = [float(np.median(m.get(k, [np.nan]))) for m in ms]
    print(k[0].ljust(30), k[1].ljust(18), "  ".join("%8.4f" % v for v in med), "  " + "  ".join("%+6.1f%%" % (100 * (v / med[0] - 1)) for v in med[1:]))
    for i, v in enumerate(med[1:]): ratios[(k[0], i)].append(v / med[0])
print("# geometric mean of time ratios vs [0], per kernel:")
for (kern, i), v in ratios.items():
    print(kern.ljust(30), "[%d]" % (i + 1), "%+6.1f%%" % (100 * (float(np.exp(np.mean(np.log(v)))) - 1)))
