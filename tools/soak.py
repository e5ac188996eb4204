#!/usr/bin/env python3
"""Soak run of the seeded fuzz tests (tests/test_hip_fuzz.py) with other seeds and longer batches, in ONE process:
    python tools/soak.py [first_seed [count [batch multiplier]]]
Prints one line per seed; exits non-zero at the first mismatch (the assertion names the case)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 5
os.environ["HEVCASM_FUZZ_NMUL"] = sys.argv[3] if len(sys.argv) > 3 else "1"
import __graft_entry__ as g
g.build()
mod = None
for seed in range(first, first + count):
    os.environ["HEVCASM_FUZZ_SEED"] = str(seed * 1000)
    mod = importlib.reload(mod) if mod else importlib.import_module("test_hip_fuzz")
    t = time.time()
    for name in sorted(dir(mod)):
        if name.startswith("test_"):
            getattr(mod, name)()
    print("seed %d x%s: all fuzz groups bit-exact (%.0f s)" % (seed, os.environ["HEVCASM_FUZZ_NMUL"], time.time() - t), flush=True)
