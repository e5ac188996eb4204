import time, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from hevcasm_amd import workload as wl, batch as B
from oracle import pyoracle
import test_cost_runs as t
rng = np.random.default_rng(20261012)
for i in range(8):
    w, h = 8 * int(rng.integers(1, 40)), 8 * int(rng.integers(1, 30))
    kind_of = ["noise", "video", "max"][int(rng.integers(0, 3))]
    t0 = time.time(); c = wl.cost_frame_case(w, h, 9700 + i, kind=kind_of, search=int(rng.integers(0, 40))); t1 = time.time()
    for kind in t.KINDS:
        a = time.time(); g = t._hip(c, kind); b = time.time(); o = t._oracle(pyoracle, c, kind); d = time.time()
        print(w, h, kind_of, kind, "case %.2f hip %.3f oracle %.3f" % (t1 - t0, b - a, d - b), np.array_equal(g, o), flush=True)
