"""First run of the intra frame kernel: every pinned and unpinned case against the oracle."""
import sys, time, traceback
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import cases
bad = 0
for lst in (cases.intra_frame_cases, cases.intra_frame_mixed_cases):
    for name, fam, c in lst():
        t = time.time()
        try:
            got = cases.run_hip(fam, c)
            want = cases.run_oracle(fam, c)
            cases.assert_same(name, got, want)
            print("ok  ", name, "%.2fs" % (time.time() - t), flush=True)
        except Exception as e:
            bad += 1
            print("FAIL", name, repr(e)[:300], flush=True)
            try:
                g, w = got[0].astype(int), want[0].astype(int)
                ys, xs = np.nonzero(g != w)
                print("     first wrong pixel (y, x):", ys[0], xs[0], "count", len(ys), "rows", sorted(set(ys // 4 * 4))[:8], "cols", sorted(set(xs // 4 * 4))[:8], flush=True)
            except Exception:
                pass
print("failures:", bad)
sys.exit(1 if bad else 0)
