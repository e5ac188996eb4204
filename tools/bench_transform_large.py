"""Transforms at 4x the blocks per launch of tools/bench_families.py: launches of about 0.2 ms instead of 0.06 ms, i.e. closer to\nsteady state (ramp-up and tail of a 60 us launch cost 15-25 %)."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
only_sizes = [int(a) for a in sys.argv[1:]]      # e.g. "16 32": only those transform sizes
sys.argv = ["x"]
import importlib.util
spec = importlib.util.spec_from_file_location("bf", "tools/bench_families.py"); bf = importlib.util.module_from_spec(spec)
import torch
torch.cuda.set_device(0)
src = open("tools/bench_families.py").read().replace('if __name__ == "__main__":', 'if False:')
exec(compile(src, "bf", "exec"), bf.__dict__)
for n in (1 << 26, 1 << 28):
    print("coeffs_per_launch", n, flush=True)
    bf.bench_transform(n, label=" 2^%d" % (n.bit_length() - 1), only=[(0, s.bit_length() - 1) for s in only_sizes] or None)
