"""Multi-GPU path on CPU: the batch shards by contiguous block ranges, one process per rank, no
data-path collective; only the timing barrier / max-reduce uses torch.distributed.  Rehearsed
here with gloo, world_size 2 (the driver runs the real thing over RCCL on 8 GPUs)."""
import os
import subprocess
import sys

import numpy as np

from hevcasm_amd.shard import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 1000, 1 << 20, (1 << 20) + 5):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
import bench
from hevcasm_amd import workload as wl
from hevcasm_amd.shard import shard_range
from oracle import pyoracle

dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=2)
rank, world = dist.get_rank(), dist.get_world_size()
c = wl.sad_case(16, 16, 1001, 31)
a, b = shard_range(c["n"], rank, world)
# each rank evaluates only its slice (the oracle stands in for the GPU on this CPU-only box)
mine = pyoracle.sad_batch(c["src"], c["stride_src"], c["ref"], c["stride_ref"], c["src_off"][a:b], c["ref_off"][a:b], 16, 16)
# bench.py's timing reduction: barrier, then MAX over ranks of the local elapsed time
t = bench.max_over_ranks(0.010 * (rank + 1))
assert abs(t - 0.020) < 1e-9, t
parts = [None, None]
dist.all_gather_object(parts, (a, b, mine))
if rank == 0:
    full = pyoracle.sad_batch(c["src"], c["stride_src"], c["ref"], c["stride_ref"], c["src_off"], c["ref_off"], 16, 16)
    got = np.concatenate([p[2] for p in sorted(parts)])
    assert np.array_equal(got, full)
    print("SHARD_OK", parts[0][:2], parts[1][:2])
dist.destroy_process_group()
'''


def test_two_rank_gloo_shards_reassemble():
    port = 29500 + os.getpid() % 2000
    code = _WORKER % dict(root=ROOT, port=port)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    assert "SHARD_OK" in outs[0][0]
