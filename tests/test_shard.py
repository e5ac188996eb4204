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


# ----------------------------------------------------------------------- config #5: CTU rows + halo

def _oracle_rows_fn(slab):
    """A rank's search on the CPU: the oracle over the slab's CTUs.  The slab is copied into exactly-sized arrays first,
    so anything the rank would need beyond its rows + halo is simply not there."""
    import numpy as np
    from hevcasm_amd import me_frame
    from oracle import pyoracle
    cur, ref = slab.cur.copy(), slab.ref.copy()
    out = np.zeros((slab.n_ctus, 3), dtype=np.int32)
    for j in me_frame.ctu_jobs(slab.frame_h, slab.frame_w, slab.ctu, slab.stride_cur, slab.stride_ref, slab.pad, slab.row0, slab.row1,
                               cur_row_base=slab.y0, ref_row_base=slab.ref_row_base):
        out[j["index"]] = pyoracle.me_full_search(cur, slab.stride_cur, ref, slab.stride_ref, j["cur_off"], j["ref_off"], j["w"], j["h"], slab.range)
    return out


def _whole_frame_oracle(f, ctu, r):
    from hevcasm_amd import me_frame
    from oracle import pyoracle
    rows, cols = -(-f["h"] // ctu), -(-f["w"] // ctu)
    want = np.zeros((rows * cols, 3), dtype=np.int32)
    for j in me_frame.ctu_jobs(f["h"], f["w"], ctu, f["stride_cur"], f["stride_ref"], f["pad"], 0, rows):
        want[j["index"]] = pyoracle.me_full_search(f["cur"], f["stride_cur"], f["ref"], f["stride_ref"], j["cur_off"], j["ref_off"], j["w"], j["h"], r)
    return want


def test_me_slab_rows_cover_exactly_the_search_windows():
    from hevcasm_amd.shard import ctu_row_range, me_slab_rows
    # 2160 rows, 64x64 CTUs, +-64, 8 ranks: 34 CTU rows as 5,5,4,...; the last rank's slab ends with the 48-row CTUs
    spans = [ctu_row_range(34, r, 8) for r in range(8)]
    assert [b - a for a, b in spans] == [5, 5, 4, 4, 4, 4, 4, 4]
    assert me_slab_rows(2160, 64, 64, *spans[0]) == (0, 320, -64, 384)
    assert me_slab_rows(2160, 64, 64, *spans[7]) == (1920, 2160, 1856, 2224)
    # neighbouring slabs share exactly 2 * range reference rows and no current-frame row
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        ya = me_slab_rows(2160, 64, 64, a0, a1)
        yb = me_slab_rows(2160, 64, 64, b0, b1)
        assert ya[1] == yb[0] and ya[3] - yb[2] == 128
    with np.testing.assert_raises(ValueError):
        me_slab_rows(2160, 64, 64, 30, 35)
    assert ctu_row_range(2, 3, 4) == (2, 2)          # more ranks than rows: empty range


def test_me_frame_splits_equal_the_unsharded_search():
    """Every split of a frame into CTU-row slabs (each searched on a private copy of its rows + halo) reassembles into
    the unsharded result: seams between ranks, the 64x48 bottom row, ragged right column, an idle rank."""
    from hevcasm_amd import me_frame
    for (fw, fh, ctu, r, worlds) in ((200, 240, 64, 64, (1, 2, 3, 4, 6)), (96, 80, 32, 20, (2, 3))):
        f = me_frame.synthetic_frame_pair(fw, fh, r, seed=fw, ctu=ctu, max_motion=min(r, 30))
        want = _whole_frame_oracle(f, ctu, r)
        assert len(set(map(tuple, want[:, :2]))) > len(want) // 2
        for world in worlds:
            parts = [me_frame.search_rank(f, ctu, r, rank, world, _oracle_rows_fn) for rank in range(world)]
            assert np.array_equal(me_frame.assemble(parts, fw, fh, ctu), want), (fw, fh, world)


def _oracle_run_fn(run):
    """A rank's run of CTUs on the CPU: the oracle over exactly those CTUs, on private copies of the slab's rows."""
    from hevcasm_amd import me_frame
    from oracle import pyoracle
    sl = run.slab
    cur, ref = sl.cur.copy(), sl.ref.copy()
    out = np.zeros((sl.n_ctus, 3), dtype=np.int32)
    for j in me_frame.ctu_jobs(sl.frame_h, sl.frame_w, sl.ctu, sl.stride_cur, sl.stride_ref, sl.pad, sl.row0, sl.row1, cur_row_base=sl.y0, ref_row_base=sl.ref_row_base):
        out[j["index"]] = pyoracle.me_full_search(cur, sl.stride_cur, ref, sl.stride_ref, j["cur_off"], j["ref_off"], j["w"], j["h"], sl.range)
    a = run.first - sl.row0 * run.cols          # the run inside the slab's CTUs
    return out[a: a + run.n_ctus]


def test_me_frame_splits_by_ctu_equal_the_unsharded_search():
    """The split balanced to one CTU (shard.ctu_range): runs that begin and end in the middle of a row, a row shared by two or
    three ranks, more ranks than CTUs."""
    from hevcasm_amd import me_frame
    from hevcasm_amd.shard import ctu_range, rows_of_ctus
    assert [ctu_range(2040, r, 8) for r in range(8)][:2] == [(0, 255), (255, 510)]
    assert rows_of_ctus(255, 510, 60) == (4, 9) and rows_of_ctus(0, 60, 60) == (0, 1) and rows_of_ctus(7, 7, 60) == (0, 0)
    for (fw, fh, ctu, r, worlds) in ((200, 240, 64, 64, (1, 2, 3, 5, 7, 16, 20)), (96, 80, 32, 20, (2, 4, 11))):
        f = me_frame.synthetic_frame_pair(fw, fh, r, seed=fw + 1, ctu=ctu, max_motion=min(r, 30))
        want = _whole_frame_oracle(f, ctu, r)
        for world in worlds:
            parts = [me_frame.search_rank_ctus(f, ctu, r, rank, world, _oracle_run_fn) for rank in range(world)]
            assert np.array_equal(me_frame.assemble_ctus(parts, fw, fh, ctu), want), (fw, fh, world)


_ME_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
from hevcasm_amd import me_frame
import bench, test_shard

dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=2)
rank, world = dist.get_rank(), dist.get_world_size()
# 320 x 240: CTU rows of 64, 64, 64 and 48 pixels -> rank 0 owns rows 0-1, rank 1 rows 2-3 (the seam at row 128 and the 64x48 bottom row)
f = me_frame.synthetic_frame_pair(320, 240, 64, seed=5)
# bench.py's own config #5 path: search this rank's slab, then collect the answers (bench.gather_me_results)
part = me_frame.search_rank(f, 64, 64, rank, world, test_shard._oracle_rows_fn)   # the oracle stands in for the GPU on this CPU-only box
table = bench.gather_me_results(part, 320, 240, 64)
if rank == 0:
    want = test_shard._whole_frame_oracle(f, 64, 64)
    assert table.shape == (20, 3) and np.array_equal(table, want), (table, want)
    print("ME_SHARD_OK", part[0], part[1])
else:
    assert table is None and (part[0], part[1]) == (2, 4)
dist.destroy_process_group()
'''


def test_two_rank_gloo_me_frame_rows_and_halo():
    port = 31500 + os.getpid() % 2000
    code = _ME_WORKER % dict(root=ROOT, port=port)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    assert "ME_SHARD_OK 0 2" in outs[0][0]


# ----------------------------------------------------------------------- round 6: F frames per launch (bench.py --workload me5 --frames F)

def _oracle_stack_fn(stack):
    """A rank's F slabs on the CPU: the oracle over the run of each frame, each on a private copy cut out of the STACKED arrays at the stack's pitches
    (so a wrong pitch, or a frame reading its neighbour's rows, shows)."""
    from hevcasm_amd import me_frame
    from oracle import pyoracle
    sl = stack.slab
    out = np.zeros((stack.n_frames, stack.n_ctus, 3), dtype=np.int32)
    for f in range(stack.n_frames):
        cur = stack.cur[f * stack.pitch_cur:(f + 1) * stack.pitch_cur].copy()
        ref = stack.ref[f * stack.pitch_ref:(f + 1) * stack.pitch_ref].copy()
        full = np.zeros((sl.n_ctus, 3), dtype=np.int32)
        for j in me_frame.ctu_jobs(sl.frame_h, sl.frame_w, sl.ctu, sl.stride_cur, sl.stride_ref, sl.pad, sl.row0, sl.row1, cur_row_base=sl.y0, ref_row_base=sl.ref_row_base):
            full[j["index"]] = pyoracle.me_full_search(cur, sl.stride_cur, ref, sl.stride_ref, j["cur_off"], j["ref_off"], j["w"], j["h"], sl.range)
        a = stack.first - sl.row0 * stack.runs[0].cols
        out[f] = full[a: a + stack.n_ctus]
    return out


def test_me_f_frames_per_launch_split_equals_the_unsharded_search():
    """Each rank's share of F frames as one stack (what one launch takes): every frame of every split reassembles into that frame's unsharded result."""
    from hevcasm_amd import me_frame
    frames = [me_frame.synthetic_frame_pair(200, 176, 32, seed=40 + i, ctu=64, max_motion=20) for i in range(3)]
    want = [_whole_frame_oracle(f, 64, 32) for f in frames]
    assert not np.array_equal(want[0], want[1])
    for world in (1, 2, 3, 5, 13):
        parts = [me_frame.search_rank_ctus_frames(frames, 64, 32, rank, world, _oracle_stack_fn) for rank in range(world)]
        for k in range(3):
            got = me_frame.assemble_ctus([(a, b, res[k]) for (a, b, res) in parts], 200, 176, 64)
            assert np.array_equal(got, want[k]), (world, k)


_ME_FRAMES_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
from hevcasm_amd import me_frame
import bench, test_shard

dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=2)
rank, world = dist.get_rank(), dist.get_world_size()
frames = [me_frame.synthetic_frame_pair(192, 112, 16, seed=60 + i, ctu=64, max_motion=12) for i in range(4)]
first, last, res = me_frame.search_rank_ctus_frames(frames, 64, 16, rank, world, test_shard._oracle_stack_fn)      # the oracle stands in for the GPU
# bench.py's own collection, frame by frame
tables = [bench.gather_me_results((first, last, res[k]), 192, 112, 64, unit="ctus") for k in range(4)]
if rank == 0:
    for k in range(4):
        assert np.array_equal(tables[k], test_shard._whole_frame_oracle(frames[k], 64, 16)), k
    print("ME_FRAMES_OK", first, last)
dist.destroy_process_group()
'''


def test_two_rank_gloo_me_f_frames():
    port = 33500 + os.getpid() % 2000
    code = _ME_FRAMES_WORKER % dict(root=ROOT, port=port)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    assert "ME_FRAMES_OK 0 3" in outs[0][0]


import pytest


@pytest.mark.gpu
def test_hip_f_frames_in_one_launch_equal_the_oracle_and_the_per_frame_launches():
    """hevcasm_hip_me_full_search_frames_ctus: every split of 3 frames, each rank's F slabs in ONE launch, reassembles per frame into the unsharded oracle;
    and a 4K-sized run of 255 CTUs x 4 frames equals the four single-frame launches (the bench's geometry, part-workgroups and their decode pass included)."""
    import torch
    from hevcasm_amd import batch as B, me_frame
    frames = [me_frame.synthetic_frame_pair(200, 176, 32, seed=40 + i, ctu=64, max_motion=20) for i in range(3)]
    want = [_whole_frame_oracle(f, 64, 32) for f in frames]
    fn = me_frame.hip_frames_fn()
    for world in (1, 2, 5):
        parts = [me_frame.search_rank_ctus_frames(frames, 64, 32, rank, world, fn) for rank in range(world)]
        for k in range(3):
            assert np.array_equal(me_frame.assemble_ctus([(a, b, res[k]) for (a, b, res) in parts], 200, 176, 64), want[k]), (world, k)
    big = [me_frame.synthetic_frame_pair(3840, 2160, 64, seed=70 + i) for i in range(4)]
    st = me_frame.CtuRunStack(big, 64, 64, 255, 510)
    many = fn(st)
    one = me_frame.hip_ctus_fn()
    for k in range(4):
        assert np.array_equal(many[k], one(st.runs[k])), k
