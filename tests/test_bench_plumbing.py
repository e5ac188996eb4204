"""bench.py's host-side pieces that need no GPU: byte accounting, the core-count report of the CPU baseline,
the record of library / environment, the full-search roof it prices against, and the collection of a sharded frame's answers."""
import json
import os

import numpy as np

import bench
from hevcasm_amd import workload as wl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_algorithmic_bytes_follow_the_survey():
    # one reference: 2WH + 4 (result) + 8 (two u32 descriptors)
    for (w, h) in wl.SAD_BENCH_SIZES:
        assert bench.algorithmic_bytes(w, h, 1) == 2 * w * h + 12
        four = bench.algorithmic_bytes(w, h, 4)
        # four references: compulsory bytes -- never more than SURVEY's disjoint-candidate figure (+ descriptors), never less than source + one candidate
        assert w * h + w * h + 16 + 20 <= four <= 5 * w * h + 16 + 20
    assert bench.algorithmic_bytes(64, 64, 4) == 8354        # the figure DESIGN.md and the bench line quote for the dominant kernel


def test_host_cores_reports_what_it_used():
    n, info = bench.host_cores(16)
    assert 1 <= n <= 16 and n <= info["usable_cpus"] <= info["host_cpus"] == (os.cpu_count() or 1)
    assert str(n) in info["note"]
    n1, _ = bench.host_cores(1)
    assert n1 == 1


def test_library_record_names_the_in_tree_build():
    rec = bench.library_record()
    assert rec["library"] == os.path.join("hevcasm_amd", "lib", "libhevcasm_hip.so")
    assert rec["env_switches"] == {k: v for k, v in os.environ.items() if k.startswith("HEVCASM_")}


def test_qsad_roof_is_the_measured_one():
    peak, src = bench.qsad_peak()
    with open(os.path.join(ROOT, "profiles", "r03_qsad_rate.json")) as f:
        rec = json.load(f)
    assert peak == rec["v_qsad_pk_u16_u8_tera_byte_differences_per_s"] and "measured" in src
    assert 0.85 < peak / bench.NOMINAL_QSAD_TBYTEDIFFS < 1.0       # 17.2 cycles per instruction against the nominal 16


def test_gather_without_a_process_group_is_the_identity():
    res = np.arange(3 * 6, dtype=np.int32).reshape(6, 3)
    table = bench.gather_me_results((0, 2, res), 3 * 64, 2 * 64, 64)
    assert np.array_equal(table, res)
