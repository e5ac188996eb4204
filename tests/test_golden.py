"""Committed golden vectors (outputs of the reference's own C path):
 - CPU: the oracle restatement reproduces them; the case generators still produce the hashed inputs.
 - GPU: the HIP kernels reproduce them through the C-ABI."""
import pytest

import cases
import golden_io


def test_oracle_reproduces_golden_small(oracle_lib):
    got_any = 0
    for name, family, c, want in golden_io.small_cases():
        cases.assert_same(name, cases.run_oracle(family, c), want)
        got_any += 1
    assert got_any >= 60


@pytest.mark.parametrize("group", sorted(cases.ALL_CASE_LISTS))
def test_oracle_reproduces_golden_hashes(group, oracle_lib):
    """Every case of tests/cases.py: inputs are still the bytes the fixture was made from, and
    the oracle's outputs hash to what the reference produced."""
    h = golden_io.hashes()
    for name, family, c in cases.ALL_CASE_LISTS[group]():
        assert name in h, "case %s has no golden hash: re-run tests/golden/make_golden.py" % name
        assert golden_io.sha(golden_io.case_inputs(c)) == h[name]["inputs"], "%s: generated inputs changed" % name
        assert golden_io.sha(cases.run_oracle(family, c)) == h[name]["outputs"], "%s: oracle output differs from the reference's" % name


@pytest.mark.gpu
def test_hip_reproduces_golden_small():
    for name, family, c, want in golden_io.small_cases():
        cases.assert_same(name, cases.run_hip(family, c), want)
