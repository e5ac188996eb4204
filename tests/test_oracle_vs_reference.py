"""Pins the oracle: our C restatement (oracle/hevc_oracle.c) must equal the reference's own C
path (oracle/_ref/refdrv = /root/reference/src/lib/*.c compiled in place) bit for bit, on every
hot-path family, all sizes, all fractions, extremes included.  CPU only."""
import pytest

import cases


@pytest.mark.parametrize("group", sorted(cases.ALL_CASE_LISTS))
def test_oracle_equals_reference(group, refproc, oracle_lib):
    for name, family, c in cases.ALL_CASE_LISTS[group]():
        cases.assert_same(name, cases.run_oracle(family, c), cases.run_reference(refproc, family, c))


def test_dct_matrix_properties(oracle_lib):
    """The generated transform matrix: even rows of size 2N are the size-N matrix, row 0 is 64,
    and rows are (nearly) orthogonal with norm about 64 * sqrt(N)."""
    import numpy as np
    m32 = oracle_lib.dct_matrix(32)
    for n in (4, 8, 16):
        assert np.array_equal(oracle_lib.dct_matrix(n), m32[:: 32 // n, :n])
    assert (m32[0] == 64).all()
    g = m32 @ m32.T
    assert np.abs(g - np.diag(np.diag(g))).max() <= 400  # HEVC's integer matrix is only approximately orthogonal
    assert np.abs(np.diag(g) - 64 * 64 * 32).max() < 64 * 64 * 32 * 0.01
