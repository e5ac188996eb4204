"""BASELINE.json config #1: the library's own hevcasm_test SAD self-test on the CPU C path
(plumbing, no GPU): reference grammar, 23 partitions incl. 16x16, no -MISMATCH, zero errors.
Also the reference's OWN self-test output (through oracle/_ref/refdrv) has the same rows."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_DRIVER = r'''
import ctypes as C, sys
sys.path.insert(0, %r)
from hevcasm_amd import capi
L = capi.lib()
e = C.c_int(0)
for f in %r:
    getattr(L, "hevcasm_test_" + f)(C.byref(e), capi.C_REF | capi.C_OPT)
L.hevcasm_print_instruction_set_support(None, 0) if False else None
print("ERRORS", e.value)
'''

ROW = re.compile(r"^\t(\d+)x(\d+): C_OPT:(\d+)$")


def _run(families):
    # functional run: the timing columns are not asserted, so cap the harness's iteration counts
    r = subprocess.run([sys.executable, "-c", _DRIVER % (ROOT, tuple(families))], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HEVCASM_TEST_MAX_ITERATIONS="40"))
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def test_sad_selftest_grammar_and_zero_errors():
    out = _run(["sad"])
    assert "hevcasm_sad - Sum of Absolute Differences" in out
    rows = [ROW.match(l) for l in out.splitlines() if l.startswith("\t")]
    assert all(rows) and len(rows) == 23, out
    assert any(m.group(1) == "16" and m.group(2) == "16" for m in rows)
    assert "MISMATCH" not in out and "ERRORS 0" in out


def test_quick_families_zero_errors():
    out = _run(["sad_multiref", "ssd", "pred_intra", "hadamard_satd", "quantize_inverse", "quantize", "quantize_reconstruct", "pred_uni", "pred_bi", "ssd_linear",
                "inverse_transform_add", "transform"])
    assert "MISMATCH" not in out and "ERRORS 0" in out
    assert out.count("C_OPT:") == 23 + 5 + 7 + 3 + 4 + 4 + 4 + 24 * 8 + 22 * 4 + 1 + 5 + 5


def test_same_rows_as_the_reference_selftest(tmp_path, refproc):
    """The reference's hevcasm_test_sad (its own C files) prints the same partitions, in the same
    grammar, as ours."""
    path = str(tmp_path / "ref_selftest.txt")
    assert refproc.selftest(1, path) == 0
    ref_rows = [ROW.match(l).groups()[:2] for l in open(path).read().splitlines() if l.startswith("\t")]
    ours = [ROW.match(l).groups()[:2] for l in _run(["sad"]).splitlines() if l.startswith("\t")]
    assert ref_rows == ours


def test_same_row_labels_as_the_reference_for_the_quick_families(tmp_path, refproc):
    """Row labels (the text before the first column) of our self-tests equal the reference's own, family by
    family, for the families whose reference self-test runs in seconds on the C path."""
    path = str(tmp_path / "ref_selftest_more.txt")
    mask = 4 | 1024 | 128 | 256 | 512 | 2048 | 4096   # ssd, hadamard_satd, quantize_inverse, quantize, quantize_reconstruct, pred_intra, ssd_linear
    assert refproc.selftest(mask, path) == 0

    def labels(text):
        out = []
        for line in text.splitlines():
            if line.startswith("\t"):
                out.append(re.split(r"\s*C_OPT:", line.strip())[0].rstrip(" :"))
        return out
    ours = _run(["ssd", "pred_intra", "hadamard_satd", "quantize_inverse", "quantize", "quantize_reconstruct", "ssd_linear"])
    ref_rows, our_rows = labels(open(path).read()), labels(ours)
    # the reference tests the unfiltered intra packing only; ours adds the three edge-filtered rows
    our_rows = [r for r in our_rows if not r.endswith("edge")]
    assert sorted(ref_rows) == sorted(our_rows), (ref_rows, our_rows)


def test_executable_reports_instruction_sets():
    exe = os.path.join(ROOT, "hevcasm_amd", "lib", "hevcasm")
    assert os.path.exists(exe)
    # the full self-test takes minutes on the C path; only check that it starts and prints the header
    p = subprocess.Popen([exe], stdout=subprocess.PIPE, text=True, env=dict(os.environ, HEVCASM_DISABLE_HIP="1"))
    head = [p.stdout.readline() for _ in range(14)]
    p.kill()
    p.wait()
    text = "".join(head)
    assert "HEVCasm self test" in text and "[x] C_REF" in text and "[ ] HIP" in text


def test_selftest_simd_columns_and_host_bench():
    """With the host's x86 bits in the mask the SAD self-tests print SSE2 / AVX2 columns (csrc/host/sad_x86.c) and report no
    mismatch; the per-block timing driver bench.py uses for its host-SIMD baseline runs and prints one row per size and way count."""
    sys.path.insert(0, ROOT)
    from hevcasm_amd import capi
    mask = capi.lib().hevcasm_instruction_set_support() & ~capi.HIP
    if not (mask & capi.SSE2):
        import pytest
        pytest.skip("not an x86 host")
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); from hevcasm_amd import capi; L = capi.lib(); e = C.c_int(0)\n"
            "L.hevcasm_test_sad(C.byref(e), %d); L.hevcasm_test_sad_multiref(C.byref(e), %d); print('ERRORS', e.value)" % (ROOT, mask, mask))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, HEVCASM_TEST_MAX_ITERATIONS="40"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ERRORS 0" in r.stdout and "MISMATCH" not in r.stdout
    assert re.search(r"\t16x16: C_OPT:\d+ SSE2:\d+", r.stdout), r.stdout[:2000]
    if mask & capi.AVX2:
        assert re.search(r"4-way 64x48 :  C_OPT:\d+ AVX2:\d+", r.stdout), r.stdout[:2000]
    exe = os.path.join(ROOT, "hevcasm_amd", "lib", "hevcasm_host_bench")
    r = subprocess.run([exe, "%x" % mask, "1", "64"], capture_output=True, text=True, timeout=300, env=dict(os.environ, HEVCASM_DISABLE_HIP="1"))
    assert r.returncode == 0, r.stderr[-500:]
    rows = [l.split() for l in r.stdout.splitlines()]
    assert len(rows) == 40 and all(len(x) == 5 and int(x[3]) >= 64 for x in rows), r.stdout
