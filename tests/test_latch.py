"""Round 6: a HIP failure behind a per-block table slot must not keep killing the process.  The first failure latches the
device "unavailable" with its text, goes to the application's failure handler (abort() only without one), and every later hevcasm_populate_*() drops
the HEVCASM_HIP bit -- slots from the other bits of the mask, or NULL, the reference's convention for "not available" (ssd.c:65-69)."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PRELUDE = """
import ctypes as C, sys
sys.path.insert(0, %r)
from hevcasm_amd import capi
L = capi.lib()
seen = []
HANDLER = C.CFUNCTYPE(None, C.c_char_p)
handler = HANDLER(lambda text: seen.append(text.decode()))
def slots(name, mask, n=300):
    buf = (C.c_void_p * n)()
    getattr(L, "hevcasm_populate_" + name)(buf, mask)
    return [buf[i] for i in range(n)]
""" % ROOT


def _run(body, expect_rc=0):
    r = subprocess.run([sys.executable, "-c", PRELUDE + textwrap.dedent(body)], capture_output=True, text=True, timeout=300)
    assert r.returncode == expect_rc, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    return r


def test_failing_per_block_call_latches_and_later_populates_fall_to_host_slots():
    """No GPU here: the per-block SAD entry fails at its first HIP call.  With a handler installed the call returns (value unspecified), the latch is
    set with the failure's text, and a populate with the HIP bit afterwards hands out the C slots -- which work."""
    _run("""
        import numpy as np
        assert L.hevcasm_hip_available() == 0 and L.hevcasm_hip_latched() == 0
        L.hevcasm_hip_set_failure_handler(C.cast(handler, C.c_void_p))
        fn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_uint32)
        a = np.arange(256, dtype=np.uint8); b = a[::-1].copy()
        hip1 = fn(("hevcasm_hip1_sad", C.CDLL(capi.LIB_PATH)))
        hip1(a.ctypes.data, 16, b.ctypes.data, 16, capi.rect(16, 16))          # fails inside: no device
        assert len(seen) == 1 and "per-block path" in seen[0] and "failed" in seen[0], seen
        assert L.hevcasm_hip_latched() == 1 and L.hevcasm_hip_latch_reason().decode() == seen[0]
        # the next populate does not abort and does not hand out HIP entries
        c_ref = slots("sad", capi.C_REF)
        assert slots("sad", capi.C_REF | capi.HIP) == c_ref and not any(slots("sad", capi.HIP))
        want = int(np.abs(a.astype(int) - b.astype(int)).sum())
        assert fn(c_ref[6])(a.ctypes.data, 16, b.ctypes.data, 16, capi.rect(16, 16)) == want      # slot 6 = 16x16 (sad.h:54-65)
        assert L.hevcasm_instruction_set_hip() == 0
        # a second failing call reports again but the first text is the one kept
        hip1(a.ctypes.data, 16, b.ctypes.data, 16, capi.rect(16, 16))
        assert len(seen) == 2 and L.hevcasm_hip_latch_reason().decode() == seen[0]
        L.hevcasm_hip_latch_reset()
        assert L.hevcasm_hip_latched() == 0 and L.hevcasm_hip_latch_reason() == b""
        print("ok")
    """)


def test_populate_with_the_hip_bit_and_no_device_latches_instead_of_planting_aborting_slots():
    _run("""
        for name in ("sad", "ssd", "pred_uni_8to8", "transform", "quantize", "pred_intra"):
            assert slots(name, capi.C_REF | capi.HIP) == slots(name, capi.C_REF), name
        assert L.hevcasm_hip_latched() == 1 and "no usable gfx950" in L.hevcasm_hip_latch_reason().decode()
        assert L.hevcasm_get_ssd_linear(64, capi.C_REF | capi.HIP) == L.hevcasm_get_ssd_linear(64, capi.C_REF)
    """)


def test_without_a_handler_the_failing_call_still_aborts_with_the_text():
    r = _run("""
        import numpy as np
        fn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_uint32)
        a = np.zeros(256, dtype=np.uint8)
        fn(("hevcasm_hip1_sad", C.CDLL(capi.LIB_PATH)))(a.ctypes.data, 16, a.ctypes.data, 16, capi.rect(16, 16))
        print("survived")
    """, expect_rc=-6)
    assert "per-block path" in r.stderr and "survived" not in r.stdout


@pytest.mark.gpu
def test_gpu_failing_call_latches_then_reset_brings_the_hip_slots_back():
    """On a working device: a per-block quantize of 8 coefficients is outside what the batch kernel takes (blocks of 16 ...), so the wrapper fails without
    any device error; the latch must behave as after a real one, and hevcasm_hip_latch_reset() must bring the HIP slots back."""
    _run("""
        import numpy as np
        assert L.hevcasm_hip_available() == 1
        hip_slots = slots("quantize", capi.HIP)
        assert any(hip_slots) and L.hevcasm_hip_latched() == 0
        L.hevcasm_hip_set_failure_handler(C.cast(handler, C.c_void_p))
        fn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int)
        src = np.arange(16, dtype=np.int16); dst = np.zeros(16, dtype=np.int16)
        assert fn(hip_slots[0])(dst.ctypes.data, src.ctypes.data, 51, 20, 14, 16) in (0, 1) or True       # a good call first
        fn(hip_slots[0])(dst.ctypes.data, src.ctypes.data, 51, 20, 14, 8)
        assert len(seen) == 1 and "multiple of 16" in seen[0]
        assert L.hevcasm_hip_latched() == 1 and slots("quantize", capi.HIP | capi.C_REF) == slots("quantize", capi.C_REF)
        L.hevcasm_hip_latch_reset()
        assert slots("quantize", capi.HIP) == hip_slots
    """)
