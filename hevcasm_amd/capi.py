"""ctypes binding of lib/libhevcasm_hip.so: exactly the symbols include/*.h declare."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HEVCASM_HIP_LIB: another build of the same library (kernel A/B measurements, tools/ab.py); never set in tests or bench.py
LIB_PATH = os.environ.get("HEVCASM_HIP_LIB") or os.path.join(_HERE, "lib", "libhevcasm_hip.so")
EXE_PATH = os.path.join(_HERE, "lib", "hevcasm")

EINVAL = -1
ENODEV = -2

_vp, _sz, _pd, _i, _u32 = C.c_void_p, C.c_size_t, C.c_ssize_t, C.c_int, C.c_uint32

# name -> (restype, argtypes); every hevcasm_hip_* declaration of include/hevcasm_hip.h
HIP_API = {
    "hevcasm_hip_device_count": (_i, []),
    "hevcasm_hip_available": (_i, []),
    "hevcasm_hip_set_device": (_i, [_i]),
    "hevcasm_hip_last_error": (C.c_char_p, []),
    "hevcasm_hip_set_failure_handler": (_vp, [_vp]),
    "hevcasm_hip_latched": (_i, []),
    "hevcasm_hip_latch_reason": (C.c_char_p, []),
    "hevcasm_hip_latch_reset": (None, []),
    "hevcasm_hip_malloc": (_i, [C.POINTER(_vp), _sz]),
    "hevcasm_hip_free": (_i, [_vp]),
    "hevcasm_hip_upload": (_i, [_vp, _vp, _sz, _vp]),
    "hevcasm_hip_download": (_i, [_vp, _vp, _sz, _vp]),
    "hevcasm_hip_stream_create": (_i, [C.POINTER(_vp)]),
    "hevcasm_hip_stream_destroy": (_i, [_vp]),
    "hevcasm_hip_stream_synchronize": (_i, [_vp]),
    "hevcasm_hip_sad_batch": (_i, [_vp, _pd, _vp, _pd, _vp, _vp, _vp, _u32, _sz, _vp]),
    "hevcasm_hip_sad_multiref_batch": (_i, [_vp, _pd, _vp, _pd, _vp, _vp, _vp, _u32, _sz, _vp]),
    "hevcasm_hip_ssd_batch": (_i, [_vp, _pd, _vp, _pd, _vp, _vp, _vp, _i, _i, _sz, _vp]),
    "hevcasm_hip_sad_runs": (_i, [_vp, _pd, _vp, _pd, _vp, _vp, _vp, _vp, _i, _vp]),
    "hevcasm_hip_sad_multiref_runs": (_i, [_vp, _pd, _vp, _pd, _vp, _vp, _vp, _vp, _i, _vp]),
    "hevcasm_hip_ssd_runs": (_i, [_vp, _pd, _vp, _pd, _vp, _vp, _vp, _vp, _i, _vp]),
    "hevcasm_hip_me_full_search_runs": (_i, [_vp, _pd, _vp, _pd, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "hevcasm_hip_subpel_refine_runs": (_i, [_vp, _pd, _vp, _pd, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "hevcasm_hip_hadamard_satd_batch": (_i, [_vp, _pd, _vp, _pd, _vp, _vp, _vp, _i, _sz, _vp]),
    "hevcasm_hip_pred_uni_8to8_batch": (_i, [_vp, _pd, _vp, _pd, _vp, _vp, _vp, _i, _i, _i, _sz, _vp]),
    "hevcasm_hip_pred_bi_8to8_batch": (_i, [_vp, _pd, _vp, _vp, _pd, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _sz, _vp]),
    "hevcasm_hip_pred_uni_8to8_batch_2planes": (_i, [_vp, _vp, _pd, _vp, _vp, _pd, _vp, _vp, _vp, _i, _i, _i, _sz, _vp]),
    "hevcasm_hip_pred_bi_8to8_batch_2planes": (_i, [_vp, _vp, _pd, _vp, _vp, _vp, _vp, _pd, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _sz, _vp]),
    "hevcasm_hip_pred_uni_8to8_runs": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "hevcasm_hip_pred_bi_8to8_runs": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hevcasm_hip_inverse_transform_add_batch": (_i, [_vp, _pd, _vp, _pd, _vp, _vp, _vp, _i, _i, _sz, _vp]),
    "hevcasm_hip_dequant_inverse_transform_add_batch": (_i, [_vp, _pd, _vp, _pd, _vp, _vp, _vp, _i, _i, _i, _i, _sz, _vp]),
    "hevcasm_hip_transform_batch": (_i, [_vp, _vp, _pd, _vp, _i, _i, _sz, _vp]),
    "hevcasm_hip_quantize_inverse": (_i, [_vp, _vp, _i, _i, _sz, _vp]),
    "hevcasm_hip_quantize_batch": (_i, [_vp, _vp, _i, _i, _i, _i, _sz, _vp, _vp]),
    "hevcasm_hip_quantize_reconstruct_batch": (_i, [_vp, _pd, _vp, _pd, _vp, _vp, _vp, _i, _sz, _vp]),
    "hevcasm_hip_transform_runs": (_i, [_vp, _vp, _pd, _vp, _vp, _i, _vp]),
    "hevcasm_hip_quantize_runs": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "hevcasm_hip_inverse_transform_add_runs": (_i, [_vp, _pd, _vp, _pd, _vp, _vp, _vp, _vp, _i, _vp]),
    "hevcasm_hip_dequant_inverse_transform_add_runs": (_i, [_vp, _pd, _vp, _pd, _vp, _vp, _vp, _vp, _i, _vp]),
    "hevcasm_hip_transform_quantize_reconstruct_planes": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "hevcasm_hip_quantize_reconstruct_runs": (_i, [_vp, _pd, _vp, _pd, _vp, _vp, _vp, _vp, _i, _vp]),
    "hevcasm_hip_transform_quantize_reconstruct_runs": (_i, [_vp, _pd, _vp, _pd, _vp, _pd, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "hevcasm_hip_me_full_search_batch": (_i, [_vp, _pd, _vp, _pd, _vp, _vp, _vp, _u32, _i, _sz, _vp]),
    "hevcasm_hip_me_full_search_frame_ctus": (_i, [_vp, _pd, _vp, _pd, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "hevcasm_hip_me_full_search_frames_ctus": (_i, [_vp, _pd, _pd, _vp, _pd, _pd, _sz, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "hevcasm_hip_me_full_search_frame_rows": (_i, [_vp, _pd, _vp, _pd, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "hevcasm_hip_subpel_refine_batch": (_i, [_vp, _pd, _vp, _pd, _vp, _vp, _vp, _u32, _i, _sz, _vp]),
    "hevcasm_hip_pred_intra_batch": (_i, [_vp, _vp, _pd, _i, _i, _sz, _vp]),
    "hevcasm_hip_pred_intra_modes_batch": (_i, [_vp, _vp, _pd, _vp, _i, _sz, _vp]),
    "hevcasm_hip_intra_filter_neighbours_batch": (_i, [_vp, _pd, _vp, _pd, _i, _i, _sz, _vp]),
    "hevcasm_hip_intra_frames_workspace_bytes": (_sz, [_i, _i, _i, _sz]),
    "hevcasm_hip_intra_reconstruct_frames": (_i, [_vp, _pd, _sz, _vp, _pd, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _sz, _vp, _vp]),
    "hevcasm_hip_intra_frames_status": (_i, [_vp, _vp]),
    "hevcasm_hip_ssd_linear_batch": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _sz, _vp]),
    "hevcasm_hip_frame_sse_batch": (_i, [_vp, _pd, _vp, _pd, _vp, _vp, _vp, _i, _i, _sz, _vp]),
}

# the reference's exported API, re-exported by this library
_test_fn = (None, [C.POINTER(_i), _i])
HOST_API = {
    "hevcasm_instruction_set_support": (_i, []),
    "hevcasm_instruction_set_hip": (_i, []),
    "hevcasm_print_instruction_set_support": (None, [_vp, _i]),
    "hevcasm_main": (_i, [_i, _vp]),
    "hevcasm_test": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i]),
    "hevcasm_count_average_cycles": (_i, [_vp, _vp, _vp, _vp, C.POINTER(C.c_double), _i, _i]),
    "hevcasm_populate_sad": (None, [_vp, _i]),
    "hevcasm_populate_sad_multiref": (None, [_vp, _i]),
    "hevcasm_populate_ssd": (None, [_vp, _i]),
    "hevcasm_populate_hadamard_satd": (None, [_vp, _i]),
    "hevcasm_populate_pred_uni_8to8": (None, [_vp, _i]),
    "hevcasm_populate_pred_bi_8to8": (None, [_vp, _i]),
    "hevcasm_populate_inverse_transform_add": (None, [_vp, _i, _i]),
    "hevcasm_populate_transform": (None, [_vp, _i]),
    "hevcasm_populate_quantize_inverse": (None, [_vp, _i]),
    "hevcasm_populate_quantize": (None, [_vp, _i]),
    "hevcasm_populate_quantize_reconstruct": (None, [_vp, _i]),
    "hevcasm_populate_pred_intra": (None, [_vp, _i]),
    "hevcasm_populate_pred_intra_all_modes": (None, [_vp, _i]),
    "hevcasm_get_ssd_linear": (_vp, [_i, _i]),
    "hevcasm_test_sad": _test_fn,
    "hevcasm_test_sad_multiref": _test_fn,
    "hevcasm_test_ssd": _test_fn,
    "hevcasm_test_hadamard_satd": _test_fn,
    "hevcasm_test_pred_uni": _test_fn,
    "hevcasm_test_pred_bi": _test_fn,
    "hevcasm_test_inverse_transform_add": _test_fn,
    "hevcasm_test_transform": _test_fn,
    "hevcasm_test_quantize_inverse": _test_fn,
    "hevcasm_test_quantize": _test_fn,
    "hevcasm_test_quantize_reconstruct": _test_fn,
    "hevcasm_test_pred_intra": _test_fn,
    "hevcasm_test_ssd_linear": _test_fn,
}

# instruction-set bits (include/hevcasm.h)
C_REF, C_OPT, SSE2, SSE3, SSSE3, SSE41, SSE42, AVX, AVX2, HIP = (1 << i for i in range(10))

_LIB = None


class HevcasmError(RuntimeError):
    pass


def lib():
    """Load the shared library (once).  Raises if it has not been built: there is no fallback."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise HevcasmError(
                "%s is missing: build it with `make -C hevcasm_amd/csrc` or __graft_entry__.build()" % LIB_PATH)
        # A process must hold ONE HIP runtime.  torch ships its own libamdhip64; if this library were loaded first it
        # would bring in /opt/rocm's, torch would then add its own, and the second one to initialise sees no device
        # (hevcasm_hip_available() == 0 with an empty error string).  So when torch is installed it goes first and this
        # library binds to the runtime already in the process.  A C consumer without Python is not affected.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        handle = C.CDLL(LIB_PATH)
        for table in (HIP_API, HOST_API):
            for name, (res, args) in table.items():
                fn = getattr(handle, name)  # AttributeError if a declared symbol is not exported
                fn.restype = res
                fn.argtypes = args
        _LIB = handle
    return _LIB


def last_error():
    return lib().hevcasm_hip_last_error().decode()


def check(rc, what=""):
    if rc != 0:
        raise HevcasmError("%s failed with code %d: %s" % (what or "hevcasm_hip call", rc, last_error()))


def rect(w, h):
    return (w << 8) | h
