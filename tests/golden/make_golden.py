#!/usr/bin/env python3
"""Generates the committed fixtures of tests/golden/ from the REFERENCE's own C path.

Run in the build container (needs oracle/_ref/refdrv, i.e. /root/reference):
    python tests/golden/make_golden.py

Two files are written:
  golden_small.npz    a compact set of cases per family WITH their full inputs and the
                      reference's outputs (tiny frames, a handful of blocks each);
  golden_hashes.json  for every case of tests/cases.py: sha256 of the inputs (which are a pure
                      function of the seed) and sha256 of the reference's outputs.
Both hold data only: inputs and expected outputs.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import cases  # noqa: E402
from hevcasm_amd import workload as wl  # noqa: E402
from oracle import pyref  # noqa: E402


def sha(arrays):
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode() + str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def case_inputs(c):
    return [np.asarray(c[k]) for k in sorted(c)]


def small_cases():
    out = []
    for (w, h) in [(64, 64), (48, 64), (32, 24), (16, 16), (12, 16), (8, 8), (4, 8), (24, 32), (16, 4)]:
        out.append(("sad_%dx%d" % (w, h), "sad", wl.sad_case(w, h, 6, 11 + w, frame=(2 * w, 2 * h))))
        out.append(("sad4_%dx%d" % (w, h), "sad4", wl.sad_case(w, h, 6, 12 + w, ways=4, frame=(2 * w, 2 * h))))
    for s in (4, 8, 16, 32, 64):
        out.append(("ssd_%d" % s, "ssd", wl.sad_case(s, s, 6, 21 + s, frame=(2 * s, 2 * s))))
    out += [x for x in cases.satd_cases(n=12) if x[0] in ("satd_2", "satd_4", "satd_8", "satd_8_max")]
    for taps, sizes in ((8, [(64, 64), (16, 16), (12, 16), (8, 4)]), (4, [(32, 32), (8, 8), (6, 8), (2, 4)])):
        for (w, h) in sizes:
            out.append(("uni%d_%dx%d" % (taps, w, h), "pred_uni", wl.pred_case(taps, w, h, 8, 31 + w, frame=(2 * w, 2 * h))))
            if (w, h) not in ((8, 4), (2, 4)):
                out.append(("bi%d_%dx%d" % (taps, w, h), "pred_bi", wl.pred_case(taps, w, h, 6, 41 + w, bi=True, frame=(2 * w, 2 * h))))
    for tr, lg in wl.TRANSFORMS:
        tag = "dst4" if tr else "dct%d" % (1 << lg)
        s = 1 << lg
        out.append(("inv_%s_full" % tag, "inverse", wl.inverse_case(tr, lg, 6, 51 + lg, "full", frame=(2 * s, 2 * s))))
        out.append(("inv_%s_natural" % tag, "inverse", wl.inverse_case(tr, lg, 6, 52 + lg, "natural", frame=(2 * s, 2 * s))))
        c = wl.inverse_case(tr, lg, 6, 53 + lg, "natural", frame=(2 * s, 2 * s))
        c["scale"], c["shift"] = 51, 14
        out.append(("deqinv_%s" % tag, "dequant_inverse", c))
        out.append(("fwd_%s_res9" % tag, "forward", wl.forward_case(tr, lg, 6, 54 + lg, "residual9", frame=(2 * s, 2 * s))))
        out.append(("fwd_%s_full" % tag, "forward", wl.forward_case(tr, lg, 6, 55 + lg, "full", frame=(2 * s, 2 * s))))
    out += [x for x in cases.quant_cases() if x[0] in ("deq_16_ref", "deq_1024_full", "q_16_ref", "q_256_hevc", "q_1024_min", "q_64_sparse", "rec_4", "rec_32_full", "deq_tail")]
    out += [x for x in cases.me_cases() if x[0] in ("me_16x16_r8", "me_ties", "me_32x32_r37", "me_64x48_r64_ties", "me_64x64_r64_double")]
    out += [x for x in cases.subpel_cases() if x[0] in ("subpel_8x8", "subpel_smooth_16x16", "subpel_ties", "subpel_satd_16x16", "subpel_satd_12x16")]
    out += [x for x in cases.intra_cases() if x[0].endswith("_s160")]
    out += [x for x in cases.ssd_linear_cases() if x[0] in ("ssdlin_17", "ssdlin_512", "ssdlin_1023", "ssdlin_4099")]
    out += [x for x in cases.frame_sse_cases() if x[0] in ("fsse_17x33", "fsse_257x5", "fsse_bottom_up")]
    out += [x for x in cases.tu_chain_cases() if x[0] in ("tuchain_white_on_black_64x64_qp0", "tuchain_video_4x4_qp20", "tuchain_noise_96x160_qp12")]      # round 6: the residual loop
    return out


def main():
    if not pyref.available():
        sys.exit("oracle/_ref/refdrv missing: run `make -C oracle` in a container that has /root/reference")
    store, index = {}, []
    with pyref.RefProc() as ref:
        for name, family, c in small_cases():
            outs = cases.run_reference(ref, family, c)
            for k, v in c.items():
                store["%s/in/%s" % (name, k)] = np.asarray(v)
            for i, o in enumerate(outs):
                store["%s/out/%d" % (name, i)] = o
            index.append([name, family, len(outs)])
        store["__index__"] = np.array(json.dumps(index))
        np.savez_compressed(os.path.join(HERE, "golden_small.npz"), **store)

        hashes = {}
        for group in sorted(cases.ALL_CASE_LISTS):
            for name, family, c in cases.ALL_CASE_LISTS[group]():
                outs = cases.run_reference(ref, family, c)
                hashes[name] = dict(family=family, inputs=sha(case_inputs(c)), outputs=sha(outs))
        with open(os.path.join(HERE, "golden_hashes.json"), "w") as f:
            json.dump(hashes, f, indent=0, sort_keys=True)
    print("golden_small.npz: %d cases; golden_hashes.json: %d cases" % (len(index), len(hashes)))


if __name__ == "__main__":
    main()
