#!/usr/bin/env python3
"""Throughput of the non-SAD hot-path families on one MI355X (BASELINE.json configs #3, #4, #5 and
the quantiser): Mblocks/s and algorithmic GB/s (SURVEY.md 8d byte counts) per kernel, measured
with HIP events over cold, device-resident inputs.  Prints one JSON line per measurement.

    python tools/bench_families.py [pred] [transform] [quant] [me]
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hevcasm_amd import batch as B, workload as wl

PEAK = 8000.0


def timed(fn, reps=5):
    ts = []
    for i in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(i); e1.record(); torch.cuda.synchronize()
        if i: ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


SINK = None      # bench.py collects the rows instead of printing them


def report(name, n, bytes_per_block, ms, **kw):
    gbs = n * bytes_per_block / (ms * 1e-3) / 1e9
    row = dict(kernel=name, blocks=n, ms=round(ms, 4), mblocks_per_s=round(n / ms / 1e3, 1), bytes_per_block=bytes_per_block,
               algorithmic_gb_per_s=round(gbs, 1), frac_of_hbm_peak=round(gbs / PEAK, 4), **kw)
    if SINK is not None:
        SINK.append(row)
    else:
        print(json.dumps(row), flush=True)


def bench_pred(frames=64, only=None, two_planes=True, taps_only=None, mv_region=0):
    """Config #3: 1920x1080 luma tiled by each PU size (chroma planes 960x540, 4-tap), all phases,
    `frames` frames per launch so that the launch is far above the launch floor."""
    for taps, fw, fh in ((8, 1920, 1080), (4, 960, 540)):
        if taps_only and taps != taps_only:
            continue
        for (w, h) in wl.PU_SIZES:
            if only and (w, h) not in only:
                continue
            pw, ph = w * taps // 8, h * taps // 8
            n = (fw // pw) * (fh // ph) * frames
            for bi in (False, True):
                if bi and (w, h) in ((8, 4), (4, 8)):
                    continue
                # distinct inputs per repetition so that nothing is served from the Infinity Cache
                cs = [wl.pred_case(taps, pw, ph, n, 50 + i, bi=bi, frame=(fw, fh), device="cuda", mv_region=mv_region * taps // 8) for i in range(3)]
                devs = []
                for c in cs:
                    d = dict(dst=torch.zeros(c["dst_bytes"], dtype=torch.uint8, device="cuda"), dst_off=B.off32(c["dst_off"]))
                    for r in range(2 if bi else 1):
                        d["ref%d" % r] = c["ref%d" % r]; d["ref%d_off" % r] = B.off32(c["ref%d_off" % r]); d["frac%d" % r] = B.u8(c["frac%d" % r])
                    devs.append(d)
                c0 = cs[0]

                def run(i):
                    d = devs[i % len(devs)]
                    if bi:
                        B.pred_bi(d["dst"], c0["stride_dst"], d["ref0"], d["ref1"], c0["stride_ref"], d["dst_off"], d["ref0_off"], d["ref1_off"], d["frac0"], d["frac1"], taps, pw, ph)
                    else:
                        B.pred_uni(d["dst"], c0["stride_dst"], d["ref0"], c0["stride_ref"], d["dst_off"], d["ref0_off"], d["frac0"], taps, pw, ph)
                ms = timed(run, reps=4)
                refs = 2 if bi else 1
                bytes_per_pu = refs * (pw + taps - 1) * (ph + taps - 1) + pw * ph + 4 * (1 + refs) + refs
                report("pred_%s_%dtap" % ("bi" if bi else "uni", taps), n, bytes_per_pu, ms, size="%dx%d" % (pw, ph), frames=frames)
                if taps == 4 and two_planes:
                    # Cb and Cr of the same frames in ONE launch (same PU list and motion vectors, hevcasm_hip_pred_*_batch_2planes): 2 n PUs;
                    # the descriptors are read once per plane, so the bytes per PU are the same
                    second = [dict(dst=torch.zeros_like(d["dst"]), **{k: wl.fill_u8(d[k].numel(), 900 + i, device="cuda") for k in d if k in ("ref0", "ref1")}) for i, d in enumerate(devs)]

                    def run2(i):
                        d, e = devs[i % len(devs)], second[i % len(devs)]
                        if bi:
                            B.pred_bi_2planes(d["dst"], e["dst"], c0["stride_dst"], d["ref0"], e["ref0"], d["ref1"], e["ref1"], c0["stride_ref"], d["dst_off"], d["ref0_off"], d["ref1_off"], d["frac0"], d["frac1"], taps, pw, ph)
                        else:
                            B.pred_uni_2planes(d["dst"], e["dst"], c0["stride_dst"], d["ref0"], e["ref0"], c0["stride_ref"], d["dst_off"], d["ref0_off"], d["frac0"], taps, pw, ph)
                    ms2 = timed(run2, reps=4)
                    report("pred_%s_%dtap Cb+Cr" % ("bi" if bi else "uni", taps), 2 * n, bytes_per_pu, ms2, size="%dx%d" % (pw, ph), frames=frames)
                    del second
                del cs, devs
                torch.cuda.empty_cache()


def bench_pred_latency(reps=30):
    """Config #3 at its literal size -- ONE 1920x1080 4:2:0 frame: luma tiled by 16x16 PUs (8-tap), Cb and Cr by 8x8 (4-tap), all
    phases -- as a codec that predicts frame by frame sees it (hard parts): (a) one plane per launch, launches
    issued back to back, HIP events around each; (b) the frame's three launches (luma + Cb + Cr) captured once into a HIP graph
    and replayed, events around `reps` replays.  Inputs rotate over 3 frames; a frame (3 MB + 3 MB) sits in the Infinity Cache
    after its first use, which is also what a decoder's reference frames do."""
    planes = (("luma", 8, 1920, 1080, 16, 16), ("cb", 4, 960, 540, 8, 8), ("cr", 4, 960, 540, 8, 8))
    for bi in (False, True):
        refs = 2 if bi else 1
        sets = []
        for k in range(3):
            one = []
            for j, (name, taps, fw, fh, pw, ph) in enumerate(planes):
                n = (fw // pw) * (fh // ph)
                c = wl.pred_case(taps, pw, ph, n, 700 + 10 * k + j, bi=bi, frame=(fw, fh), device="cuda")
                d = dict(c=c, n=n, taps=taps, pw=pw, ph=ph, dst=torch.zeros(c["dst_bytes"], dtype=torch.uint8, device="cuda"), dst_off=B.off32(c["dst_off"]))
                for r in range(refs):
                    d["ref%d" % r] = c["ref%d" % r]; d["ref%d_off" % r] = B.off32(c["ref%d_off" % r]); d["frac%d" % r] = B.u8(c["frac%d" % r])
                one.append(d)
            sets.append(one)

        def launch(d):
            c = d["c"]
            if bi:
                B.pred_bi(d["dst"], c["stride_dst"], d["ref0"], d["ref1"], c["stride_ref"], d["dst_off"], d["ref0_off"], d["ref1_off"], d["frac0"], d["frac1"], d["taps"], d["pw"], d["ph"])
            else:
                B.pred_uni(d["dst"], c["stride_dst"], d["ref0"], c["stride_ref"], d["dst_off"], d["ref0_off"], d["frac0"], d["taps"], d["pw"], d["ph"])

        def bytes_of(d):
            return refs * (d["pw"] + d["taps"] - 1) * (d["ph"] + d["taps"] - 1) + d["pw"] * d["ph"] + 4 * (1 + refs) + refs

        for one in sets:                       # warm-up: code objects, first touch
            for d in one:
                launch(d)
        torch.cuda.synchronize()
        # (a) one plane per launch
        for j in (0, 1):
            ts = []
            for i in range(reps):
                d = sets[i % 3][j]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); launch(d); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            d = sets[0][j]
            report("pred_%s_%dtap" % ("bi" if bi else "uni", d["taps"]), d["n"], bytes_of(d), float(np.median(ts)), size="%dx%d" % (d["pw"], d["ph"]), frames=1,
                   note="one %s plane of one 1080p frame per launch, HIP events around the launch" % planes[j][0])
        # (b) luma + (Cb and Cr in one launch: same PU list, hevcasm_hip_pred_*_batch_2planes) as one graph
        def launch_chroma(cb, cr):
            c = cb["c"]
            if bi:
                B.pred_bi_2planes(cb["dst"], cr["dst"], c["stride_dst"], cb["ref0"], cr["ref0"], cb["ref1"], cr["ref1"], c["stride_ref"], cb["dst_off"], cb["ref0_off"], cb["ref1_off"],
                                  cb["frac0"], cb["frac1"], cb["taps"], cb["pw"], cb["ph"])
            else:
                B.pred_uni_2planes(cb["dst"], cr["dst"], c["stride_dst"], cb["ref0"], cr["ref0"], c["stride_ref"], cb["dst_off"], cb["ref0_off"], cb["frac0"], cb["taps"], cb["pw"], cb["ph"])
        graphs = []
        side = torch.cuda.Stream()
        for one in sets:
            launch_chroma(one[1], one[2])      # warm-up outside capture
            g = torch.cuda.CUDAGraph()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.graph(g, stream=side):
                launch(one[0])
                launch_chroma(one[1], one[2])
            graphs.append(g)
        torch.cuda.synchronize()
        for g in graphs:
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            graphs[i % 3].replay()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        n_all = sum(d["n"] for d in sets[0])
        # (b') the same uniform frame through the one-launch entry point (3 runs: luma 16x16, Cb 8x8, Cr 8x8): what the run lookup and the general kernel form cost
        def uniform_runs(i):
            one = sets[i % 3]
            lp, cp = one[0], one[1]
            n_l, n_c = lp["n"], cp["n"]
            key = "runs_arrays"
            if key not in lp:
                cat = lambda k: torch.cat([lp[k], cp[k]])
                lp[key] = dict(dst_off=cat("dst_off"), ref0_off=cat("ref0_off"), frac0=cat("frac0"))
                if bi:
                    lp[key].update(ref1_off=cat("ref1_off"), frac1=cat("frac1"))
                lp["runs"] = np.array([(0, n_l, 8, 16, 16, 0), (n_l, n_c, 4, 8, 8, 1), (n_l, n_c, 4, 8, 8, 2)], dtype=wl.PRED_RUN_DTYPE)
            a = lp[key]
            pl = [(d["dst"], d["c"]["stride_dst"], d["ref0"], d["ref1"] if bi else None, d["c"]["stride_ref"]) for d in one]
            if bi:
                B.pred_runs(pl, lp["runs"], a["dst_off"], a["ref0_off"], a["frac0"], ref1_off=a["ref1_off"], frac1=a["frac1"])
            else:
                B.pred_runs(pl, lp["runs"], a["dst_off"], a["ref0_off"], a["frac0"])
        for i in range(3):
            uniform_runs(i)
        torch.cuda.synchronize()
        report("pred_%s uniform frame, one launch" % ("bi" if bi else "uni"), n_all, sum(bytes_of(d) * d["n"] for d in sets[0]) // n_all, timed(uniform_runs, reps), size="16x16 luma + 2 x 8x8 chroma",
               frames=1, launches=1, note="the frame of the graph row as 3 runs of hevcasm_hip_pred_*_runs: one launch, HIP events around it")
        report("pred_%s frame graph" % ("bi" if bi else "uni"), n_all, sum(bytes_of(d) * d["n"] for d in sets[0]) // n_all, ms, size="16x16 luma + 2 x 8x8 chroma", frames=1, launches=2,
               note="one 1080p 4:2:0 frame = 2 launches (luma; Cb + Cr together) captured in one HIP graph, %d replays back to back; %.0f frames/s" % (reps, 1e3 / ms))
        del sets, graphs
        torch.cuda.empty_cache()
        # (c) round 6: a REAL partition -- a quad-tree over every PU size of pred_inter.c:438-444, luma + Cb + Cr -- as ONE launch (hevcasm_hip_pred_*_runs),
        # next to what it replaces: one launch per (size, plane), issued back to back.  Three frames rotate, as above.
        frames3 = []
        for k in range(3):
            c = wl.pred_frame_case(1920, 1080, 900 + k, bi=bi)
            dev_planes = [(torch.zeros(p["dst_bytes"], dtype=torch.uint8, device="cuda"), p["stride_dst"], B.u8(p["ref0"]), B.u8(p["ref1"]) if bi else None, p["stride_ref"]) for p in c["planes"]]
            arrays = dict(dst_off=B.off32(c["dst_off"]), ref0_off=B.off32(c["ref0_off"]), frac0=B.u8(c["frac0"]), ref1_off=B.off32(c["ref1_off"]), frac1=B.u8(c["frac1"]))
            frames3.append((c, dev_planes, arrays))

        def one_launch(i):
            c, dev_planes, a = frames3[i % 3]
            if bi:
                B.pred_runs(dev_planes, c["runs"], a["dst_off"], a["ref0_off"], a["frac0"], ref1_off=a["ref1_off"], frac1=a["frac1"])
            else:
                B.pred_runs(dev_planes, c["runs"], a["dst_off"], a["ref0_off"], a["frac0"])

        def per_run_launches(i):
            c, dev_planes, a = frames3[i % 3]
            for r in c["runs"]:
                dst, sd, r0, r1, sr = dev_planes[int(r["plane"])]
                s = slice(int(r["first"]), int(r["first"]) + int(r["count"]))
                if bi:
                    B.pred_bi(dst, sd, r0, r1, sr, a["dst_off"][s], a["ref0_off"][s], a["ref1_off"][s], a["frac0"][s], a["frac1"][s], int(r["taps"]), int(r["w"]), int(r["h"]))
                else:
                    B.pred_uni(dst, sd, r0, sr, a["dst_off"][s], a["ref0_off"][s], a["frac0"][s], int(r["taps"]), int(r["w"]), int(r["h"]))

        c0 = frames3[0][0]
        bytes_frame = sum(int(r["count"]) * (refs * (int(r["w"]) + int(r["taps"]) - 1) * (int(r["h"]) + int(r["taps"]) - 1) + int(r["w"]) * int(r["h"]) + 4 * (1 + refs) + refs) for r in c0["runs"])
        n_pus = int(sum(int(r["count"]) for r in c0["runs"]))
        for name, fn, launches in (("frame partition, one launch", one_launch, 1), ("frame partition, a launch per (size, plane)", per_run_launches, len(c0["runs"]))):
            for i in range(3):
                fn(i)
            torch.cuda.synchronize()
            report("pred_%s %s" % ("bi" if bi else "uni", name), n_pus, bytes_frame // n_pus, timed(fn, reps), size="%d sizes x 3 planes" % (len(c0["runs"]) // 3), frames=1, launches=launches,
                   note="one 1080p 4:2:0 frame cut by a random quad-tree into PUs of every size, all three planes; HIP events around the frame")
        del frames3
        torch.cuda.empty_cache()


def bench_cost_latency(reps=30):
    """Round 6: the block costs of ONE 1080p frame's partition -- a quad-tree over every PU size of pred_inter.c:438-444, one candidate (sad, ssd) or four
    (sad_multiref, +-7 pixels) per PU -- as ONE launch (hevcasm_hip_*_runs) next to what it replaces: a launch per size.  Three frames rotate."""
    frames3 = []
    for k in range(3):
        c = wl.cost_frame_case(1920, 1080, 950 + k)
        frames3.append((c, dict(src=B.u8(c["src"]), ref=B.u8(c["ref"]), src_off=B.off32(c["src_off"]), ref_off=B.off32(c["ref_off"]), ref_off4=B.off32(c["ref_off4"]),
                                out=torch.zeros(c["n_pus"], dtype=torch.int32, device="cuda"), out4=torch.zeros((c["n_pus"], 4), dtype=torch.int32, device="cuda"))))
    c0 = frames3[0][0]
    for kind in ("sad", "sad4", "ssd"):
        ways = 4 if kind == "sad4" else 1

        def one_launch(i):
            c, d = frames3[i % 3]
            B.cost_runs(kind, d["src"], c["stride_src"], d["ref"], c["stride_ref"], d["src_off"], d["ref_off4"] if ways == 4 else d["ref_off"], c["runs"], out=d["out4"] if ways == 4 else d["out"])

        def per_size_launches(i):
            c, d = frames3[i % 3]
            for (w, h, first, count) in c["sizes"]:
                s = slice(first, first + count)
                if kind == "sad":
                    B.sad(d["src"], c["stride_src"], d["ref"], c["stride_ref"], d["src_off"][s], d["ref_off"][s], w, h, out=d["out"][s])
                elif kind == "ssd":
                    B.ssd(d["src"], c["stride_src"], d["ref"], c["stride_ref"], d["src_off"][s], d["ref_off"][s], w, h, out=d["out"][s])
                else:
                    B.sad_multiref(d["src"], c["stride_src"], d["ref"], c["stride_ref"], d["src_off"][s], d["ref_off4"][4 * first:4 * (first + count)], w, h, out=d["out4"][s])

        bytes_frame = sum(n * ((1 + ways) * w * h + 4 * (1 + 2 * ways)) for (w, h, _, n) in c0["sizes"])
        for name, fn, launches in (("frame partition, one launch", one_launch, 1), ("frame partition, a launch per size", per_size_launches, len(c0["sizes"]))):
            for i in range(3):
                fn(i)
            torch.cuda.synchronize()
            report("%s %s" % ({"sad": "sad", "sad4": "sad_multiref_4", "ssd": "ssd"}[kind], name), c0["n_pus"], bytes_frame // c0["n_pus"], timed(fn, reps), size="%d sizes" % len(c0["sizes"]), frames=1,
                   launches=launches, note="one 1080p frame cut by a random quad-tree into PUs of every size; HIP events around the frame")
    # the search (+-8) and the quarter-pel refinement of the same partition: one launch against a launch per size
    res = torch.zeros((max(c["n_pus"] for c, _ in frames3), 3), dtype=torch.int32, device="cuda")
    for kind in ("me_full_search +-8", "subpel_refine"):
        def one_launch(i):
            c, d = frames3[i % 3]
            if kind.startswith("me"):
                B.me_full_search_runs(d["src"], c["stride_src"], d["ref"], c["stride_ref"], d["src_off"], d["ref_off"], c["runs"], 8, out=res[:c["n_pus"]])
            else:
                B.subpel_refine_runs(d["src"], c["stride_src"], d["ref"], c["stride_ref"], d["src_off"], d["ref_off"], c["runs"], out=res[:c["n_pus"]])

        def per_size_launches(i):
            c, d = frames3[i % 3]
            for (w, h, first, count) in c["sizes"]:
                s = slice(first, first + count)
                if kind.startswith("me"):
                    B.me_full_search(d["src"], c["stride_src"], d["ref"], c["stride_ref"], d["src_off"][s], d["ref_off"][s], w, h, 8, out=res[s])
                else:
                    B.subpel_refine(d["src"], c["stride_src"], d["ref"], c["stride_ref"], d["src_off"][s], d["ref_off"][s], w, h, out=res[s])

        margin = 16 if kind.startswith("me") else 8
        bytes_frame = sum(n * (w * h + (w + margin) * (h + margin) + 8 + 12) for (w, h, _, n) in c0["sizes"])
        for name, fn, launches in (("frame partition, one launch", one_launch, 1), ("frame partition, a launch per size", per_size_launches, len(c0["sizes"]))):
            for i in range(3):
                fn(i)
            torch.cuda.synchronize()
            report("%s %s" % (kind, name), c0["n_pus"], bytes_frame // c0["n_pus"], timed(fn, reps), size="%d sizes" % len(c0["sizes"]), frames=1, launches=launches,
                   note="one 1080p frame cut by a random quad-tree into PUs of every size; HIP events around the frame; VALU-bound kernels, the GB/s column is not their roofline")
    del frames3
    torch.cuda.empty_cache()


def bench_tu_latency(reps=30):
    """Round 6: the transform stages of ONE 1080p frame's residual quad-tree (4x4 DST / DCT, 8x8, 16x16, 32x32; each size with its HEVC quantiser parameters) as
    ONE launch per stage (hevcasm_hip_transform_runs, _quantize_runs, _dequant_inverse_transform_add_runs) next to what they replace: a launch per size and stage.
    Three frames rotate."""
    frames3 = []
    for k in range(3):
        c = wl.tu_frame_case(1920, 1080, 970 + k, pad=64)
        frames3.append((c, dict(res=torch.from_numpy(c["residual"]).cuda(), pred=B.u8(c["pred"]), off=B.off32(c["off"]), coeffs=torch.zeros(c["n_coeffs"], dtype=torch.int16, device="cuda"),
                                levels=torch.zeros(c["n_coeffs"], dtype=torch.int16, device="cuda"), cbf=torch.zeros(c["n_tus"], dtype=torch.int32, device="cuda"),
                                dst=torch.zeros(c["stride"] * c["h"], dtype=torch.uint8, device="cuda"))))
    c0 = frames3[0][0]

    def stage_runs(stage):
        def fn(i):
            c, d = frames3[i % 3]
            if stage == "transform":
                B.transform_runs(d["res"], c["stride"], d["off"], c["runs_q"], c["n_coeffs"], out=d["coeffs"])
            elif stage == "quantize":
                B.quantize_runs(d["coeffs"], c["runs_q"], c["n_tus"], out=d["levels"], cbf=d["cbf"])
            else:
                B.inverse_transform_add_runs(d["dst"], c["stride"], d["pred"], c["stride"], d["off"], d["off"], d["levels"], c["runs_dq"], dequant=True)
        return fn

    def stage_per_size(stage):
        def fn(i):
            c, d = frames3[i % 3]
            for (log2, tr, first, count, cf), rq, rd in zip(c["sizes"], c["runs_q"], c["runs_dq"]):
                e = cf + (count << (2 * log2))
                if stage == "transform":
                    B.transform(d["res"], c["stride"], d["off"][first:first + count], tr, log2, out=d["coeffs"][cf:e])
                elif stage == "quantize":
                    B.quantize(d["coeffs"][cf:e], int(rq["scale"]), int(rq["shift"]), int(rq["offset"]), 1 << (2 * log2), out=d["levels"][cf:e], cbf=d["cbf"][first:first + count])
                else:
                    B.dequant_inverse_transform_add(d["dst"], c["stride"], d["pred"], c["stride"], d["off"][first:first + count], d["off"][first:first + count], d["levels"][cf:e],
                                                    int(rd["scale"]), int(rd["shift"]), tr, log2)
        return fn

    def chain(per_size):
        fns = [(stage_per_size if per_size else stage_runs)(s) for s in ("transform", "quantize", "dequant_inverse")]
        def fn(i):
            for f in fns:
                f(i)
        return fn

    px = 1920 * 1080
    rows = [("transform", 4 * px + 4 * c0["n_tus"]), ("quantize", 4 * px + 4 * c0["n_tus"]), ("dequant_inverse_transform_add", 4 * px + 8 * c0["n_tus"])]
    for (stage, bytes_frame) in rows:
        key = "dequant_inverse" if stage.startswith("dequant") else stage
        for name, fn, launches in (("frame TUs, one launch", stage_runs(key), 1), ("frame TUs, a launch per size", stage_per_size(key), len(c0["sizes"]))):
            for i in range(3):
                fn(i)
            torch.cuda.synchronize()
            report("%s %s" % (stage, name), c0["n_tus"], bytes_frame // c0["n_tus"], timed(fn, reps), size="%d sizes" % len(c0["sizes"]), frames=1, launches=launches,
                   note="one 1080p frame's residual quad-tree (%d TUs); HIP events around the frame" % c0["n_tus"])
    srcs = [B.u8(c["src"]) for c, _ in frames3]

    def fused(i):
        c, d = frames3[i % 3]
        B.transform_quantize_reconstruct_runs(d["dst"], c["stride"], srcs[i % 3], c["stride"], d["pred"], c["stride"], d["off"], d["off"], d["off"], c["runs_q"], c["runs_dq"], c["n_coeffs"], c["n_tus"],
                                              levels=d["levels"], cbf=d["cbf"])

    def residual_then_chain(i):
        c, d = frames3[i % 3]
        d["res"] = srcs[i % 3].to(torch.int16) - d["pred"].to(torch.int16)      # the step a codec does itself in front of hevcasm_transform
        chain_runs(i)

    chain_runs = chain(False)
    for i in range(3):
        fused(i)
    torch.cuda.synchronize()
    report("transform_quantize_reconstruct frame TUs, ONE launch (fused: residual, coefficients, levels stay in the workgroup)", c0["n_tus"], (5 * px + 16 * c0["n_tus"]) // c0["n_tus"], timed(fused, reps),
           size="%d sizes" % len(c0["sizes"]), frames=1, launches=1, note="source - prediction -> transform -> quantise -> dequantise -> inverse + add of one 1080p frame's TUs; 5 bytes per pixel")
    for i in range(3):
        residual_then_chain(i)
    torch.cuda.synchronize()
    report("residual (torch) + three stages, one launch each", c0["n_tus"], 16 * px // c0["n_tus"], timed(residual_then_chain, reps), size="%d sizes" % len(c0["sizes"]), frames=1, launches=6,
           note="what the fused launch replaces: int16 residual plane, coefficients and levels each cross HBM twice: 16 bytes per pixel")
    for name, fn, launches in (("three stages, one launch each", chain(False), 3), ("three stages, a launch per size and stage", chain(True), 3 * len(c0["sizes"]))):
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        report("transform+quantize+dequant_inverse %s" % name, c0["n_tus"], 12 * px // c0["n_tus"], timed(fn, reps), size="%d sizes" % len(c0["sizes"]), frames=1, launches=launches,
               note="forward transform -> quantise -> dequantise + inverse + add of one 1080p frame's TUs")
    del frames3, srcs
    torch.cuda.empty_cache()
    # throughput: the TUs of F frames in one call (a run per size over all frames), cold planes rotating; units inside a run in z-scan order and in raster order
    for order_name, craster in (("z-scan", c0), ("raster", wl.tu_frame_case(1920, 1080, 970, pad=64, raster=True))):
        _tu_throughput(craster, order_name, px)


def _tu_throughput(c, order_name, px):
    F = 32
    plane = c["stride"] * c["h"]
    off_parts, runs_q, runs_dq, first, coeff = [], [], [], 0, 0
    for (log2, tr, f0, count, cf), rq, rd in zip(c["sizes"], c["runs_q"], c["runs_dq"]):
        off_parts.append((c["off"][f0:f0 + count][None, :].astype(np.int64) + (np.arange(F, dtype=np.int64) * plane)[:, None]).reshape(-1))
        runs_q.append((first, count * F, coeff, log2, tr, 0, rq["scale"], rq["shift"], rq["offset"]))
        runs_dq.append((first, count * F, coeff, log2, tr, 0, rd["scale"], rd["shift"], 0))
        first += count * F
        coeff += (count * F) << (2 * log2)
    off = B.off32(np.concatenate(off_parts).astype(np.uint32))
    runs_q, runs_dq = np.array(runs_q, dtype=wl.TU_RUN_DTYPE), np.array(runs_dq, dtype=wl.TU_RUN_DTYPE)
    n_tus, n_coeffs = c["n_tus"] * F, c["n_coeffs"] * F
    sets = []
    for k in range(3):
        g = torch.Generator(device="cuda"); g.manual_seed(500 + k)
        src = torch.randint(0, 256, (plane * F,), generator=g, device="cuda", dtype=torch.uint8)
        pred = (src.to(torch.int16) + torch.randint(-12, 13, (plane * F,), generator=g, device="cuda", dtype=torch.int16)).clamp_(0, 255).to(torch.uint8)
        sets.append(dict(src=src, pred=pred, dst=torch.zeros(plane * F, dtype=torch.uint8, device="cuda"), res=torch.zeros(plane * F, dtype=torch.int16, device="cuda"),
                         coeffs=torch.zeros(n_coeffs, dtype=torch.int16, device="cuda"), levels=torch.zeros(n_coeffs, dtype=torch.int16, device="cuda"), cbf=torch.zeros(n_tus, dtype=torch.int32, device="cuda")))

    def fused_many(i):
        d = sets[i % 3]
        B.transform_quantize_reconstruct_runs(d["dst"], c["stride"], d["src"], c["stride"], d["pred"], c["stride"], off, off, off, runs_q, runs_dq, n_coeffs, n_tus, levels=d["levels"], cbf=d["cbf"])

    def stages_many(i):
        d = sets[i % 3]
        B.transform_runs(d["res"], c["stride"], off, runs_q, n_coeffs, out=d["coeffs"])
        B.quantize_runs(d["coeffs"], runs_q, n_tus, out=d["levels"], cbf=d["cbf"])
        B.inverse_transform_add_runs(d["dst"], c["stride"], d["pred"], c["stride"], off, off, d["levels"], runs_dq, dequant=True)

    def residual_stages_many(i):
        d = sets[i % 3]
        torch.sub(d["src"].to(torch.int16), d["pred"].to(torch.int16), out=d["res"])
        stages_many(i)

    pxF = px * F
    for name, fn, bytes_all, launches in (("transform_quantize_reconstruct, ONE launch (fused)", fused_many, 5 * pxF + 16 * n_tus, 1),
                                          ("three stages, one launch each (residual plane given)", stages_many, 12 * pxF + 16 * n_tus, 3),
                                          ("residual (torch) + three stages", residual_stages_many, 16 * pxF + 16 * n_tus, 6)):
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        ms = timed(fn, 9)
        report("%d frames' TUs (%s inside a run): %s" % (F, order_name, name), n_tus, bytes_all // n_tus, ms, size="%d sizes" % len(c["sizes"]), frames=F, launches=launches, mpixels_per_s=round(pxF / ms / 1e3, 1),
               note="1080p residual quad-trees of %d frames in one call, cold planes; bytes = what that form moves through HBM" % F)
    del sets
    torch.cuda.empty_cache()


def bench_tu_fused_sizes(frames=32):
    """The fused residual loop on planes cut into ONE TU size (1920x1088 x `frames` per call), next to the three stage launches on the same units: which size
    costs what.  Gpixel/s and the fraction of the HBM peak at 5 (fused) / 12 (three stages, residual given) bytes per pixel."""
    w, h, stride = 1920, 1088, 1920
    plane = stride * h
    q = wl.intra_qp(27)
    g = torch.Generator(device="cuda"); g.manual_seed(77)
    sets = []
    for k in range(3):
        src = torch.randint(0, 256, (plane * frames,), generator=g, device="cuda", dtype=torch.uint8)
        pred = (src.to(torch.int16) + torch.randint(-12, 13, (plane * frames,), generator=g, device="cuda", dtype=torch.int16)).clamp_(0, 255).to(torch.uint8)
        sets.append(dict(src=src, pred=pred, res=src.to(torch.int16) - pred.to(torch.int16), dst=torch.zeros_like(src), coeffs=torch.zeros(plane * frames, dtype=torch.int16, device="cuda"),
                         levels=torch.zeros(plane * frames, dtype=torch.int16, device="cuda")))
    for (log2, tr) in ((2, 1), (2, 0), (3, 0), (4, 0), (5, 0)):
        n = 1 << log2
        by, bx = np.meshgrid(np.arange(h // n), np.arange(w // n), indexing="ij")
        one = (by * n * stride + bx * n).reshape(-1).astype(np.int64)
        off = B.off32((one[None, :] + (np.arange(frames, dtype=np.int64) * plane)[:, None]).reshape(-1).astype(np.uint32))
        n_tus = one.size * frames
        qs, qsh, qo, ds, dsh = (int(v) for v in q[log2 - 2])
        rq = np.array([(0, n_tus, 0, log2, tr, 0, qs, qsh, qo)], dtype=wl.TU_RUN_DTYPE)
        rdq = np.array([(0, n_tus, 0, log2, tr, 0, ds, dsh, 0)], dtype=wl.TU_RUN_DTYPE)
        cbf = torch.zeros(n_tus, dtype=torch.int32, device="cuda")

        def fused(i):
            d = sets[i % 3]
            B.transform_quantize_reconstruct_runs(d["dst"], stride, d["src"], stride, d["pred"], stride, off, off, off, rq, rdq, plane * frames, n_tus, levels=d["levels"], cbf=cbf)

        def stages(i):
            d = sets[i % 3]
            B.transform_runs(d["res"], stride, off, rq, plane * frames, out=d["coeffs"])
            B.quantize_runs(d["coeffs"], rq, n_tus, out=d["levels"], cbf=cbf)
            B.inverse_transform_add_runs(d["dst"], stride, d["pred"], stride, off, off, d["levels"], rdq, dequant=True)

        px = w * h * frames
        for name, fn, bpp in (("fused", fused, 5), ("three stages", stages, 12)):
            for i in range(3):
                fn(i)
            torch.cuda.synchronize()
            ms = timed(fn, 9)
            report("TU %dx%d%s %s" % (n, n, " DST" if tr else "", name), n_tus, bpp * n * n + 16, ms, size="%dx%d" % (n, n), frames=frames, gpixels_per_s=round(px / ms / 1e6, 1))
    del sets
    torch.cuda.empty_cache()


def bench_transform(coeffs_per_launch=1 << 26, only=None, label=""):
    for tr, lg in wl.TRANSFORMS:
        if only and (tr, lg) not in only:
            continue
        size = 1 << lg
        n = coeffs_per_launch >> (2 * lg)
        tag = ("dst4" if tr else "dct%d" % size) + label
        cs = [wl.inverse_case(tr, lg, n, 70 + i, "natural", frame=(wl.FRAME_W, wl.FRAME_H)) for i in range(2)]
        devs = [dict(pred=B.u8(c["pred"]), pred_off=B.off32(c["pred_off"]), dst=torch.zeros(c["dst_bytes"], dtype=torch.uint8, device="cuda"),
                     dst_off=B.off32(c["dst_off"]), coeffs=B.i16(c["coeffs"])) for c in cs]
        c0 = cs[0]
        ms = timed(lambda i: B.inverse_transform_add(devs[i % 2]["dst"], c0["stride_dst"], devs[i % 2]["pred"], c0["stride_pred"], devs[i % 2]["dst_off"], devs[i % 2]["pred_off"], devs[i % 2]["coeffs"], tr, lg))
        report("inverse_transform_add", n, 4 * size * size + 8, ms, size=tag)
        ms = timed(lambda i: B.dequant_inverse_transform_add(devs[i % 2]["dst"], c0["stride_dst"], devs[i % 2]["pred"], c0["stride_pred"], devs[i % 2]["dst_off"], devs[i % 2]["pred_off"], devs[i % 2]["coeffs"], 51, 14, tr, lg))
        report("dequant_inverse_transform_add", n, 4 * size * size + 8, ms, size=tag)
        fs = [wl.forward_case(tr, lg, n, 80 + i, frame=(wl.FRAME_W, wl.FRAME_H)) for i in range(2)]
        fd = [dict(src=B.i16(f["src"]), off=B.off32(f["src_off"]), out=torch.empty(n * size * size, dtype=torch.int16, device="cuda")) for f in fs]
        ms = timed(lambda i: B.transform(fd[i % 2]["src"], fs[0]["src_stride"], fd[i % 2]["off"], tr, lg, out=fd[i % 2]["out"]))
        report("transform", n, 4 * size * size + 4, ms, size=tag)
        del cs, devs, fs, fd
        torch.cuda.empty_cache()


def bench_quant(n=1 << 27):
    srcs = [wl.fill_i16_device(n, 90 + i, "cuda") for i in range(2)]
    outs = [torch.empty_like(s) for s in srcs]
    ms = timed(lambda i: B.quantize_inverse(srcs[i % 2], 51, 14, out=outs[i % 2]))
    report("quantize_inverse", n // 256, 4 * 256, ms, size="flat (per 16x16)")
    for npb in (16, 64, 256, 1024):
        cbf = torch.empty(n // npb, dtype=torch.int32, device="cuda")
        ms = timed(lambda i: B.quantize(srcs[i % 2], 51, 20, 14, npb, out=outs[i % 2], cbf=cbf))
        report("quantize", n // npb, 4 * npb + 4, ms, size="%d coeffs" % npb)
    for lg in (2, 3, 4, 5):
        size = 1 << lg
        nb = (1 << 26) >> (2 * lg)
        cs = [wl.inverse_case(0, lg, nb, 95 + i, "residual9", frame=(wl.FRAME_W, wl.FRAME_H)) for i in range(2)]
        devs = [dict(pred=B.u8(c["pred"]), pred_off=B.off32(c["pred_off"]), rec=torch.zeros(c["dst_bytes"], dtype=torch.uint8, device="cuda"),
                     rec_off=B.off32(c["dst_off"]), res=B.i16(c["coeffs"])) for c in cs]
        c0 = cs[0]
        ms = timed(lambda i: B.quantize_reconstruct(devs[i % 2]["rec"], c0["stride_dst"], devs[i % 2]["pred"], c0["stride_pred"], devs[i % 2]["rec_off"], devs[i % 2]["pred_off"], devs[i % 2]["res"], size))
        report("quantize_reconstruct", nb, 4 * size * size + 8, ms, size="%dx%d" % (size, size))
        del cs, devs
        torch.cuda.empty_cache()


def bench_me():
    """Config #5: one 3840x2160 luma frame, 64x64 CTUs (+ one row of 64x48), +-64 full search."""
    fw, fh, r = 3840, 2160, 64
    stride_s, stride_r = fw, fw + 2 * r
    src = wl.fill_u8(fw * fh, 5, device="cuda")
    ref = wl.fill_u8(stride_r * (fh + 2 * r), 6, device="cuda")
    total_ms, total_sads, total_blocks = 0.0, 0, 0
    for (w, h, y0, rows) in ((64, 64, 0, 33), (64, 48, 33 * 64, 1)):
        by, bx = np.meshgrid(np.arange(rows), np.arange(fw // w), indexing="ij")
        so = ((y0 + by * h) * stride_s + bx * w).reshape(-1).astype(np.uint32)
        ro = ((y0 + by * h + r) * stride_r + bx * w + r).reshape(-1).astype(np.uint32)
        so_d, ro_d = B.off32(so), B.off32(ro)
        ms = timed(lambda i: B.me_full_search(src, stride_s, ref, stride_r, so_d, ro_d, w, h, r), reps=3)
        total_ms += ms; total_blocks += so.size; total_sads += so.size * (2 * r + 1) ** 2
        report("me_full_search", so.size, w * h + (w + 2 * r) * (h + 2 * r) + 8 + 12, ms, size="%dx%d +-%d" % (w, h, r), gsad_evals_per_s=round(so.size * (2 * r + 1) ** 2 / ms / 1e6, 2))
    print(json.dumps(dict(kernel="me_full_search frame total", ms=round(total_ms, 3), ctus=total_blocks, frames_per_s=round(1e3 / total_ms, 1),
                          gsad_evals_per_s=round(total_sads / total_ms / 1e6, 2))), flush=True)
    # the 64x48 bottom row is an independent call: issue it on a second stream so that it fills the main launch's tail
    parts = []
    for (w, h, y0, rows) in ((64, 64, 0, 33), (64, 48, 33 * 64, 1)):
        by, bx = np.meshgrid(np.arange(rows), np.arange(fw // w), indexing="ij")
        so = ((y0 + by * h) * stride_s + bx * w).reshape(-1).astype(np.uint32)
        ro = ((y0 + by * h + r) * stride_r + bx * w + r).reshape(-1).astype(np.uint32)
        parts.append((w, h, B.off32(so), B.off32(ro), torch.empty((so.size, 3), dtype=torch.int32, device="cuda")))
    side = torch.cuda.Stream()

    def frame(i):
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            w, h, so_d, ro_d, out = parts[1]
            B.me_full_search(src, stride_s, ref, stride_r, so_d, ro_d, w, h, r, out=out)
        w, h, so_d, ro_d, out = parts[0]
        B.me_full_search(src, stride_s, ref, stride_r, so_d, ro_d, w, h, r, out=out)
        main.wait_stream(side)
    ms2 = timed(frame, reps=3)
    print(json.dumps(dict(kernel="me_full_search frame total, bottom row on a second stream", ms=round(ms2, 3), ctus=total_blocks,
                          frames_per_s=round(1e3 / ms2, 1), gsad_evals_per_s=round(total_sads / ms2 / 1e6, 2))), flush=True)


def bench_me_frame(reps=5, quiet=False):
    """Config #5 through the frame-rows entry point: the whole 3840x2160 frame (33 rows of 64x64 CTUs + one of 64x48) in ONE launch."""
    from hevcasm_amd import me_frame
    fw, fh, ctu, r = 3840, 2160, 64, 64
    frames = [me_frame.synthetic_frame_pair(fw, fh, r, seed=900 + i) for i in range(2)]
    devs = [(B.u8(f["cur"]), B.u8(f["ref"])) for f in frames]
    f0 = frames[0]
    rows, cols = -(-fh // ctu), -(-fw // ctu)
    out = torch.empty((rows * cols, 3), dtype=torch.int32, device="cuda")
    run = lambda i: B.me_full_search_frame_rows(devs[i % 2][0], 0, f0["stride_cur"], devs[i % 2][1], r * f0["stride_ref"] + r, f0["stride_ref"], fw, fh, ctu, r, 0, rows, out=out)
    run(0); run(1)
    # launches back to back, as a search over a sequence of frames issues them (an event pair around a single launch adds ~0.1 ms of start-up to a 1.1 ms kernel)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(4 * reps):
        run(i)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (4 * reps)
    heights = [min(ctu, fh - k * ctu) for k in range(rows)]
    side = (2 * r + 1) ** 2
    res = dict(ms=ms, ctus=rows * cols, sad_evals=rows * cols * side, byte_diffs=sum(cols * ctu * h for h in heights) * side,
               bytes=sum(cols * (ctu * h + (ctu + 2 * r) * (h + 2 * r) + 12) for h in heights))
    if SINK is None and not quiet:
        print(json.dumps(dict(kernel="me_full_search_frame_rows, whole 4K frame in one launch", ms=round(ms, 4), ctus=res["ctus"], frames_per_s=round(1e3 / ms, 1),
                              gsad_evals_per_s=round(res["sad_evals"] / ms / 1e6, 2), tera_byte_diffs_per_s=round(res["byte_diffs"] / ms / 1e9, 2))), flush=True)
    return res


def bench_satd():
    """Hadamard SATD 2x2 / 4x4 / 8x8 over 2^26 pixels per launch, displaced reference."""
    for n in (2, 4, 8):
        nb = (1 << 26) // (n * n)
        c = wl.sad_case(n, n, nb, 300 + n, frame=(wl.FRAME_W, wl.FRAME_H), device="cuda") if n >= 4 else None
        if c is None:   # 2x2 tiles: the SAD workload generator needs multiples of 4; tile by hand
            sp = wl.Plane(wl.FRAME_W, wl.FRAME_H, wl.Plane.frames_for(nb, n, n, wl.FRAME_W, wl.FRAME_H))
            c = dict(src=wl.fill_u8(sp.size, 1, device="cuda"), ref=wl.fill_u8(sp.size, 2, device="cuda"), stride_src=sp.stride, stride_ref=sp.stride,
                     src_off=sp.tile_offsets(nb, n, n).astype(np.uint32), ref_off=sp.tile_offsets(nb, n, n).astype(np.uint32))
        so, ro = B.off32(c["src_off"]), B.off32(c["ref_off"])
        out = torch.empty(nb, dtype=torch.int32, device="cuda")
        ms = timed(lambda i: B.hadamard_satd(c["src"], c["stride_src"], c["ref"], c["stride_ref"], so, ro, n.bit_length() - 1, out=out))
        report("hadamard_satd", nb, 2 * n * n + 4 + 8, ms, size="%dx%d" % (n, n))
        del c
        torch.cuda.empty_cache()


def bench_intra(pixels=1 << 28):
    """INTRA_DC over 2^28 output pixels per launch (2^26 until round 5: a 64 MB launch is 8 us at the HBM peak, i.e. mostly ramp), 128-byte
    neighbour records and packed ones."""
    for log2 in (2, 3, 4, 5):
        for edge in ((0, 1) if log2 < 5 else (0,)):
            for stride in (128, 2 << log2):     # the reference's 128-byte record, and records packed back to back
                size = 1 << log2
                nb = pixels >> (2 * log2)
                neigh = wl.fill_u8(stride * nb + 128, 400 + log2, device="cuda")
                dst = torch.empty(nb * size * size, dtype=torch.uint8, device="cuda")
                packed = (8 << 8) | (log2 << 1) | edge
                ms = timed(lambda i: B.pred_intra(dst, neigh, stride, 1, packed, nb))
                report("pred_intra_dc", nb, size * size + 2 * size, ms, size="%dx%d%s" % (size, size, " edge" if edge else ""), neighbour_stride=stride)
                del neigh, dst
                torch.cuda.empty_cache()


def bench_intra_modes(pixels=1 << 28):
    """SURVEY 8f-3 beyond DC (unpinned): 2^28 output pixels per launch (a 2^26-pixel launch is 64 MB = 8 us at the HBM peak: launch ramp, not kernel),
    144-byte records; one mode for the batch (planar, pure vertical, a positive and a negative angle from each family) and one random mode per block."""
    for log2 in (2, 3, 4, 5):
        size = 1 << log2
        nb = pixels >> (2 * log2)
        neigh = wl.fill_u8(144 * nb + 128, 450 + log2, device="cuda")
        dst = torch.empty(nb * size * size, dtype=torch.uint8, device="cuda")
        packed = (8 << 8) | (log2 << 1) | (1 if log2 < 5 else 0)
        for mode in (0, 26, 30, 21, 6, 14):
            ms = timed(lambda i: B.pred_intra(dst, neigh, 144, mode, packed, nb))
            report("pred_intra_mode%d" % mode, nb, size * size + 4 * size + 1, ms, size="%dx%d" % (size, size))
        modes = B.u8(np.random.default_rng(log2).integers(0, 35, nb).astype(np.uint8))
        ms = timed(lambda i: B.pred_intra_modes(dst, neigh, 144, modes, packed))
        report("pred_intra_modes (random per block)", nb, size * size + 4 * size + 2, ms, size="%dx%d" % (size, size))
        # compact records (HEVCASM_HIP_INTRA_COMPACT: 4N + 1 contiguous samples per block instead of the reference's 129-byte layout)
        cs = 4 * size + 1
        comp = wl.fill_u8(cs * nb + 64, 460 + log2, device="cuda")
        ms = timed(lambda i: B.pred_intra_modes(dst, comp, cs, modes, packed | B.INTRA_COMPACT))
        report("pred_intra_modes compact records", nb, size * size + 4 * size + 2, ms, size="%dx%d" % (size, size))
        del comp
        filt = torch.zeros(144 * nb, dtype=torch.uint8, device="cuda")
        ms = timed(lambda i: B.intra_filter_neighbours(neigh, 144, log2, 1, nb, out=filt, out_stride=144))
        report("intra_filter_neighbours", nb, 2 * (4 * size + 1), ms, size="%dx%d" % (size, size))
        del neigh, dst, modes, filt
        torch.cuda.empty_cache()


def bench_intra_frames(reps=3, frames=(1, 64)):
    """SURVEY 8f row 3, second half: whole 1080p intra frames reconstructed on the device (hevcasm_hip_intra_reconstruct_frames) -- a random quad-tree of TUs
    4x4 ... 32x32, every mode, QP 27, strong smoothing on: one frame alone (the dependency-limited latency: cols + 2 (rows - 1) = 62 wavefront steps) and 64 frames
    in one call (throughput: frames are independent and interleave inside every step).  Blocks = TUs; bytes per TU = (source + reconstruction + 2-byte levels
    per pixel) + 12 of descriptor and cbf."""
    w, h, ctu = 1920, 1080, 64
    maps = [wl.intra_tu_map(w, h, ctu, 100 + f, modes="mixed") for f in range(4)]
    srcs = [B.u8(wl.frame_2d(w, h, w, 7 + f)) for f in range(4)]
    for F in frames:
        tus = [maps[f % 4][0] for f in range(F)]
        job = B.IntraFrames(w, h, ctu, tus, [maps[f % 4][1] for f in range(F)], wl.intra_qp(27), c_idx=0, strong=1, stride=w)
        src = torch.cat([srcs[f % 4] for f in range(F)])
        job.run(src)
        torch.cuda.synchronize()
        job.status()
        ms = timed(lambda i: job.run(src), reps)
        n_tus = sum(len(t) for t in tus)
        report("intra_frames", n_tus, (4 * w * h * F + 12 * n_tus) // n_tus, ms, size="1080p x %d" % F, frames=F, ms_per_frame=round(ms / F, 4), frames_per_s=round(F / ms * 1e3, 1),
               note="mixed quad-tree 4x4..32x32, all 35 modes, QP 27; wavefront of 62 CTU steps per frame; %s" % ("one frame alone: dependency-limited" if F == 1 else "%d independent frames in one call" % F))
        del job, src
        torch.cuda.empty_cache()


def bench_ssd_linear():
    """linear SSD, 2 x 256 MiB per launch cut into runs of `size` bytes (whole-frame PSNR = sum of rows)."""
    total = 1 << 28
    a, b = wl.fill_u8(total, 500, device="cuda"), wl.fill_u8(total, 501, device="cuda")
    for size in (64, 0x200, 1920, 4096, 65536):
        n = total // size
        off = B.off32((np.arange(n, dtype=np.int64) * size).astype(np.uint32))
        out = torch.empty(n, dtype=torch.int32, device="cuda")
        ms = timed(lambda i: B.ssd_linear(a, b, off, off, size, out=out))
        report("ssd_linear", n, 2 * size + 12, ms, size="%d bytes" % size)
    del a, b
    torch.cuda.empty_cache()
    # whole-frame SSE, 64-bit sums (hevcasm_hip_frame_sse_batch): frames inside planes with a 64-byte border, cold (two sets of planes alternate)
    for (fw, fh, frames) in ((3840, 2160, 32), (1920, 1080, 128), (1920, 1080, 1), (3840, 2160, 1)):
        stride = fw + 128
        span = stride * fh
        planes = [(wl.fill_u8(span * frames + 64, 510 + k, device="cuda"), wl.fill_u8(span * frames + 64, 520 + k, device="cuda")) for k in range(2)]
        off = B.off32((np.arange(frames, dtype=np.int64) * span + 64).astype(np.uint32))
        out = torch.empty(frames, dtype=torch.int64, device="cuda")
        ms = timed(lambda i: B.frame_sse(planes[i % 2][0], stride, planes[i % 2][1], stride, off, off, fw, fh, out=out))
        report("frame_sse", frames, 2 * fw * fh + 16, ms, size="%dx%d x%d" % (fw, fh, frames))
        del planes
        torch.cuda.empty_cache()


def bench_subpel():
    """quarter-pel refinement of every block of a 3840x2160 luma frame, fused kernel vs
    the unfused composition (49 x pred_uni_8to8_batch into a scratch frame + sad_batch)."""
    fw, fh, m = 3840, 2160, 16
    stride_r = fw + 2 * m
    src = wl.fill_u8(fw * fh, 7, device="cuda")
    ref = wl.fill_u8(stride_r * (fh + 2 * m), 8, device="cuda")
    scratch = torch.zeros(fw * fh, dtype=torch.uint8, device="cuda")
    for (w, h) in ((8, 8), (16, 16), (32, 32), (64, 64), (64, 32), (16, 8)):
        by, bx = np.meshgrid(np.arange(fh // h), np.arange(fw // w), indexing="ij")
        so = (by * h * fw + bx * w).reshape(-1).astype(np.uint32)
        ro = ((by * h + m) * stride_r + bx * w + m).reshape(-1).astype(np.int64)
        n = so.size
        so_d, ro_d = B.off32(so), B.off32(ro.astype(np.uint32))
        out = torch.empty((n, 3), dtype=torch.int32, device="cuda")
        ms = timed(lambda i: B.subpel_refine(src, fw, ref, stride_r, so_d, ro_d, w, h, out=out), reps=5)
        offs = [B.off32((ro + (qy >> 2) * stride_r + (qx >> 2)).astype(np.uint32)) for qy in range(-3, 4) for qx in range(-3, 4)]
        fracs = [torch.full((n,), (qx & 3) | ((qy & 3) << 4), dtype=torch.uint8, device="cuda") for qy in range(-3, 4) for qx in range(-3, 4)]
        cost = torch.empty(n, dtype=torch.int32, device="cuda")

        def unfused(i):
            for k in range(49):
                B.pred_uni(scratch, fw, ref, stride_r, so_d, offs[k], fracs[k], 8, w, h)
                B.sad(src, fw, scratch, fw, so_d, so_d, w, h, out=cost)
        ms_unfused = timed(unfused, reps=2)
        report("subpel_refine", n, (w + 8) * (h + 8) + w * h + 8 + 12, ms, size="%dx%d" % (w, h),
               candidates_per_s_g=round(49 * n / ms / 1e6, 2), unfused_ms=round(ms_unfused, 3), fused_speedup=round(ms_unfused / ms, 2),
               frames_per_s=round(1e3 / ms, 1))
        ms_satd = timed(lambda i: B.subpel_refine(src, fw, ref, stride_r, so_d, ro_d, w, h, metric=1, out=out), reps=5)
        report("subpel_refine_satd", n, (w + 8) * (h + 8) + w * h + 8 + 12, ms_satd, size="%dx%d" % (w, h),
               candidates_per_s_g=round(49 * n / ms_satd / 1e6, 2), frames_per_s=round(1e3 / ms_satd, 1))


if __name__ == "__main__":
    what = sys.argv[1:] or ["quant", "transform", "pred", "satd", "intra", "ssdlin", "me", "subpel"]
    torch.cuda.set_device(0)
    for w in what:
        if w == "pred16":                                 # the four headline rows: luma 16x16 + chroma 8x8 (Cb + Cr), uni and bi, 64 frames (bench.py runs this on the pattern-only library)
            bench_pred(frames=64, only=((16, 16),))
            continue
        if w == "pred4tap":                               # chroma only, every PU size
            bench_pred(taps_only=4)
            continue
        if w.startswith("pred8ctu") and w[8:].isdigit():  # the same, luma only
            bench_pred(taps_only=8, only=((16, 16), (32, 32), (64, 64), (64, 32), (16, 64)), mv_region=int(w[8:]))
            continue
        if w.startswith("predctu") and w[7:].isdigit():   # one vector per N x N luma region (N / 2 chroma) instead of one per PU: what coherent motion is worth
            bench_pred(only=((8, 8), (16, 16), (32, 32), (16, 32), (32, 16)), mv_region=int(w[7:]))
            continue
        if w.startswith("pred8wide") and w[9:].isdigit():  # luma, the 64- and 48-wide PU sizes, N frames per launch
            bench_pred(frames=int(w[9:]), taps_only=8, only=((64, 16), (64, 32), (64, 48), (64, 64), (48, 64)))
            continue
        if w == "pred8odd":                               # luma, the 24- and 48-wide PU sizes (+ 32x32 as a yardstick)
            bench_pred(taps_only=8, only=((24, 32), (48, 64), (32, 32)))
            continue
        if w == "pred8w8":                                # luma, the 8-wide PU sizes
            bench_pred(taps_only=8, only=((8, 8), (8, 16), (8, 32), (8, 4)))
            continue
        if w == "pred8tap":                               # luma only, every PU size
            bench_pred(taps_only=8)
            continue
        if w.startswith("pred8sub") and w[8:].isdigit():  # luma, square PU sizes, N frames per launch
            bench_pred(frames=int(w[8:]), taps_only=8, only=((16, 16), (32, 32), (64, 64), (64, 32), (16, 64)))
            continue
        if w.startswith("pred4sub") and w[8:].isdigit():  # chroma, a subset of PU sizes, N frames per launch (ramp / tail against steady state)
            bench_pred(frames=int(w[8:]), taps_only=4, only=((16, 16), (32, 32), (32, 64), (16, 64), (64, 64), (8, 16), (4, 16)))
            continue
        if w == "pred16":
            bench_pred(only=((16, 16),))
            continue
        if w == "predq":                                  # a quick subset of PU sizes for same-box A/B runs (tools/ab.py)
            bench_pred(only=((8, 8), (16, 16), (32, 32), (64, 64), (16, 64), (64, 16), (8, 32)))
            continue
        if w.startswith("pred") and w[4:].isdigit():      # pred256: the prediction rows at 256 frames per launch
            bench_pred(frames=int(w[4:]))
            continue
        dict(pred=bench_pred, predlat=bench_pred_latency, costlat=bench_cost_latency, tulat=bench_tu_latency, tufusedsizes=bench_tu_fused_sizes, intraframes=bench_intra_frames, meframe=bench_me_frame, transform=bench_transform, quant=bench_quant, me=bench_me, subpel=bench_subpel, satd=bench_satd, intra=bench_intra, intramodes=bench_intra_modes, ssdlin=bench_ssd_linear)[w]()
