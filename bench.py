#!/usr/bin/env python3
"""bench.py -- BASELINE.json config #2 on MI355X: batched SAD + 4-way multi-reference SAD over
all 20 block sizes >= 8x8, 1 Mi blocks per size per GPU, inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One *step* = one pass of the hot path over the whole batch: for each of the 20 sizes, one
hevcasm_hip_sad_batch launch and one hevcasm_hip_sad_multiref_batch launch over that size's
blocks (sizes whose frame stack would exceed the 4 GiB offset window run as two launches).
Every size has its own frame stacks (about 46 GB in total), so a launch never finds its data in
the 256 MiB Infinity Cache left by the previous one: the numbers are cold-HBM numbers.

value      = block evaluations per second of the whole job, in Mblocks/s (a 4-way call counts as
             one block), all ranks, steps timed between barrier + synchronize brackets, MAX over ranks.
roofline   = the kernel with the largest share of the timed region: algorithmic bytes per launch /
             its mean duration, measured live with HIP events on the launch stream, against 8 TB/s.
cpu_baseline = the REFERENCE's own C path (oracle/_ref/refdrv, gcc -O3) timed on this host's
             cores on a bounded sample of the same mix (rank 0, N = 1 only).  Baseline, not target.
Multi-GPU: weak scaling, each rank owns its own 1 Mi-block batch per size; no data-path collective.

roofline.per_family = the other kernels BASELINE.json's metric names, measured in the same run after the
             timed region (N = 1): SAD 16x16, uni / bi inter prediction (config #3: luma 16x16 8-tap, chroma 8x8
             4-tap; 64 frames per launch, ONE frame per launch, and one frame's luma + Cb + Cr replayed as a HIP
             graph), inverse / fused-dequantised / forward transforms 8x8..32x32 (config #4), and the full-search
             frame of config #5 (VALU-bound: SAD evaluations/s against the v_qsad_pk_u16_u8 issue rate).

    python bench.py --workload me5 [--gpus N ...]
BASELINE.json configs[4] instead: full-search motion estimation, +-64, over 3840x2160 frames of 64x64 CTUs (bottom row
64x48), every frame's CTUs sharded across the ranks -- contiguous CTU rows, handed out balanced to one CTU (--me-split rows: to one
row) -- each rank holding the rows its CTUs lie in + the +-64-row halo of the reference frame (hevcasm_amd/me_frame.py, shard.py).  A step = one frame; strong scaling; no data-path collective
(after the timed region the 12-byte answers are collected and rank 0 checks a sample of them against the
reference's hevcasm_sad composed per candidate).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
OFFSET_WINDOW = (1 << 32) - (1 << 24)


MAX_DISP = 7   # reference displacement of the workload: uniform in [-7, 7]^2 per candidate


def algorithmic_bytes(w, h, ways):
    """Bytes one block must move (SURVEY.md 8(d) plus the u32 descriptors this API reads: one
    source offset and `ways` reference offsets).
      1-way: 2WH + 4 + 8.
      4-way: SURVEY prices it 5WH + 16, which counts the four candidates as if they were disjoint.
             In this workload they all lie within +-7 pixels of one position, so they overlap, and
             because the blocks tile the frame the (W + 14) x (H + 14) windows of neighbouring
             blocks overlap too: over a launch every byte of the reference plane is needed exactly
             once.  Pricing candidates (or windows) separately lets "achieved" exceed the HBM peak
             (8.2 TB/s at 32x8), so the 4-way kernel is priced at its COMPULSORY bytes: the source
             block plus the block's share of the reference plane (frame + borders, 1.031 WH), never
             more than the window or four disjoint candidates.  The measured HBM traffic
             (profiles/traffic.json) is within 1-3 % of this."""
    if ways == 1:
        return 2 * w * h + 4 + 8
    from hevcasm_amd import workload as wl
    plane_share = w * h * ((wl.FRAME_W + 2 * 16) * (wl.FRAME_H + 2 * 8)) / float(wl.FRAME_W * wl.FRAME_H)
    ref = min(ways * w * h, (w + 2 * MAX_DISP) * (h + 2 * MAX_DISP), int(round(plane_share)))
    return w * h + ref + 4 * ways + 4 * (1 + ways)


def max_over_ranks(seconds, device=None):
    """Barrier, then the maximum of a per-rank elapsed time (identity when not distributed)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return seconds
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class SizeBatch:
    """Device-resident inputs of one block size: one or more chunks, each within the offset window."""

    def __init__(self, w, h, n_blocks, seed, device):
        import torch
        from hevcasm_amd import batch as B, workload as wl
        self.w, self.h, self.n = w, h, n_blocks
        self.chunks = []
        per_frame = (wl.FRAME_W // w) * (wl.FRAME_H // h)
        ref_frame_bytes = (wl.FRAME_W + 32) * (wl.FRAME_H + 16)
        max_frames = OFFSET_WINDOW // ref_frame_bytes
        max_blocks = max_frames * per_frame
        n_chunks = -(-n_blocks // max_blocks)
        per_chunk = -(-n_blocks // n_chunks)          # equal launches, each inside the 4 GiB offset window
        done = 0
        while done < n_blocks:
            m = min(n_blocks - done, per_chunk)
            c = wl.sad_case(w, h, m, seed + done, ways=4, frame=(wl.FRAME_W, wl.FRAME_H), device=device)
            ro4 = B.off32(c["ref_off"])
            self.chunks.append(dict(
                n=m, src=c["src"], ref=c["ref"], stride_src=c["stride_src"], stride_ref=c["stride_ref"],
                src_off=B.off32(c["src_off"]), ref_off4=ro4, ref_off1=ro4.view(m, 4)[:, 0].contiguous(),
                out1=torch.empty(m, dtype=torch.int32, device=device), out4=torch.empty((m, 4), dtype=torch.int32, device=device)))
            done += m

    def launch(self, ways, chunk):
        from hevcasm_amd import batch as B
        c = chunk
        if ways == 1:
            B.sad(c["src"], c["stride_src"], c["ref"], c["stride_ref"], c["src_off"], c["ref_off1"], self.w, self.h, out=c["out1"])
        else:
            B.sad_multiref(c["src"], c["stride_src"], c["ref"], c["stride_ref"], c["src_off"], c["ref_off4"], self.w, self.h, out=c["out4"])


def host_cores(max_procs):
    """(processes to run, description).  A GPU box gives one GPU's job a share of the host (16 cores per GPU on the
    pool this was written on) whatever os.cpu_count() says, so the baselines run on min(cores usable, max_procs)."""
    total = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = total
    try:    # cgroup v2 CPU quota, if any
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            usable = max(1, min(usable, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    n = max(1, min(max_procs, usable))
    note = "%d procs; os.cpu_count()=%d, usable=%d" % (n, total, usable)
    if n < usable:
        note += ", capped at %d (--cpu-procs)" % max_procs
    return n, dict(host_cpus=total, usable_cpus=usable, note=note)


def cpu_baseline(sizes, sample_blocks, budget_s, max_procs=16):
    """The reference's C path on the host: one refdrv process per core, each timing every
    (size, ways) of the mix on `sample_blocks` blocks for an equal share of `budget_s`."""
    from concurrent.futures import ThreadPoolExecutor
    from hevcasm_amd import workload as wl
    from oracle import pyref
    if not pyref.available():
        return None
    procs_n, cores_info = host_cores(max_procs)
    combos = [(w, h, ways) for (w, h) in sizes for ways in (1, 4)]
    ms_each = max(20, int(1000.0 * budget_s / len(combos)))
    cases = {(w, h): wl.sad_case(w, h, sample_blocks, 9000 + w * 64 + h, ways=4, frame=(wl.FRAME_W, wl.FRAME_H)) for (w, h) in sizes}

    def worker(ms):
        rates = {}
        with pyref.RefProc() as p:
            for (w, h, ways) in combos:
                c = cases[(w, h)]
                ro = c["ref_off"] if ways == 4 else np.ascontiguousarray(c["ref_off"].reshape(-1, 4)[:, 0])
                secs, blocks = p.bench_sad(c["src"], c["stride_src"], c["ref"], c["stride_ref"], c["src_off"], ro, w, h, ways, ms)
                rates[(w, h, ways)] = blocks / secs
        return rates

    mix = lambda r: len(combos) / sum(1.0 / r[c] for c in combos)     # throughput of the mix (equal block counts per (size, ways)): harmonic combination
    # "each on 1 core and on all host cores".  The 1-core column is a RUN of its own (round 6; it used to be value / cores): ONE refdrv
    # process with the host otherwise idle, same mix, a quarter of the per-combination time (a single core has the whole memory system to itself,
    # so it is not the all-core rate divided by the cores)
    ms_one = max(20, ms_each // 4)
    one = worker(ms_one)
    with ThreadPoolExecutor(procs_n) as ex:
        per_proc = list(ex.map(worker, [ms_each] * procs_n))
    total = sum(mix(r) for r in per_proc)
    sad16 = sum(r[(16, 16, 1)] for r in per_proc)
    return dict(value=round(total / 1e6, 3), unit="Mblocks/s", cores=procs_n, kind="reference", value_per_core=round(total / procs_n / 1e6, 3),
                value_1core=round(mix(one) / 1e6, 3), sad16_1core_mblocks=round(one[(16, 16, 1)] / 1e6, 3),
                sample_1core="ONE refdrv process alone on the host: %d (size, ways) x %d blocks x %d ms, same mix" % (len(combos), sample_blocks, ms_one),
                sample="reference C (refdrv, gcc -O3): %d procs x %d (size, ways) x %d blocks x %d ms, same mix" % (procs_n, len(combos), sample_blocks, ms_each),
                sad16x16_single_ref_mblocks_per_s=round(sad16 / 1e6, 3), **cores_info)


def host_simd_baseline(sizes, sample_blocks, budget_s, max_procs=16):
    """This library's own host table entries with the HIP bit masked off -- what a codec that keeps calling per block gets:
    the SSE2 / AVX2 intrinsics kernels of csrc/host/sad_x86.c where the reference has x86 assembly, the C path elsewhere -- timed by
    hevcasm_amd/lib/hevcasm_host_bench, one process per core, on the same size mix.  (The reference's assembly itself cannot be
    assembled in this image: no yasm / nasm.)"""
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    from hevcasm_amd import capi
    exe = os.path.join(os.path.dirname(capi.LIB_PATH), "hevcasm_host_bench")
    if not os.path.exists(exe):
        return None
    procs_n, cores_info = host_cores(max_procs)
    mask = capi.lib().hevcasm_instruction_set_support() & ~capi.HIP
    ms_each = max(20, int(1000.0 * budget_s / (2 * len(sizes))))
    want = set((w, h) for (w, h) in sizes)

    def worker(i, ms=None):
        out = subprocess.run([exe, "%x" % mask, str(ms or ms_each), str(sample_blocks), str(1234 + i)], capture_output=True, text=True,
                             env=dict(os.environ, HEVCASM_DISABLE_HIP="1"), timeout=600)
        if out.returncode != 0:
            raise RuntimeError("hevcasm_host_bench failed: " + out.stderr[-500:])
        rates = {}
        for line in out.stdout.splitlines():
            w, h, ways, blocks, secs = line.split()
            if (int(w), int(h)) in want:
                rates[(int(w), int(h), int(ways))] = int(blocks) / float(secs)
        return rates

    one = worker(99, max(20, ms_each // 4))      # the 1-core column: one process alone on the host, a run of its own
    with ThreadPoolExecutor(procs_n) as ex:
        per_proc = list(ex.map(worker, range(procs_n)))
    combos = sorted(per_proc[0])
    total = sum(len(combos) / sum(1.0 / r[c] for c in combos) for r in per_proc)
    names = [n for n, bit in (("SSE2", capi.SSE2), ("AVX2", capi.AVX2)) if mask & bit]
    return dict(value=round(total / 1e6, 3), unit="Mblocks/s", cores=procs_n, kind="port", value_per_core=round(total / procs_n / 1e6, 3),
                value_1core=round(len(combos) / sum(1.0 / one[c] for c in combos) / 1e6, 3),
                sample="host tables, mask %#x (%s intrinsics): %d procs x %d combos x %d blocks x %d ms, same mix"
                       % (mask, "/".join(names) or "no SIMD", procs_n, len(combos), sample_blocks, ms_each),
                sad16x16_single_ref_mblocks_per_s=round(sum(r.get((16, 16, 1), 0.0) for r in per_proc) / 1e6, 3), **cores_info)


# ---------------------------------------------------------------------------------------------------------------------------
# The driver's record keeps the SCALAR members of config / roofline / cpu_baseline and cuts strings at 120 characters (VERDICT
# round 3): everything the metric names is therefore flattened into scalar keys; the row lists go to the top-level "detail".

FAMILY_KEYS = {   # (family, kernel + size as bench_families reports them) -> key prefix
    ("sad", "sad 16x16 single-ref"): "sad16",
    ("inter_pred", "pred_uni_8tap 16x16"): "pred_uni_luma16", ("inter_pred", "pred_bi_8tap 16x16"): "pred_bi_luma16",
    ("inter_pred", "pred_uni_4tap 8x8"): "pred_uni_chroma8_1plane", ("inter_pred", "pred_bi_4tap 8x8"): "pred_bi_chroma8_1plane",
    ("inter_pred", "pred_uni_4tap Cb+Cr 8x8"): "pred_uni_chroma8", ("inter_pred", "pred_bi_4tap Cb+Cr 8x8"): "pred_bi_chroma8",
    ("inter_pred_256f", "pred_uni_4tap Cb+Cr 8x8"): "pred_uni_chroma8_256f", ("inter_pred_256f", "pred_bi_4tap Cb+Cr 8x8"): "pred_bi_chroma8_256f",
    ("inter_pred_256f", "pred_uni_8tap 16x16"): "pred_uni_luma16_256f", ("inter_pred_256f", "pred_bi_8tap 16x16"): "pred_bi_luma16_256f",
    ("inter_pred_latency", "pred_uni_8tap 16x16"): "pred_uni_luma16_1frame", ("inter_pred_latency", "pred_bi_8tap 16x16"): "pred_bi_luma16_1frame",
    ("inter_pred_latency", "pred_uni_4tap 8x8"): "pred_uni_chroma8_1frame", ("inter_pred_latency", "pred_bi_4tap 8x8"): "pred_bi_chroma8_1frame",
    ("inter_pred_latency", "pred_uni frame graph 16x16 luma + 2 x 8x8 chroma"): "pred_uni_frame_graph",
    ("inter_pred_latency", "pred_bi frame graph 16x16 luma + 2 x 8x8 chroma"): "pred_bi_frame_graph",
    ("inter_pred_latency", "pred_uni frame partition, one launch 24 sizes x 3 planes"): "pred_uni_frame_runs",
    ("inter_pred_latency", "pred_bi frame partition, one launch 22 sizes x 3 planes"): "pred_bi_frame_runs",
    ("inter_pred_latency", "pred_uni frame partition, a launch per (size, plane) 24 sizes x 3 planes"): "pred_uni_frame_72_launches",
    ("inter_pred_latency", "pred_bi frame partition, a launch per (size, plane) 22 sizes x 3 planes"): "pred_bi_frame_66_launches",
    ("block_cost_latency", "sad frame partition, one launch 24 sizes"): "sad_frame_runs", ("block_cost_latency", "sad_multiref_4 frame partition, one launch 24 sizes"): "sad4_frame_runs",
    ("block_cost_latency", "sad frame partition, a launch per size 24 sizes"): "sad_frame_24_launches",
    ("block_cost_latency", "sad_multiref_4 frame partition, a launch per size 24 sizes"): "sad4_frame_24_launches",
    ("intra_frames", "intra_frames 1080p x 1"): "intra_frame", ("intra_frames", "intra_frames 1080p x 64"): "intra_frames64",
    # round 6: the transform stages of a frame's TUs (4x4 DST ... 32x32, each size with its quantiser parameters): the residual loop as ONE launch, as three, as fifteen
    ("tu_latency", "transform_quantize_reconstruct frame TUs, ONE launch (fused: residual, coefficients, levels stay in the workgroup) 5 sizes"): "tu_fused_frame",
    ("tu_latency", "residual (torch) + three stages, one launch each 5 sizes"): "tu_residual_3stages_frame",
    ("tu_latency", "transform+quantize+dequant_inverse three stages, one launch each 5 sizes"): "tu_3stages_frame",
    ("tu_latency", "transform+quantize+dequant_inverse three stages, a launch per size and stage 5 sizes"): "tu_15_launches_frame",
    ("tu_latency", "32 frames' TUs (z-scan inside a run): transform_quantize_reconstruct, ONE launch (fused) 5 sizes"): "tu_fused_32frames",
    ("tu_latency", "32 frames' TUs (z-scan inside a run): three stages, one launch each (residual plane given) 5 sizes"): "tu_3stages_32frames",
    ("tu_latency", "32 frames' TUs (z-scan inside a run): residual (torch) + three stages 5 sizes"): "tu_residual_3stages_32frames",
}
for _n in (8, 16, 32):
    for _fam, _sfx in (("", ""), ("_large_launch", "_2p28")):
        FAMILY_KEYS[("idct" + _fam, "inverse_transform_add dct%d" % _n)] = "idct%d%s" % (_n, _sfx)
        FAMILY_KEYS[("idct" + _fam, "dequant_inverse_transform_add dct%d" % _n)] = "fused_idct%d%s" % (_n, _sfx)
        FAMILY_KEYS[("fdct" + _fam, "transform dct%d" % _n)] = "fdct%d%s" % (_n, _sfx)


# Integer VALU roof of the part: 256 CUs x 4 SIMDs x 16 lanes per clock at 2.4 GHz (tools/ubench/valu_rates.hip measures one full-rate wave-instruction
# per ~4 cycles per SIMD: profiles/r01_valu_issue_rates.txt), and the vector lane-operations the luma prediction kernels issue per output pixel of a
# 16x16 PU (SQ_INSTS_VALU x 64 / pixels, rocprofv3 --pmc: uni = pred_stream_kernel, profiles/r04_pred_transform_counters.txt; bi = pred_hfirst_bi8_kernel,
# profiles/r04_pred_hfirst_bi_ab.txt -- the stream kernel it replaced at this size issued 43.7).  <key>_valu_frac says how much of that roof a row uses, next
# to the HBM fraction the metric asks for; what bounds these kernels is their access pattern (profiles/r04_pred_access_mix.txt).
VALU_PEAK_LANE_OPS = 256 * 4 * 16 * 2.4e9
PRED_LANE_OPS_PER_PIXEL = {"pred_uni_luma16": 20.1, "pred_bi_luma16": 34.1, "pred_uni_luma16_256f": 20.1, "pred_bi_luma16_256f": 34.1}


def flatten_families(rows):
    """roofline.per_family rows -> flat scalar keys: <prefix>_frac (of the 8 TB/s HBM peak), _gbs (algorithmic), _ms (per launch), _mblocks (per s)."""
    out = {}
    for r in rows or []:
        if r.get("family") == "full_search":
            out.update(me_frame_ms=r["ms"], me_frames_per_s=r["frames_per_s"], me_gsad_per_s=r["gsad_evals_per_s"], me_valu_frac=r["frac_of_valu"],
                       me_valu_frac_nominal=r.get("frac_of_valu_nominal"), me_valu_peak_tdiffs=r["peak"], me_hbm_frac=r["frac"])
            continue
        key = FAMILY_KEYS.get((r.get("family"), r.get("kernel")))
        if key is None:
            continue
        out[key + "_frac"] = r["frac"]
        out[key + "_gbs"] = r["gbs"]
        out[key + "_ms"] = r["ms"]
        out[key + "_mblocks"] = r["mblocks_per_s"]
        # the lane-operation counts are counter readings of the kernels the DEFAULT launcher picks: with any prediction switch (or another build of the
        # library) in the environment a different kernel may run, and the figure would be silently wrong -- it is then left out
        if key in PRED_LANE_OPS_PER_PIXEL and not any(k.startswith("HEVCASM_PRED_") or k in ("HEVCASM_HIP_LIB", "HEVCASM_HFIRST_AHEAD") for k in os.environ):
            out[key + "_valu_frac"] = round(r["mblocks_per_s"] * 1e6 * 256 * PRED_LANE_OPS_PER_PIXEL[key] / VALU_PEAK_LANE_OPS, 4)
    return out


def scalars_only(d):
    """What the driver's record keeps of a config / roofline / cpu_baseline object: scalar members, strings cut at 120 characters."""
    out = {}
    for k, v in (d or {}).items():
        if isinstance(v, (dict, list, tuple)):
            continue
        out[k] = v[:120] if isinstance(v, str) else v
    return out


CPU_FAMILY_CASES = (   # key, what
    ("pred_uni_luma16", ("pred", 8, 16, 16, False)), ("pred_bi_luma16", ("pred", 8, 16, 16, True)),
    ("pred_uni_chroma8", ("pred", 4, 8, 8, False)), ("pred_bi_chroma8", ("pred", 4, 8, 8, True)),
    ("idct8", ("inv", 3, False)), ("idct16", ("inv", 4, False)), ("idct32", ("inv", 5, False)), ("fused_idct32", ("inv", 5, True)),
)


def cpu_family_legs(budget_s, max_procs=16, sample_pus=8100):
    """CPU baselines BESIDE inter prediction and the inverse transforms (north_star: "alongside ... timed on the node's host cores ... in
    the same run"; SURVEY 8(d) "CPU beside it"), one process per usable core, same generators as the GPU rows:
      ref_<key>_mblocks  the REFERENCE's own C table entries (pred_inter.c:154-228, :490-530; residual_decode.c:371-413, quantize.c:53-62 in
                         front when fused) through oracle/_ref/refdrv;
      simd_<key>_mblocks this library's host table entries under the host's x86 mask (intrinsics kernels where the reference has assembly),
                         per-block calls, hevcasm_amd/lib/hevcasm_host_bench.
    Samples: one 1080p frame's worth of PUs (8100 16x16 luma PUs; as many 8x8 chroma PUs), 2^19 coefficients per transform size."""
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    from hevcasm_amd import capi, workload as wl
    from oracle import pyref
    procs_n, _ = host_cores(max_procs)
    out = {"family_cores": procs_n, "family_sample_pus": sample_pus, "family_sample_coeffs": 1 << 19}
    ms_each = max(50, int(1000.0 * budget_s / 2 / len(CPU_FAMILY_CASES)))
    if pyref.available():
        cases = {}
        for key, what in CPU_FAMILY_CASES:
            if what[0] == "pred":
                _, taps, w, h, bi = what
                cases[key] = wl.pred_case(taps, w, h, sample_pus, 4000 + taps + bi, bi=bi, frame=(1920, 1080) if taps == 8 else (960, 540))
            else:
                cases[key] = wl.inverse_case(0, what[1], (1 << 19) >> (2 * what[1]), 4100 + what[1], "natural", frame=(wl.FRAME_W, wl.FRAME_H))

        def ref_worker(_):
            rates = {}
            with pyref.RefProc() as p:
                for key, what in CPU_FAMILY_CASES:
                    c = cases[key]
                    if what[0] == "pred":
                        refs = 2 if what[4] else 1
                        secs, n = p.bench_pred(what[1], c["dst_bytes"], c["stride_dst"], [c["ref%d" % r] for r in range(refs)], c["stride_ref"], c["dst_off"],
                                               [c["ref%d_off" % r] for r in range(refs)], [c["frac%d" % r] for r in range(refs)], what[2], what[3], ms_each)
                    else:
                        secs, n = p.bench_inverse(0, what[1], c["dst_bytes"], c["stride_dst"], c["pred"], c["stride_pred"], c["dst_off"], c["pred_off"], c["coeffs"],
                                                  ms_each, dequant=(51, 14) if what[2] else None)
                    rates[key] = n / secs
            return rates

        with ThreadPoolExecutor(procs_n) as ex:
            per = list(ex.map(ref_worker, range(procs_n)))
        for key, _ in CPU_FAMILY_CASES:
            out["ref_%s_mblocks" % key] = round(sum(r[key] for r in per) / 1e6, 4)
    exe = os.path.join(os.path.dirname(capi.LIB_PATH), "hevcasm_host_bench")
    if os.path.exists(exe):
        mask = capi.lib().hevcasm_instruction_set_support() & ~capi.HIP
        out["simd_mask"] = "%#x" % mask

        def simd_worker(i):
            rates = {}
            for fam, blocks in (("pred", sample_pus), ("inv", (1 << 19) >> 6)):
                o = subprocess.run([exe, "%x" % mask, str(ms_each), str(blocks), str(77 + i), fam], capture_output=True, text=True,
                                   env=dict(os.environ, HEVCASM_DISABLE_HIP="1"), timeout=600)
                if o.returncode != 0:
                    raise RuntimeError("hevcasm_host_bench %s failed: %s" % (fam, o.stderr[-300:]))
                for line in o.stdout.splitlines():
                    t = line.split()
                    if t[0].startswith("pred"):
                        key = "%s_%s" % (t[0], "luma16" if t[1] == "8" else "chroma8")
                    else:
                        key = ("fused_idct%d" if t[2] == "1" else "idct%d") % (1 << int(t[1]))
                    rates[key] = int(t[-2]) / float(t[-1])
            return rates

        with ThreadPoolExecutor(procs_n) as ex:
            per = list(ex.map(simd_worker, range(procs_n)))
        for key in per[0]:
            out["simd_%s_mblocks" % key] = round(sum(r[key] for r in per) / 1e6, 4)
    return out


# ---------------------------------------------------------------------------------------------------------------------------
# The ONE line on stdout (a 26 kB line did not parse).  The line the driver reads is compact() of the full
# record: < 4 kB, strict JSON, printed last and alone on the real stdout; the full record (detail, per_family, every flattened
# key) goes to gpurun_out/bench_last.json (untracked; --full-out names another place) and nothing else may reach stdout (main() points fd 1 at stderr while it runs).

HEADLINE_FAMILIES = ("sad16", "pred_uni_luma16", "pred_bi_luma16", "pred_uni_chroma8", "pred_bi_chroma8", "idct16", "idct32", "fused_idct32", "fdct32")
# families whose headline is a TIME, not a fraction of the HBM peak: a frame's partition in one launch; one intra frame alone (dependency-limited) and 64 per call
HEADLINE_MS = ("sad_frame_runs", "sad4_frame_runs", "pred_uni_frame_runs", "pred_bi_frame_runs", "intra_frame", "intra_frames64", "tu_fused_frame", "tu_fused_32frames", "tu_residual_3stages_32frames")
PATTERN_HEADLINES = ("pred_uni_luma16", "pred_bi_luma16", "pred_uni_chroma8", "pred_bi_chroma8")
LINE_LIMIT = 4096
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_stale", "algorithmic_bytes_per_launch", "avg_launch_ms", "bytes_per_block",
                 "me_valu_frac", "me_frame_ms", "frac_nominal", "hbm_frac", "per_family_error")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "value_1core", "simd_value", "simd_value_1core", "simd_cores", "gpu_multiple", "checked_ctus", "mismatches", "simd_error", "family_error")
CONFIG_KEYS = ("workload", "blocks_per_step_per_gpu", "launches_per_step", "parallelism", "frame", "step_ms_median", "step_ms_min", "library", "library_sha256",
               "frames", "predicted_slowest_rank_ms_at_8_gpus", "me_split", "ctus_per_rank", "slowest_rank_ms", "rank0_kernel_ms", "frames_per_s", "gsad_evals_per_s")


def _finite(v):
    """Strict JSON has no NaN / Infinity: a non-finite number becomes null (recursively)."""
    if isinstance(v, float):
        return v if v == v and v not in (float("inf"), float("-inf")) else None
    if isinstance(v, dict):
        return {k: _finite(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_finite(x) for x in v]
    if isinstance(v, (np.floating,)):
        return _finite(float(v))
    if isinstance(v, (np.integer,)):
        return int(v)
    return v


def compact(full):
    """The driver's line: the contract's top-level keys; config / roofline / cpu_baseline cut down to the scalars the record needs
    (one _frac + _ms per headline family, the four _vs_pattern keys, one ref_ / simd_ pair per CPU family leg); strings <= 120."""
    cut = lambda v: v[:120] if isinstance(v, str) else v
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                 "dtype", "data", "valid") if k in full}
    cfg = full.get("config") or {}
    line["config"] = {k: cut(cfg[k]) for k in CONFIG_KEYS if k in cfg}
    if cfg.get("env_switches"):
        line["config"]["env_switches"] = ",".join(sorted(cfg["env_switches"]))[:120]
    rf = full.get("roofline") or {}
    out = {k: cut(rf[k]) for k in ROOFLINE_KEYS if k in rf and (rf[k] is not None or k in ("achieved", "peak", "frac", "traffic"))}
    for fam in HEADLINE_FAMILIES:
        for sfx in ("_frac", "_ms"):
            if fam + sfx in rf:
                out[fam + sfx] = rf[fam + sfx]
    for fam in PATTERN_HEADLINES:
        if fam + "_vs_pattern" in rf:
            out[fam + "_vs_pattern"] = rf[fam + "_vs_pattern"]
    for fam in HEADLINE_MS:
        if fam + "_ms" in rf:
            out[fam + "_ms"] = rf[fam + "_ms"]
    line["roofline"] = out
    cpu = full.get("cpu_baseline")
    if cpu is None:
        line["cpu_baseline"] = None
    else:
        c = {k: cut(cpu[k]) for k in CPU_KEYS if k in cpu and cpu[k] is not None or k in ("value", "unit", "cores", "kind") and k in cpu}
        for key, _ in CPU_FAMILY_CASES:
            for col in ("ref", "simd"):
                k = "%s_%s_mblocks" % (col, key)
                if k in cpu:
                    c[k] = cpu[k]
        line["cpu_baseline"] = c
    line["full_record"] = full.get("full_record", "gpurun_out/bench_last.json")
    return _finite(line)


def emit(full, out_fd=None, full_path=None):
    """Write the full record beside the profiles and the compact line -- alone, last, newline-terminated -- to the real stdout."""
    full = _finite(full)
    # the full record goes to an UNTRACKED place by default (gpurun_out/ is git-ignored and is merged back from the GPU box): a plain `python bench.py`
    # -- the driver's included -- must not rewrite committed evidence; the snapshots that are evidence are copied to profiles/r0N_*
    full_path = full_path or os.path.join(ROOT, "gpurun_out", "bench_last.json")
    try:
        os.makedirs(os.path.dirname(os.path.abspath(full_path)), exist_ok=True)
        with open(full_path, "w") as f:
            json.dump(full, f, indent=1, allow_nan=False)
            f.write("\n")
        full["full_record"] = os.path.relpath(full_path, ROOT)
    except OSError as e:       # a read-only tree must not cost the line
        full["full_record"] = "not written: %s" % repr(e)[:80]
    text = json.dumps(compact(full), allow_nan=False, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:      # never again a line the driver cannot read: drop the optional keys, keep the contract's
        slim = compact(full)
        slim["roofline"] = {k: v for k, v in slim["roofline"].items() if k in ROOFLINE_KEYS}
        if slim.get("cpu_baseline"):
            slim["cpu_baseline"] = {k: v for k, v in slim["cpu_baseline"].items() if k in CPU_KEYS}
        text = json.dumps(slim, allow_nan=False, separators=(",", ":"))
    if len(text) >= LINE_LIMIT or "\n" in text:      # last resort, never an exception: the contract's top-level keys and the three roofline numbers
        slim = compact(full)
        rf = slim.get("roofline") or {}
        slim["roofline"] = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
        cb = slim.get("cpu_baseline") or {}
        slim["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind")} if cb else None
        slim["config"] = {"workload": str((slim.get("config") or {}).get("workload", ""))[:120]}
        slim["truncated"] = True
        text = json.dumps(slim, allow_nan=False, separators=(",", ":")).replace("\n", " ")
    data = (text + "\n").encode()
    if out_fd is None:
        sys.stdout.flush()
        sys.stdout.write(text + "\n")
        sys.stdout.flush()
    else:
        os.write(out_fd, data)
    return text


def plumbing_run(args):
    """--plumbing: what a rank does around the measurement, without the measurement (no GPU is touched): join the gloo group, meet at the barrier,
    take the max over ranks, and let rank 0 emit the record it was handed.  The record's numbers are the test's, never a measurement."""
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo")
        dist.barrier()
    slowest = max_over_ranks(1e-3 * (rank + 1), device="cpu")
    if rank == 0:
        with open(args.plumbing) as f:
            full = json.load(f)
        full.update(n_gpus=world, steps=args.steps, warmup=args.warmup, data="canned record (--plumbing): not a measurement")
        full.setdefault("config", {})["slowest_rank_ms"] = round(slowest * 1e3, 3)
        emit(full, args.out_fd, args.full_out or os.devnull)
    if world > 1:
        dist.destroy_process_group()


def quiet_stdout():
    """Point fd 1 at stderr for the rest of the process and return a descriptor of the REAL stdout: library printf()s, child probes and stray
    print()s (tools/bench_families.py printed a JSON row of its own ahead of round 4's line) can then never precede or follow the result line."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)
    return real


def relaunch_as_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher: start `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD
    (never exec: nothing here has touched the GPU, and a child is always safe), relay rank 0's one line and the exit code."""
    import subprocess
    # --standalone: the launcher picks its own free rendezvous port (a port chosen here by bind / close / reuse could be taken in between when several benches
    # share a box: ADVICE round 5)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    for l in p.stdout.splitlines():
        if not lines or l != lines[-1]:
            print(l, file=sys.stderr)
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    return p.returncode if p.returncode or lines else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--blocks", type=int, default=1 << 20, help="blocks per size per GPU (config #2: 1 Mi)")
    ap.add_argument("--sizes", default="all", help="'all' (20 sizes >= 8x8) or a comma list like 16x16,64x64")
    ap.add_argument("--cpu-seconds", type=float, default=16.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--cpu-sample-blocks", type=int, default=8192, help="blocks per size and process of BOTH CPU columns (memory-resident at the default)")
    ap.add_argument("--cpu-procs", type=int, default=16, help="host processes of the CPU baseline legs (one GPU's share of the host)")
    ap.add_argument("--no-families", action="store_true", help="skip the per-family measurements (inter prediction, transforms, full search) after the timed region")
    ap.add_argument("--me-split", default="ctus", choices=("rows", "ctus"), help="me5: ranks own raster runs of CTUs balanced to one CTU (default; each still holds whole CTU rows + halo), or whole CTU rows")
    ap.add_argument("--workload", default="sad", choices=("sad", "me5"), help="sad = BASELINE.json configs[1] (the metric's configuration); me5 = configs[4], full search sharded by CTU rows")
    ap.add_argument("--frames", type=int, default=1, help="me5: consecutive frames per step, each rank's slab of every one of them searched in ONE launch (1 = the config literally: strong scaling of a 1.1 ms frame)")
    ap.add_argument("--plumbing", default=None, metavar="RECORD.json", help="TEST ONLY (tests/test_bench_plumbing.py): no GPU, no measurement -- the launcher / rank / "
                    "barrier / emit path alone, emitting the given full record with n_gpus = the ranks that met")
    ap.add_argument("--full-out", default=None, help="where the full record goes (default gpurun_out/bench_last.json, untracked)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_as_ranks(args, sys.argv[1:]))
    args.out_fd = quiet_stdout()
    stale = sorted(k for k in os.environ if k.startswith("HEVCASM_") and k not in ("HEVCASM_BENCH_ONE_DEVICE", "HEVCASM_DIST_BACKEND"))
    if stale:
        # (a plain print(): fd 1 points at stderr until the result line, so this cannot land beside it -- tests/test_bench_plumbing.py leans on exactly that)
        print("bench.py: WARNING: %s set in the environment -- the measurement is not of the default library configuration (recorded in config.env_switches)" % ", ".join(stale))
    if args.plumbing:
        return plumbing_run(args)

    import torch
    import torch.distributed as dist
    from hevcasm_amd import capi, workload as wl

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: measuring the %d ranks the launcher started" % (args.gpus, world, world), file=sys.stderr)
    # one process per GPU.  HEVCASM_BENCH_ONE_DEVICE=1 + HEVCASM_DIST_BACKEND=gloo is a rehearsal mode for
    # boxes with a single GPU (all ranks share cuda:0); the real multi-GPU run uses RCCL ("nccl").
    if os.environ.get("HEVCASM_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        backend = os.environ.get("HEVCASM_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)   # RCCL; used for the timing barrier / max only
        else:
            dist.init_process_group(backend)

    L = capi.lib()
    if not L.hevcasm_hip_available():
        sys.exit("no usable gfx950 device: the HIP path has no fallback")

    if args.workload == "me5":
        return bench_me5(args, rank, world, device)

    sizes = wl.SAD_BENCH_SIZES if args.sizes == "all" else [tuple(int(v) for v in s.split("x")) for s in args.sizes.split(",")]
    t0 = time.time()
    batches = [SizeBatch(w, h, args.blocks, 1000 * (rank + 1) + 17 * i, device) for i, (w, h) in enumerate(sizes)]
    torch.cuda.synchronize()
    gen_s = time.time() - t0
    launches = [(b, ways, c) for b in batches for ways in (1, 4) for c in b.chunks]

    def step(events=None):
        for (b, ways, c) in launches:
            if events is None:
                b.launch(ways, c)
            else:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                b.launch(ways, c)
                e1.record()
                events.append(((b.w, b.h, ways), c["n"], e0, e1))

    def bracket():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    events = []
    bracket()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(events)
    torch.cuda.synchronize()
    elapsed_local = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed_local)
    bracket()

    # per-kernel durations from the HIP events recorded inside the timed region
    per = {}
    for key, n, e0, e1 in events:
        d = per.setdefault(key, dict(ms=0.0, launches=0, blocks=0, each=[]))
        t = e0.elapsed_time(e1)
        d["ms"] += t
        d["each"].append(t)
        d["launches"] += 1
        d["blocks"] += n
    # the step as the sum of its launches' event times, step by step: median and min beside the mean the contract's value is made of
    per_step = len(events) // max(1, args.steps)
    step_ms = [sum(e0.elapsed_time(e1) for (_, _, e0, e1) in events[i * per_step:(i + 1) * per_step]) for i in range(args.steps)]
    detail = []
    for (w, h, ways), d in per.items():
        bytes_total = algorithmic_bytes(w, h, ways) * d["blocks"]
        gbs = bytes_total / (d["ms"] * 1e-3) / 1e9
        detail.append(dict(kernel="sad_multiref_4" if ways == 4 else "sad", size="%dx%d" % (w, h), ms_per_pass=round(d["ms"] / args.steps, 4),
                           launches_per_pass=d["launches"] // args.steps, mblocks_per_s=round(d["blocks"] / (d["ms"] * 1e-3) / 1e6, 1),
                           algorithmic_gb_per_s=round(gbs, 1), frac_of_hbm_peak=round(gbs / HBM_PEAK_GBS, 4), total_ms=d["ms"],
                           launch_ms_median=round(float(np.median(d["each"])), 4), launch_ms_min=round(min(d["each"]), 4),
                           bytes_per_block=algorithmic_bytes(w, h, ways)))
    dom = max(detail, key=lambda r: r["total_ms"])
    dom_key = (int(dom["size"].split("x")[0]), int(dom["size"].split("x")[1]), 4 if dom["kernel"] != "sad" else 1)
    dom_d = per[dom_key]
    dom_avg_s = dom_d["ms"] * 1e-3 / dom_d["launches"]
    dom_bytes = algorithmic_bytes(*dom_key) * dom_d["blocks"] / dom_d["launches"]
    for r in detail:
        del r["total_ms"]
    head = [r for r in detail if r["kernel"] == "sad" and r["size"] == "16x16"]

    # HBM bytes per launch of the dominant kernel from the PMC passes (profiles/traffic.json, made by
    # tools/make_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this command)
    # -- and only when that file was measured on THIS build of the library: it carries the sha256 of the .so the counters were taken on (round 6:
    # a kernel change that alters the traffic must not keep reporting the old bytes); otherwise traffic = null and traffic_stale says why
    traffic, traffic_stale = traffic_for("%s %s" % (dom["kernel"], dom["size"]), dom_d["blocks"] // dom_d["launches"])

    blocks_per_step = sum(c["n"] for (_, _, c) in launches)
    n_launches = len(launches)
    value = world * blocks_per_step * args.steps / elapsed / 1e6

    # The metric also names inter prediction and the IDCT, and BASELINE.json's last config is the full search: their kernels are
    # measured here, outside the timed region, on the workloads of configs #3 / #4 / #5 (tools/bench_families.py: HIP events, cold
    # inputs), so that the one JSON line carries all of it.  `value` stays the whole-job rate of config #2.
    per_family, fam_error = None, None
    if rank == 0 and world == 1 and not args.no_families:
        del batches, launches
        torch.cuda.empty_cache()
        try:
            per_family = measure_families()
        except Exception as e:   # noqa: BLE001 -- a failure here must not cost the line its timed result
            fam_error = repr(e)[:110]

    if rank == 0:
        cpu = None
        if world == 1 and args.cpu_seconds > 0:
            cpu = cpu_baseline(sizes, args.cpu_sample_blocks, args.cpu_seconds, args.cpu_procs)
        if cpu:
            cpu["gpu_multiple"] = round(value / cpu["value"], 1)      # this GPU run / the host cores above, same mix
            cpu["sample_blocks"] = args.cpu_sample_blocks
        # the host-SIMD column is measured whether or not the reference build is there: same block count per size as the
        # reference column (16 Ki blocks per size and process = up to 260 MB of frames per process: memory-resident, like the GPU's inputs)
        if world == 1 and args.cpu_seconds > 0:
            cpu = cpu or dict(value=None, unit="Mblocks/s", cores=0, kind="port", sample="reference build (oracle/_ref/refdrv) absent: host-SIMD column only")
            try:
                simd = host_simd_baseline(sizes, args.cpu_sample_blocks, min(args.cpu_seconds, 8.0), args.cpu_procs)
                if simd:
                    cpu.update(simd_value=simd["value"], simd_cores=simd["cores"], simd_gpu_multiple=round(value / simd["value"], 1), simd_value_1core=simd["value_1core"],
                               simd_sad16_mblocks=simd["sad16x16_single_ref_mblocks_per_s"], simd_sample=simd["sample"])
            except Exception as e:   # noqa: BLE001 -- a host-side failure must not discard the GPU result
                cpu["simd_error"] = repr(e)[:110]
            if not args.no_families:
                try:
                    cpu.update(cpu_family_legs(min(args.cpu_seconds, 10.0), args.cpu_procs))
                except Exception as e:   # noqa: BLE001
                    cpu["family_error"] = repr(e)[:110]
        head = head[0] if head else None
        if per_family is not None and head:
            per_family.insert(0, dict(family="sad", kernel="sad 16x16 single-ref", config="#2, inside the timed region", ms=head["ms_per_pass"], mblocks_per_s=head["mblocks_per_s"],
                                      gbs=head["algorithmic_gb_per_s"], frac=head["frac_of_hbm_peak"], bytes_per_block=head["bytes_per_block"]))
        flat = flatten_families(per_family)
        if per_family is not None:
            for key, ms in pred_pattern_ms().items():     # the prediction rows against their access pattern alone, same box, same run
                if flat.get(key + "_ms"):
                    flat[key + "_pattern_ms"] = round(ms, 4)
                    flat[key + "_vs_pattern"] = round(flat[key + "_ms"] / ms, 3)
        if cpu:   # GPU rate over the CPU legs of the same family (Mblocks/s both)
            for key, _ in CPU_FAMILY_CASES:
                g = flat.get(key + "_mblocks")
                for col in ("ref", "simd"):
                    c = cpu.get("%s_%s_mblocks" % (col, key))
                    if g and c:
                        cpu["%s_%s_gpu_multiple" % (col, key)] = round(g / c, 1)
        roofline = {"bound": "hbm", "kernel": "%s %s" % (dom["kernel"], dom["size"]), "achieved": round(dom_bytes / dom_avg_s / 1e9, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(dom_bytes / dom_avg_s / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "algorithmic_bytes_per_launch": int(dom_bytes), "avg_launch_ms": round(dom_avg_s * 1e3, 4),
                    "launch_ms_median": dom["launch_ms_median"], "launch_ms_min": dom["launch_ms_min"],
                    "bytes_per_block": algorithmic_bytes(*dom_key),
                    # what a plain read stream reaches on this part (profiles/r01_hbm_stream.txt: tools/ubench/hbm_stream.hip, 4.3 GB, streaming loads)
                    "measured_read_stream_gbs": 7043.6,
                    "traffic_over_stream": round(traffic / dom_avg_s / 1e9 / 7043.6, 4) if traffic else None,
                    "traffic_stale": traffic_stale,
                    # SURVEY 8(d)'s literal 5WH+16 counts the four overlapping candidates as disjoint: what it would read
                    "survey_5wh16_gbs": round((5 * dom_key[0] * dom_key[1] + 16) * (dom_d["blocks"] / dom_d["launches"]) / dom_avg_s / 1e9, 1) if dom_key[2] == 4 else None,
                    "note": "4-way priced at COMPULSORY bytes (src + plane share + descriptors): candidates overlap; see DESIGN.md 5",
                    "per_family_error": fam_error}
        roofline.update(flat)     # every other kernel the metric and BASELINE.json's configs name: same run, outside the timed region, scalar keys
        line = {
            "metric": "Mblocks/s (SAD-16x16, inter-pred, IDCT) batched; achieved HBM GB/s vs peak",
            "value": round(value, 1), "unit": "Mblocks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "configs[1]: SAD + 4-way SAD, %d sizes %s, %d blocks/size/GPU, ref +-7 px, frames in HBM"
                                   % (len(sizes), "8x8..64x64" if args.sizes == "all" else args.sizes, args.blocks),
                       "blocks_per_step_per_gpu": blocks_per_step, "launches_per_step": n_launches, "parallelism": "batch split x%d, no collectives" % world,
                       "frame": "%dx%d" % (wl.FRAME_W, wl.FRAME_H), "input_generation_s": round(gen_s, 1),
                       "step_ms_median": round(float(np.median(step_ms)), 3), "step_ms_min": round(min(step_ms), 3), **library_record()},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "detail": detail, "per_family": per_family,
        }
        emit(line, args.out_fd, args.full_out)
    if world > 1:
        dist.destroy_process_group()


def library_sha256():
    """Content hash of the library that is loaded: the identity profiles/traffic.json is matched on."""
    import hashlib
    from hevcasm_amd import capi
    h = hashlib.sha256()
    with open(capi.LIB_PATH, "rb") as f:
        for piece in iter(lambda: f.read(1 << 20), b""):
            h.update(piece)
    return h.hexdigest()


def traffic_for(kernel_key, blocks_per_launch, path=None):
    """(HBM bytes per launch from the PMC passes, None) when profiles/traffic.json was measured on the library now loaded and on this launch size;
    (None, reason) otherwise -- a counter figure of another build is not this build's traffic."""
    try:
        with open(path or os.path.join(ROOT, "profiles", "traffic.json")) as f:
            doc = json.load(f)
    except (OSError, ValueError):
        return None, "profiles/traffic.json absent or unreadable"
    t = doc.get(kernel_key)
    if not t:
        return None, "no PMC pass of %s on record" % kernel_key
    if t.get("blocks_per_launch") != blocks_per_launch:
        return None, "PMC pass was taken at %s blocks per launch, this run has %d" % (t.get("blocks_per_launch"), blocks_per_launch)
    meta = doc.get("_measured_on") or {}
    if not meta.get("library_sha256"):
        return None, "traffic.json carries no library hash (measured before round 6)"
    if meta["library_sha256"] != library_sha256():
        return None, "PMC passes were taken on library %s (git %s), loaded library is %s" % (meta["library_sha256"][:12], str(meta.get("git_head"))[:12], library_sha256()[:12])
    return t["hbm_bytes_per_launch"], None


def library_record():
    """Which build of the library was measured, and any HEVCASM_* environment switch that was set (there should be none)."""
    from hevcasm_amd import capi
    return dict(library=os.path.relpath(capi.LIB_PATH, ROOT), library_sha256=library_sha256()[:16],
                env_switches={k: v for k, v in os.environ.items() if k.startswith("HEVCASM_")})


# byte differences per second the chip can issue through v_qsad_pk_u16_u8 (16 per lane and instruction): measured by
# tools/ubench/qsad_rate.hip on MI355X (profiles/r03_qsad_rate.txt / .json); the nominal figure is the fallback when that record is absent
NOMINAL_QSAD_TBYTEDIFFS = 256 * 4 * 64 * 16 / 16.0 * 2.4e9 / 1e12     # quarter rate: 16 differences per lane per 16 cycles at 2.4 GHz = 157 T/s


def qsad_peak():
    """(T byte differences/s, where the number comes from).  First choice: hevcasm_amd/lib/ubench_qsad_rate run NOW on this box (a 6 ms kernel:
    8 waves per SIMD of independent v_qsad_pk_u16_u8 chains) -- boxes of this pool differ in sustained clock by more than the full search's
    distance from its roof, so a stored figure from another box misprices it either way; then the stored round-3 measurement; then the nominal rate."""
    import subprocess
    try:
        from hevcasm_amd import capi
        exe = os.path.join(os.path.dirname(capi.LIB_PATH), "ubench_qsad_rate")
        out = subprocess.run([exe, "live"], capture_output=True, text=True, timeout=120)
        v = float(out.stdout.strip().splitlines()[-1])
        if out.returncode == 0 and 50.0 < v < 400.0:
            return v, "measured live: hevcasm_amd/lib/ubench_qsad_rate (tools/ubench/qsad_rate.hip), same box, same run"
    except Exception:   # noqa: BLE001 -- no probe, no GPU, nonsense output: fall through to the stored figure
        pass
    try:
        with open(os.path.join(ROOT, "profiles", "r03_qsad_rate.json")) as f:
            return float(json.load(f)["v_qsad_pk_u16_u8_tera_byte_differences_per_s"]), "measured: tools/ubench/qsad_rate.hip, profiles/r03_qsad_rate.txt"
    except (OSError, ValueError, KeyError):
        return NOMINAL_QSAD_TBYTEDIFFS, "nominal quarter rate: 16 byte differences per lane per 16 cycles, 256 CUs x 4 SIMDs x 64 lanes at 2.4 GHz"


PATTERN_KEYS = {"idct16_2p28": "idct16_2p28", "idct32_2p28": "idct32_2p28"}
PATTERN_PROBES = ("ubench_transform_mix",)


PATTERN_ROWS = {("pred_uni_8tap", "16x16"): "pred_uni_luma16", ("pred_bi_8tap", "16x16"): "pred_bi_luma16",
                ("pred_uni_4tap Cb+Cr", "8x8"): "pred_uni_chroma8", ("pred_bi_4tap Cb+Cr", "8x8"): "pred_bi_chroma8"}


def pred_pattern_ms():
    """{family key: ms} -- what the ACCESS PATTERN of a kernel alone takes, run NOW on this box.
    Prediction (round 6): the shipped kernels themselves with their arithmetic taken out -- hevcasm_amd/lib/pattern/libhevcasm_hip.so =
    pred_inter.hip built with -DHEVCASM_PRED_PATTERN_ONLY: every load, every point where a row is first used, every store as shipped, held to the
    shipped kernels' occupancy -- timed by tools/bench_families.py pred16 (the same rows, the same inputs) in a child process that loads that library.
    Such a build does strictly less than the kernel, so <key>_vs_pattern = kernel / pattern >= 1 by construction (round 5's stand-alone probe, which the
    kernels beat by 11-26 %, was not a floor).  Inverse transforms: tools/ubench/transform_mix.hip as before.
    {} when a probe is absent or fails (the figures are a comment on the measurement, never part of it)."""
    import subprocess
    got = {}
    try:
        from hevcasm_amd import capi
        lib_dir = os.path.dirname(capi.LIB_PATH)
        pattern_lib = os.path.join(lib_dir, "pattern", "libhevcasm_hip.so")
        if os.path.exists(pattern_lib) and not os.environ.get("HEVCASM_HIP_LIB"):
            # two runs of the child, the faster figure per row: a floor is the LEAST time the pattern takes, and a single run's median of four launches is now
            # and then caught by something else on the box (seen once: 0.158 ms for a row that is 0.088 ms in every other run -- a "floor" above the kernel)
            for _ in range(2):
                out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_families.py"), "pred16"], capture_output=True, text=True, timeout=300,
                                     env=dict(os.environ, HEVCASM_HIP_LIB=pattern_lib))
                for l in out.stdout.splitlines():
                    if l.startswith("{"):
                        r = json.loads(l)
                        key = PATTERN_ROWS.get((r.get("kernel"), r.get("size")))
                        if key and 0.0 < r["ms"] < 100.0:
                            got[key] = min(got.get(key, 1e9), float(r["ms"]))
        for probe in PATTERN_PROBES:
            out = subprocess.run([os.path.join(lib_dir, probe), "live"], capture_output=True, text=True, timeout=120)
            if out.returncode != 0:
                continue
            for l in out.stdout.splitlines():
                name, _, v = l.partition(" ")
                if name in PATTERN_KEYS and PATTERN_KEYS[name] not in got and 0.0 < float(v) < 100.0:
                    got[PATTERN_KEYS[name]] = float(v)
    except Exception:   # noqa: BLE001 -- no probe, no GPU, nonsense output
        pass
    return got


def measure_families():
    """roofline.per_family: one compact row per kernel (ms per launch, algorithmic GB/s, fraction of the 8 TB/s peak)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_families as fam
    rows = []

    def collect(config, family_of):
        for r in fam.SINK:
            row = dict(family=family_of(r), kernel="%s %s" % (r["kernel"], r["size"]), config=config, ms=r["ms"], mblocks_per_s=r["mblocks_per_s"],
                       gbs=r["algorithmic_gb_per_s"], frac=r["frac_of_hbm_peak"], bytes_per_block=r["bytes_per_block"])
            for k in ("frames", "launches", "note"):
                if k in r:
                    row[k] = r[k]
            rows.append(row)
        fam.SINK = []

    fam.SINK = []
    fam.bench_pred(frames=64, only=((16, 16),))            # luma 8-tap 16x16 + chroma 4-tap 8x8, uni and bi, all fractional phases
    collect("#3, 64 frames of 1080p per launch", lambda r: "inter_pred")
    fam.bench_pred(frames=256, only=((16, 16),))           # the same rows at 4x the PUs per launch: ramp-up and tail against the steady state
    collect("#3, 256 frames of 1080p per launch", lambda r: "inter_pred_256f")
    fam.bench_pred_latency()                                  # the same PUs, ONE frame per launch; and luma + Cb + Cr of a frame as one HIP graph
    collect("#3, one 1080p frame", lambda r: "inter_pred_latency")
    fam.bench_cost_latency()                                  # the block costs of a frame's partition (24 PU sizes) in one launch, and as a launch per size
    collect("#2's kernels, one 1080p frame's partition", lambda r: "block_cost_latency")
    fam.bench_intra_frames()                                  # whole intra frames in wavefront order, one alone and 64 per call
    collect("8f-3, 1080p intra frames", lambda r: "intra_frames")
    fam.bench_tu_latency()                                    # the residual loop of a frame's TUs (every transform size): fused, one launch per stage, a launch per size and stage
    collect("#4's kernels, a 1080p frame's TUs", lambda r: "tu_latency")
    fam.bench_transform(only=((0, 3), (0, 4), (0, 5)))      # inverse (+ fused dequantisation) and forward 8x8 / 16x16 / 32x32
    collect("#4, 2^26 coefficients per launch", lambda r: "idct" if "inverse" in r["kernel"] else "fdct")
    fam.bench_transform(1 << 28, only=((0, 4), (0, 5)))       # the same kernels nearer their steady state: a 60 us launch pays 10-13 us of ramp-up and tail
    collect("#4, 2^28 coefficients per launch", lambda r: "idct_large_launch" if "inverse" in r["kernel"] else "fdct_large_launch")
    me = fam.bench_me_frame(quiet=True)
    fam.SINK = None
    peak, src = qsad_peak()
    rows.append(dict(family="full_search", kernel="me_full_search 64x64 (+ 64x48 row) +-64", config="#5, one 3840x2160 frame = 2040 CTUs, one launch",
                     ms=round(me["ms"], 4), mblocks_per_s=round(me["ctus"] / me["ms"] / 1e3, 3), frames_per_s=round(1e3 / me["ms"], 1),
                     gbs=round(me["bytes"] / me["ms"] / 1e6, 1), frac=round(me["bytes"] / me["ms"] / 1e6 / HBM_PEAK_GBS, 4),
                     bound="valu", gsad_evals_per_s=round(me["sad_evals"] / me["ms"] / 1e6, 2),
                     achieved=round(me["byte_diffs"] / me["ms"] / 1e9, 2), peak=round(peak, 1), unit="T byte differences/s",
                     frac_of_valu=round(me["byte_diffs"] / me["ms"] / 1e9 / peak, 4), peak_source=src,
                     frac_of_valu_nominal=round(me["byte_diffs"] / me["ms"] / 1e9 / NOMINAL_QSAD_TBYTEDIFFS, 4)))
    torch.cuda.empty_cache()
    return rows


def gather_me_results(part, frame_w, frame_h, ctu, device="cpu", unit="rows"):
    """Collect every rank's (first, last, results) into the frame's table on rank 0 (None elsewhere); first / last count CTU rows
    (unit="rows") or CTUs in raster order (unit="ctus").  The answers are 12 bytes per CTU, collected after the search: each rank
    writes its CTUs into a table of INT_MIN and its ownership into a vector of zeros; MAX / SUM all-reduces merge them, and the
    ownership vector proves every CTU had exactly one owner."""
    import torch
    import torch.distributed as dist
    from hevcasm_amd import me_frame
    rows, cols = -(-frame_h // ctu), -(-frame_w // ctu)
    a, b, res = part
    if unit == "rows":
        a, b = a * cols, b * cols
    if not (dist.is_available() and dist.is_initialized()):
        return me_frame.assemble_ctus([(a, b, res)], frame_w, frame_h, ctu)
    table = torch.full((rows * cols, 3), -2 ** 31, dtype=torch.int32)
    table[a:b] = torch.from_numpy(np.asarray(res, dtype=np.int32).reshape(-1, 3))
    owners = torch.zeros(rows * cols, dtype=torch.int32)
    owners[a:b] = 1
    table, owners = table.to(device), owners.to(device)
    dist.all_reduce(table, op=dist.ReduceOp.MAX)
    dist.all_reduce(owners, op=dist.ReduceOp.SUM)
    assert bool((owners == 1).all()), "CTUs without exactly one owner"
    return table.cpu().numpy() if dist.get_rank() == 0 else None


def bench_me5(args, rank, world, device):
    """BASELINE.json configs[4]: a step = full search (+-64) of one 3840x2160 frame, CTU rows sharded across the ranks."""
    import torch
    import torch.distributed as dist
    from hevcasm_amd import batch as B, me_frame, shard
    fw, fh, ctu, R = 3840, 2160, 64, 64
    rows, cols = -(-fh // ctu), -(-fw // ctu)
    # the rank's CTUs: whole rows or, --me-split ctus, a raster run balanced to one CTU (255 each at 8 ranks where rows are 300 or 240)
    if args.me_split == "rows":
        row0, row1 = shard.ctu_row_range(rows, rank, world)
        first, last = row0 * cols, row1 * cols
    else:
        first, last = shard.ctu_range(rows * cols, rank, world)
        row0, row1 = shard.rows_of_ctus(first, last, cols)
    # a few distinct groups of F frame pairs, rotated over the steps; every rank builds the same frames from the seed and keeps only its slabs
    # (the rows its CTUs lie in + reference rows with the halo, F of them stacked) -- its H2D copy is the slabs, never the frames
    F = max(1, args.frames)
    n_pairs = 3
    slabs, frames0 = [], None
    for i in range(n_pairs):
        group = [me_frame.synthetic_frame_pair(fw, fh, R, seed=500 + i + 10 * k) for k in range(F)]
        if i == 0 and rank == 0:
            frames0 = group[0]
        if last > first:
            st = me_frame.CtuRunStack(group, ctu, R, first, last)
            slabs.append(dict(stack=st, cur=torch.from_numpy(st.cur).to(device), ref=torch.from_numpy(st.ref).to(device),
                              out=torch.empty((F, last - first, 3), dtype=torch.int32, device=device)))
        del group
    torch.cuda.synchronize()

    def step(i):
        if not slabs:
            return
        d = slabs[i % n_pairs]
        st = d["stack"]
        sl = st.slab
        B.me_full_search_frames_ctus(d["cur"], sl.cur_byte0, sl.stride_cur, st.pitch_cur, d["ref"], sl.ref_byte0, sl.stride_ref, st.pitch_ref, F, fw, fh, ctu, R,
                                     first, last - first, out=d["out"])

    def bracket():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    bracket()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        step(i)
    ev1.record()
    torch.cuda.synchronize()
    elapsed_local = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed_local)
    bracket()
    kernel_ms = ev0.elapsed_time(ev1) / max(1, args.steps)          # this rank's launches back to back: the kernel's average duration
    # the answers of the first frame pair, from every rank (outside the timed region)
    step(0)
    torch.cuda.synchronize()
    mine = slabs[0]["out"].cpu().numpy() if slabs else np.zeros((F, 0, 3), dtype=np.int32)
    gather_dev = device if world > 1 and dist.get_backend() == "nccl" else "cpu"
    table = gather_me_results((first, last, mine[0]), fw, fh, ctu, device=gather_dev, unit="ctus")      # frame 0 of the group: checked against the reference below
    # what the slowest of 8 ranks would take for its share of these F frames, measured HERE when this is a one-GPU run: the largest 8-way run (255 CTUs), F slabs, one launch
    predicted8 = None
    if world == 1 and slabs:
        a8, b8 = shard.ctu_range(rows * cols, 0, 8)
        group = [me_frame.synthetic_frame_pair(fw, fh, R, seed=500 + 10 * k) for k in range(F)]
        st8 = me_frame.CtuRunStack(group, ctu, R, a8, b8)
        c8, r8 = torch.from_numpy(st8.cur).to(device), torch.from_numpy(st8.ref).to(device)
        o8 = torch.empty((F, b8 - a8, 3), dtype=torch.int32, device=device)
        run8 = lambda: B.me_full_search_frames_ctus(c8, st8.slab.cur_byte0, st8.slab.stride_cur, st8.pitch_cur, r8, st8.slab.ref_byte0, st8.slab.stride_ref, st8.pitch_ref, F,
                                                    fw, fh, ctu, R, a8, b8 - a8, out=o8)
        run8(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run8()
        e1.record(); torch.cuda.synchronize()
        predicted8 = e0.elapsed_time(e1) / 5
        assert torch.equal(o8[0].cpu(), torch.from_numpy(mine[0][a8:b8])), "the 255-CTU launch of F slabs disagrees with the whole-frame launch"
        del group, st8, c8, r8, o8

    exit_code = 0
    if rank == 0:
        n_ctus = rows * cols
        heights = [min(ctu, fh - r * ctu) for r in range(rows)]
        byte_diffs = sum(cols * ctu * h for h in heights) * (2 * R + 1) ** 2
        sad_evals = n_ctus * (2 * R + 1) ** 2
        alg_bytes = sum(cols * (ctu * h + (ctu + 2 * R) * (h + 2 * R) + 12) for h in heights)     # source + window + result per CTU
        my_diffs = sum(ctu * heights[g // cols] for g in range(first, last)) * (2 * R + 1) ** 2
        peak, peak_src = qsad_peak()
        # CPU baseline: the reference's own hevcasm_sad composed per candidate (refdrv), a bounded sample of this frame's CTUs,
        # one process per core -- and the same CTUs checked against the GPU's table
        cpu = None
        if args.cpu_seconds > 0:
            cpu = me_cpu_baseline(frames0, table, ctu, R, min(args.cpu_seconds, 12.0), args.cpu_procs)
        value = n_ctus * F * args.steps / elapsed / 1e6
        achieved = my_diffs * F / (kernel_ms * 1e-3) / 1e12
        checked = (cpu or {}).get("checked_ctus")
        mismatches = (cpu or {}).get("mismatches")
        line = {
            "metric": "Mblocks/s (SAD-16x16, inter-pred, IDCT) batched; achieved HBM GB/s vs peak",
            "value": round(value, 4), "unit": "Mblocks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            # valid: the GPU's answers equal the reference's on every CTU checked; null when the check did not run (no refdrv, --cpu-seconds 0)
            "valid": None if checked is None else mismatches == 0,
            "config": {"workload": "configs[4]: full search +-64, %d x 3840x2160 frame/step = %d CTUs (33 rows 64x64 + one 64x48 each); block = CTU" % (F, F * n_ctus),
                       "frames": F, "predicted_slowest_rank_ms_at_8_gpus": round(predicted8, 4) if predicted8 else None,
                       "parallelism": "%s split x%d, rows + 64-row halo per rank, no collective" % ("CTU rows" if args.me_split == "rows" else "CTU runs", world),
                       "me_split": args.me_split, "ctus_per_rank": last - first, "rank0_first_ctu": first, "rank0_rows": "%d..%d" % (row0, row1),
                       "slowest_rank_ms": round(elapsed / args.steps * 1e3, 4), "rank0_kernel_ms": round(kernel_ms, 4),
                       "frames_per_s": round(F * args.steps / elapsed, 1), "gsad_evals_per_s": round(sad_evals * F * args.steps / elapsed / 1e9, 2), **library_record()},
            "roofline": {"bound": "valu", "kernel": "me_full_search_kernel<frame>", "achieved": round(achieved, 2), "peak": round(peak, 1),
                         "unit": "T byte differences/s", "frac": round(achieved / peak, 4), "traffic": None,
                         "frac_nominal": round(achieved / NOMINAL_QSAD_TBYTEDIFFS, 4), "peak_nominal": round(NOMINAL_QSAD_TBYTEDIFFS, 1),
                         "avg_launch_ms": round(kernel_ms, 4), "peak_source": peak_src[:118],
                         "hbm_algorithmic_bytes_per_frame": alg_bytes, "hbm_gbs": round(alg_bytes * F * args.steps / elapsed / 1e9, 1),
                         "hbm_frac": round(alg_bytes * F * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, 5),
                         "note": "bound by the v_qsad_pk_u16_u8 issue rate, not HBM (1.7 k byte differences per byte fetched); rank 0's launches"},
            "cpu_baseline": cpu,
        }
        emit(line, args.out_fd, args.full_out)
        if mismatches:
            print("bench.py: %d of %d CTUs differ from the reference's hevcasm_sad composition" % (mismatches, checked), file=sys.stderr)
            exit_code = 1
    if world > 1:
        dist.destroy_process_group()
    if exit_code:
        sys.exit(exit_code)


def me_cpu_baseline(frame, gpu_table, ctu, R, budget_s, max_procs):
    """The reference's own hevcasm_sad (C path) composed per candidate, one refdrv process per core: process k searches CTUs
    k, k + P, k + 2P ... of the frame for its share of `budget_s`, each CTU handed over as a cut-out holding its window;
    every CTU searched is also compared with the GPU's answer."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pyref
    if not pyref.available():
        return None
    procs_n, cores_info = host_cores(max_procs)
    fw, fh, pad = frame["w"], frame["h"], frame["pad"]
    rows, cols = -(-fh // ctu), -(-fw // ctu)
    cur2, ref2 = frame["cur"].reshape(fh, -1), frame["ref"].reshape(fh + 2 * pad, -1)
    last_row = (rows - 1) * cols

    def worker(k):
        order = [last_row + k % cols] + list(range(k, rows * cols, procs_n))       # one CTU of the short bottom row first, then a stride over the frame
        t, out = 0.0, []
        with pyref.RefProc() as p:
            for idx in order:
                r, c = divmod(int(idx), cols)
                h = min(ctu, fh - r * ctu)
                y0, x0 = r * ctu, c * ctu
                cur = cur2[y0:y0 + h, x0:x0 + ctu].copy()
                ref = ref2[y0 + pad - R: y0 + pad + h + R, x0 + pad - R: x0 + pad + ctu + R].copy()
                t0 = time.perf_counter()
                res = p.me_full_search(cur.reshape(-1), ctu, ref.reshape(-1), ctu + 2 * R, np.array([0], dtype=np.uint32),
                                       np.array([R * (ctu + 2 * R) + R], dtype=np.uint32), ctu, h, R)
                t += time.perf_counter() - t0
                out.append((int(idx), res[0]))
                if t > budget_s:
                    break
        return t, out

    with ThreadPoolExecutor(procs_n) as ex:
        res = list(ex.map(worker, range(procs_n)))
    n = sum(len(o) for _, o in res)
    rate = sum(len(o) / t for t, o in res if t > 0)          # CTUs per second, all processes
    bad = [idx for _, o in res for idx, r in o if not np.array_equal(r, gpu_table[idx])]
    return dict(value=round(rate / 1e6, 6), unit="Mblocks/s", cores=procs_n, kind="reference", ctus_per_s=round(rate, 2),
                gsad_evals_per_s=round(rate * (2 * R + 1) ** 2 / 1e9, 4),
                sample="reference hevcasm_sad (refdrv) per candidate, +-%d: %d CTUs (stride over all rows), %d procs, <= %.0f s each" % (R, n, procs_n, budget_s),
                checked_ctus=n, mismatches=len(bad), **cores_info)


if __name__ == "__main__":
    main()
