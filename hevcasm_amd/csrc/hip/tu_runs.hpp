// tu_runs.hpp -- the run table of the transform-unit `_runs` entry points (transform.hip, quant.hip; round 6).
//
// A frame's residual is cut into transform units of four sizes (4x4 DST or DCT, 8x8, 16x16, 32x32: residual_decode.h:54, :82); a launch per size and stage is
// 10-20 us each for data that moves in a few microseconds.  As for the prediction and block-cost `_runs` entry points the host lists the frame's TUs as RUNS --
// `count` consecutive descriptors of one size -- and one launch serves them all: a workgroup finds its run by comparing its index with the runs' first
// workgroups (the table travels in the kernel arguments) and runs the one-size kernel's own workgroup body for that size.
#pragma once

#include <vector>
#include <algorithm>

#include "common.hpp"

namespace hevc {

constexpr int kMaxTuRuns = 32;     // runs per launch (sizes x planes of a frame: 5 x 3); more runs = more launches, made by the call

struct TuRunDev
{
	uint32_t wg_count;             // workgroups of the run
	uint32_t first, count;         // descriptor entries
	uint32_t coeff_first;          // first int16 element of the run's coefficient blocks
	int scale, shift, offset;      // the run's (de)quantiser parameters
	int dscale, dshift;            // the dequantiser's beside the quantiser's (the fused transform + quantise + reconstruct kernel only)
	uint32_t variant;              // which workgroup body
	uint32_t plane;                // which plane of the call (the fused residual loop over a frame's planes; 0 elsewhere)
};

struct TuRunTable
{
	uint32_t wg_start[kMaxTuRuns];     // first workgroup of run i; unused entries hold 0xffffffff
	TuRunDev runs[kMaxTuRuns];
};

// the run a workgroup belongs to and its index inside it (wave-uniform: everything here lives in scalar registers)
__device__ __forceinline__ TuRunDev find_tu_run(const TuRunTable &t, uint32_t &wg_in_run)
{
	uint32_t at_or_before = 0, start = 0;
#pragma unroll
	for (int i = 0; i < kMaxTuRuns; ++i)
	{
		const bool in = t.wg_start[i] <= blockIdx.x;
		at_or_before += in ? 1u : 0u;
		start = in ? t.wg_start[i] : start;      // starts ascend: the last one taken is the run's
	}
	const TuRunDev r = t.runs[at_or_before - 1];
	wg_in_run = xcd_contiguous(blockIdx.x - start, r.wg_count);
	return r;
}

// host: the caller's runs -> device tables of at most kMaxTuRuns runs each, LARGEST units first (the launch ends when its last wave does, and a 32x32
// unit that starts in the last generation of workgroups would be the whole tail).  `variant_of(run, &blocks_per_wg)` validates a run and names its body.
template <class VariantOf>
static int build_tu_tables(const hevcasm_hip_tu_run *runs, int n_runs, VariantOf variant_of, std::vector<TuRunTable> &tables, std::vector<uint32_t> &grids,
		const hevcasm_hip_tu_run *dequant_runs = nullptr)
{
	std::vector<TuRunDev> all;
	all.reserve((size_t)n_runs);
	for (int i = 0; i < n_runs; ++i)
	{
		uint32_t blocks_per_wg = 0, variant = 0;
		if (int e = variant_of(runs[i], variant, blocks_per_wg)) return e;
		const uint32_t n2 = 1u << (2 * runs[i].log2TrafoSize);
		if ((uint64_t)runs[i].first + runs[i].count > 0xffffffffull) return set_error(HEVCASM_HIP_EINVAL, "run exceeds 2^32-1 descriptors");
		if ((uint64_t)runs[i].coeff_first + (uint64_t)runs[i].count * n2 > 0xffffffffull) return set_error(HEVCASM_HIP_EINVAL, "run's coefficients exceed 2^32-1 elements");
		if (runs[i].count == 0) continue;
		TuRunDev d;
		d.wg_count = ceil_div(runs[i].count, blocks_per_wg);
		d.first = runs[i].first; d.count = runs[i].count; d.coeff_first = runs[i].coeff_first;
		d.scale = runs[i].scale; d.shift = runs[i].shift; d.offset = runs[i].offset;
		d.dscale = dequant_runs ? dequant_runs[i].scale : 0; d.dshift = dequant_runs ? dequant_runs[i].shift : 0;
		d.variant = variant;
		d.plane = runs[i].plane;
		all.push_back(d);
	}
	std::stable_sort(all.begin(), all.end(), [](const TuRunDev &a, const TuRunDev &b) { return a.variant > b.variant; });   // variants ascend with the unit's size
	for (size_t base = 0; base < all.size(); base += kMaxTuRuns)
	{
		TuRunTable t = {};
		uint32_t wgs = 0;
		for (int i = 0; i < kMaxTuRuns; ++i) t.wg_start[i] = 0xffffffffu;
		for (size_t i = base, used = 0; i < all.size() && i < base + kMaxTuRuns; ++i, ++used)
		{
			t.runs[used] = all[i];
			t.wg_start[used] = wgs;
			if ((uint64_t)wgs + all[i].wg_count > 0x7fffffffull) return set_error(HEVCASM_HIP_EINVAL, "too many workgroups in one launch");
			wgs += all[i].wg_count;
		}
		tables.push_back(t);
		grids.push_back(wgs);
	}
	return 0;
}

}  // namespace hevc
