// prologue.hpp -- what opens a wavefront iteration on the device: set_wavefront_globals (kernel.cu:227-244) and the padding of the
// queue segments' ends.  Run by the last block of k_primary (frame.hip), of k_scan_words (frame.hip, FrameParams::foldNextPrologue)
// or of k_shade (shade.hip, FrameParams::shadeOpensNext) -- whichever kernel is the last in front of the iteration's traversal launch.
#pragma once
#include "device_common.hpp"

namespace tyr {

// ======================================================================================
// set_wavefront_globals, kernel.cu:227-244 (+ reset of the compaction descriptors)
// ======================================================================================
// Runs in the LAST block of k_primary to finish (every other block has read the counts it changes): one launch and
// one gap between dependent kernels fewer per iteration (20 iterations per frame at the reference's queue size).
// segNext / kc: the segment counters the iteration's shade appends to (its next ray queue, its shadow queue).  For the iteration
// the launch belongs to these are P.segNext / P.kc; k_scan_words' last block runs the same for the iteration BEHIND its own
// (P.foldNextPrologue), whose roles are this iteration's work queue and the other set of shadow counters.
// cntKnown: the caller has the survivor count in a register (k_shade's last block: it read the counter with an agent-scope load;
// a plain load there could be served from a cache line this kernel fetched before the other blocks' atomics).
__device__ __forceinline__ void wavefront_globals_for(const FrameParams& P, uint32_t* segNext, ConnectCounters* kc, bool cntKnown = false, uint32_t cntIn = 0u) {
	const uint32_t i = threadIdx.x;
	if (i < kTicketWords) {
		P.k->extend_chunks[i * 32] = 0;
		kc->chunks[i * 32] = 0;
		P.k->shade_tiles[i * 32] = 0;
		// what this iteration's shade appends to: the next ray queue and this iteration's shadow queue
		segNext[i * kSegStride] = 0;
		segNext[kClassWords + i * kSegStride] = 0;
		kc->seg[i * kSegStride] = 0;
	}
	if (i == 0) {
		DevCounters* k = P.k;
		const uint32_t cnt = cntKnown ? cntIn : k->primary_ray_cnt;
		// (run for the NEXT iteration by k_scan_words: the host's snapshot of this iteration's counts is copied out behind that
		// launch -- the two it steers by are kept where the reset below does not reach them)
		k->reserved0 = cnt;
		k->reserved1 = k->shadow_ray_cnt;
		const unsigned long long room = (unsigned long long)(P.N - cnt);
		const unsigned long long budget = k->budget_remaining;
		const uint32_t nNew = (uint32_t)(room < budget ? room : budget);
		k->start_position = (uint32_t)(((unsigned long long)k->start_position + nNew) % P.localPixels);
		k->n_live = cnt + nNew;
		k->shade_blocks_done = 0;
		k->scan_blocks_done = 0;
		k->primary_blocks_done = 0;
		for (uint32_t w = 0; w < kTicketWords; ++w)
			k->primary_done[w * 32] = 0;
		k->shadow_ray_cnt = 0;
		k->primary_ray_cnt = 0;
		kc->shadow_cnt = 0;
		if (budget != ~0ull)
			k->budget_remaining = budget - nNew;
		k->total_primary_rays += nNew;
		k->total_extend_rays += cnt + nNew;
#if defined(TYR_QUAD_STATS) || defined(TYR_LAUNCH_ANATOMY)
		k->debug[13] = k->debug[14] = k->debug[15] = 0ull; // launch anatomy of this iteration's extend (tools/launch_tail.py)
		k->debug[9] = k->debug[10] = k->debug[11] = k->debug[12] = 0ull; // ... and its longest rays (k_trace_flat, TYR_QUAD_STATS)
#endif
	}
}

__device__ __forceinline__ void wavefront_globals(const FrameParams& P) { wavefront_globals_for(P, P.segNext, P.kc); }

// the slots at the ends of a queue's eight segments that hold no record, up to the queue's extent, become rays that enter
// nothing (what k_pad_holes does with a block per segment; here: one block, for k_scan_words' last block)
__device__ __forceinline__ void pad_work_holes_block(const RayQ& q, const uint32_t* cnt) {
	const uint32_t ext = queue_extent(cnt) / kSegs;
	for (uint32_t w = 0; w < kSegs; ++w)
		for (uint32_t j = cnt[w * kSegStride] + threadIdx.x; j < ext; j += kBlock)
			write_dead_ray(q, seg_phys(w, j));
}
__device__ __forceinline__ void pad_shadow_holes_block(const ShadowQ& q, const uint32_t* cnt) {
	const uint32_t ext = queue_extent(cnt) / kSegs;
	for (uint32_t w = 0; w < kSegs; ++w)
		for (uint32_t j = cnt[w * kSegStride] + threadIdx.x; j < ext; j += kBlock)
			reinterpret_cast<float*>(&q.color[seg_phys(w, j)])[3] = 1.0f;
}
// the same two with the eight counts handed in (cnt8[w], e.g. in LDS: k_shade's last block reads them with agent-scope loads)
__device__ __forceinline__ uint32_t extent_per_segment(const uint32_t* cnt8) {
	uint32_t m = 0;
	for (uint32_t w = 0; w < kSegs; ++w)
		m = cnt8[w] > m ? cnt8[w] : m;
	return ((m + 63u) >> 6) * 64u;
}
__device__ __forceinline__ void pad_work_holes_counts(const RayQ& q, const uint32_t* cnt8) {
	const uint32_t ext = extent_per_segment(cnt8);
	for (uint32_t w = 0; w < kSegs; ++w)
		for (uint32_t j = cnt8[w] + threadIdx.x; j < ext; j += kBlock)
			write_dead_ray(q, seg_phys(w, j));
}
__device__ __forceinline__ void pad_shadow_holes_counts(const ShadowQ& q, const uint32_t* cnt8) {
	const uint32_t ext = extent_per_segment(cnt8);
	for (uint32_t w = 0; w < kSegs; ++w)
		for (uint32_t j = cnt8[w] + threadIdx.x; j < ext; j += kBlock)
			reinterpret_cast<float*>(&q.color[seg_phys(w, j)])[3] = 1.0f;
}


} // namespace tyr
