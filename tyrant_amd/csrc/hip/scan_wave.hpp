// scan_wave.hpp -- k_scan_words' work (frame.hip: survive bytes -> the tables v_lookup() reads) done by WAVES of another kernel:
// one wave per 16384-slot block, four rounds of 64 words with a running carry, no LDS, no barrier.  k_trace_flat calls it on its way
// in (FrameParams::scanPrevInTrace): the scan of iteration i is only read by shade(i + 1), the traversal launch of iteration i + 1
// needs none of it -- so that launch does it, beside its own start, and the k_scan_words launch in front of it (and its gap) is gone.
#pragma once
#include "device_common.hpp"

namespace tyr {

constexpr uint32_t kScanWaveBlockSlots = 64u * 256u; // = frame.hip kScanBlockSlots: the tables' block size

// the wave's block `sb`: words, prefixes inside the block, the block's total; returns true when this wave finished the LAST block
__device__ __forceinline__ void scan_blocks_by_wave(uint8_t* __restrict__ survFlag, unsigned long long* __restrict__ vWord, uint32_t* __restrict__ vPre, uint32_t* __restrict__ vBlk,
                                                             uint32_t* __restrict__ blocksDone, uint32_t n, uint32_t firstBlock, uint32_t strideBlocks, bool clearBytes) {
	const uint32_t lane = lane_id();
	const uint32_t nBlocks = (n + kScanWaveBlockSlots - 1) / kScanWaveBlockSlots;
	for (uint32_t sb = firstBlock; sb < nBlocks; sb += strideBlocks) {
		uint32_t carry = 0;
		for (uint32_t round = 0; round < 4u; ++round) {
			const uint32_t e = sb * 256u + round * 64u + lane, first = e * 64u;
			unsigned long long word = 0ull;
			if (first < n) {
				uint4* p = reinterpret_cast<uint4*>(survFlag + first);
#pragma unroll
				for (int k = 0; k < 4; ++k) {
					const uint4 q = p[k];
					if (clearBytes) // (k_scan_words: P.retireGhosts)
						p[k] = make_uint4(0u, 0u, 0u, 0u);
					const uint32_t x[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
					for (int j = 0; j < 4; ++j) {
						const uint32_t b = (x[j] & 1u) | ((x[j] >> 7) & 2u) | ((x[j] >> 14) & 4u) | ((x[j] >> 21) & 8u);
						word |= (unsigned long long)b << (16 * k + 4 * j);
					}
				}
				if (n - first < 64u)
					word &= (1ull << (n - first)) - 1ull;
			}
			const uint32_t c = (uint32_t)__popcll(word);
			uint32_t incl = c;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) {
				const uint32_t v = __shfl_up(incl, o, 64);
				if (lane >= (uint32_t)o)
					incl += v;
			}
			vWord[e] = word;
			vPre[e] = carry + incl - c;
			carry += (uint32_t)__shfl(incl, 63, 64);
		}
		// the blocks' totals become exclusive prefixes in the wave that finishes last (agent-scope atomics: frame.hip's invariant)
		uint32_t last = 0;
		if (lane == 0) {
			__hip_atomic_store(&vBlk[sb], carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
			last = atomicAdd(blocksDone, 1u) + 1u == nBlocks ? 1u : 0u;
		}
		last = (uint32_t)__shfl(last, 0, 64);
		if (last) {
			uint32_t run = 0;
			for (uint32_t base = 0; base < nBlocks; base += 64u) {
				const uint32_t i = base + lane;
				const uint32_t v = i < nBlocks ? __hip_atomic_load(&vBlk[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
				uint32_t inc = v;
#pragma unroll
				for (int o = 1; o < 64; o <<= 1) {
					const uint32_t u = __shfl_up(inc, o, 64);
					if (lane >= (uint32_t)o)
						inc += u;
				}
				if (i < nBlocks)
					vBlk[i] = run + inc - v;
				run += (uint32_t)__shfl(inc, 63, 64);
			}
			if (lane == 0)
				*blocksDone = 0u; // (as k_scan_words leaves it)
		}
	}
}

} // namespace tyr
