// bvh_layout_dev.hip -- host/bvh_layout.cpp's pass ON THE DEVICE: the reference's flat depth-first 32-byte nodes (bvh.h:55-68) and
// 40-byte triangles (loader.h:13-19), already in device memory, become the 128-byte quad records and 48-byte triangles the
// traversal kernels read (hip/traverse.hpp) -- the same bytes the host pass writes (tests/test_bvh_build_device.py compares the
// arrays), without the tree ever crossing the bus as records: tyr_scene_upload ships 32 + 40 bytes per node / triangle instead of
// laying out and shipping 128 + 48, and tyr_scene_build_upload (the tree built by hip/bvh_build_dev.hip) ships no node at all.
//
// What the host pass does per subtree range on its threads is done here per node:
//   * depth parity (every interior node at even depth is the root of a quad record) flows down the tree level by level -- a
//     breadth-first list of the nodes, built with one atomic per BLOCK and level (a single word serves ~88 atomics a microsecond:
//     one per node would be 0.14 s on C5's 12.6 M nodes); the list also proves the array is a tree (every node reached once);
//   * the quad numbering is a prefix sum in array order; the 64 records of the top move to the front in breadth-first order
//     (one wave walks them, a generation of the queue at a time), everything else keeps its order -- new_index() is a search
//     in that 64-entry table;
//   * a record is written once, at its final place with its final references;
//   * the deepest stack a traversal can need flows UP the same level lists, deepest level first.
// What it leaves to the host pass (TYR_ERR_UNSUPPORTED, the caller falls back): the pair nodes of the counting build, leaves longer
// than kMaxLeafPrims (chains of synthetic records), a tree that is one leaf, trees deeper than kMaxLevels -- and every malformed
// input, so that the error code is the host pass's.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstring>

#include "../host/host.hpp"

namespace tyr {

namespace {

constexpr int kB = 256;
constexpr uint32_t kMaxLevels = 2048;
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kScanPer = 1024; // items per block of the prefix sum

struct LayState {
	uint32_t bad;         // malformed input
	uint32_t notHere;     // well-formed, but the host pass's business (over-long leaf)
	uint32_t nRealQuads;
	uint32_t nTop;
	uint32_t quadRootRef;
	uint32_t maxStack;
	uint32_t topSortedOld[kStagedNodes], topSortedNew[kStagedNodes];
	float rootMin[3], rootMax[3];
	uint32_t levelOffset[kMaxLevels + 2]; // the breadth-first list: level l = list[levelOffset[l], levelOffset[l] + levelCount[l])
	uint32_t levelCount[kMaxLevels + 2];
};

__device__ __forceinline__ bool finite3(const float* p) { return isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]); }
__device__ __forceinline__ uint32_t leaf_ref_d(uint32_t off, uint32_t cnt) { return kRefLeaf | ((cnt - 1u) << 26) | off; }

// ---- triangles: 40-byte records -> 3 x dwordx4 (bvh_layout.cpp "triangles") ----
__global__ void k_lay_tris(const tyr_triangle* __restrict__ prims, float4* __restrict__ tris, int n, LayState* S) {
	const int i = blockIdx.x * kB + threadIdx.x;
	if (i >= n)
		return;
	const tyr_triangle t = prims[i];
	if (!finite3(t.vert) || !finite3(t.e1) || !finite3(t.e2))
		S->bad = 1u;
	tris[3 * i + 0] = make_float4(t.vert[0], t.vert[1], t.vert[2], t.e1[0]);
	tris[3 * i + 1] = make_float4(t.e1[1], t.e1[2], t.e2[0], t.e2[1]);
	tris[3 * i + 2] = make_float4(t.e2[2], __uint_as_float((uint32_t)t.materialType), __uint_as_float((uint32_t)t.pad_[0]), 0.0f);
}

// ---- every node on its own (bvh_layout.cpp "validate"); seen[] zeroed for the level passes ----
__global__ void k_lay_validate(const tyr_bvh_node* __restrict__ nodes, uint32_t* __restrict__ seen, int nNodes, int nPrims, LayState* S) {
	const int i = blockIdx.x * kB + threadIdx.x;
	if (i >= nNodes)
		return;
	seen[i] = 0u;
	const tyr_bvh_node n = nodes[i];
	bool bad = !finite3(n.bbox.bounds[0]) || !finite3(n.bbox.bounds[1]);
	if (n.primitiveCount > 0) {
		bad = bad || n.offset < 0 || (long long)n.offset + n.primitiveCount > nPrims;
		if (n.primitiveCount > kMaxLeafPrims || i == 0)
			S->notHere = 1u;
	} else {
		bad = bad || n.splitAxis > 2 || (long long)n.offset <= (long long)i + 1 || n.offset >= nNodes || i + 1 >= nNodes;
	}
	if (bad)
		S->bad = 1u;
	if (i == 0) {
		for (int k = 0; k < 3; ++k) {
			S->rootMin[k] = n.bbox.bounds[0][k];
			S->rootMax[k] = n.bbox.bounds[1][k];
		}
	}
}

// ---- one level of the tree: the nodes of level `level` name their children; seen[c] = 1 | parity << 1 ----
// (a fixed grid walks the level's list; a launch for a level that does not exist reads a zero count and ends)
__global__ void k_lay_level(const tyr_bvh_node* __restrict__ nodes, uint32_t* __restrict__ list, uint32_t* __restrict__ seen, LayState* S, uint32_t level) {
	const uint32_t begin = S->levelOffset[level], count = S->levelCount[level];
	if (count == 0u)
		return;
	const uint32_t nextBegin = begin + count;
	if (blockIdx.x == 0 && threadIdx.x == 0)
		S->levelOffset[level + 1] = nextBegin;
	__shared__ uint32_t waveTotal[kB / 64];
	__shared__ uint32_t blockBase;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	for (uint32_t k0 = blockIdx.x * kB; k0 < count; k0 += gridDim.x * kB) {
		const uint32_t k = k0 + threadIdx.x;
		int32_t c0 = -1, c1 = -1;
		if (k < count) {
			const uint32_t node = list[begin + k];
			const tyr_bvh_node n = nodes[node];
			if (n.primitiveCount == 0) {
				c0 = (int32_t)node + 1;
				c1 = n.offset;
				const uint32_t mark = 1u | (((level + 1u) & 1u) << 1);
				if (atomicExch(&seen[c0], mark) != 0u || atomicExch(&seen[c1], mark) != 0u) { // somebody's child twice: not a tree
					S->bad = 1u;
					c0 = c1 = -1;
				}
			}
		}
		const unsigned long long m = __ballot(c0 >= 0);
		const uint32_t rank = 2u * (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
		if (lane == 0)
			waveTotal[wave] = 2u * (uint32_t)__popcll(m);
		__syncthreads();
		if (threadIdx.x == 0) {
			uint32_t tot = 0;
			for (int w = 0; w < kB / 64; ++w) {
				const uint32_t t = waveTotal[w];
				waveTotal[w] = tot;
				tot += t;
			}
			blockBase = tot ? atomicAdd(&S->levelCount[level + 1], tot) : 0u;
		}
		__syncthreads();
		if (c0 >= 0) {
			const uint32_t at = nextBegin + blockBase + waveTotal[wave] + rank;
			list[at] = (uint32_t)c0;
			list[at + 1] = (uint32_t)c1;
		}
		__syncthreads();
	}
}

// ---- prefix sum of "this node is the root of a quad record" in array order ----
__device__ __forceinline__ uint32_t quad_flag(const tyr_bvh_node* nodes, const uint32_t* seen, int i) { return (nodes[i].primitiveCount == 0 && (seen[i] & 2u) == 0u) ? 1u : 0u; }
__global__ void k_lay_scan_sums(const tyr_bvh_node* __restrict__ nodes, const uint32_t* __restrict__ seen, uint32_t* __restrict__ blockSums, int n) {
	__shared__ uint32_t part[kB / 64];
	uint32_t s = 0;
	for (uint32_t j = 0; j < kScanPer / kB; ++j) {
		const int i = (int)(blockIdx.x * kScanPer + j * kB + threadIdx.x);
		if (i < n)
			s += quad_flag(nodes, seen, i);
	}
	for (int o = 32; o > 0; o >>= 1)
		s += __shfl_down(s, o, 64);
	if ((threadIdx.x & 63u) == 0u)
		part[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t t = 0;
		for (int w = 0; w < kB / 64; ++w)
			t += part[w];
		blockSums[blockIdx.x] = t;
	}
}
__global__ void k_lay_scan_blocks(uint32_t* blockSums, int nBlocks, LayState* S) { // one block: exclusive scan in place, the total to S
	__shared__ uint32_t waveTot[kB / 64];
	__shared__ uint32_t carry;
	if (threadIdx.x == 0)
		carry = 0;
	__syncthreads();
	for (int b0 = 0; b0 < nBlocks; b0 += kB) {
		const int i = b0 + (int)threadIdx.x;
		const uint32_t v = i < nBlocks ? blockSums[i] : 0u;
		uint32_t x = v;
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t y = __shfl_up(x, o, 64);
			if ((int)(threadIdx.x & 63u) >= o)
				x += y;
		}
		if ((threadIdx.x & 63u) == 63u)
			waveTot[threadIdx.x >> 6] = x;
		__syncthreads();
		uint32_t before = carry;
		for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w)
			before += waveTot[w];
		if (i < nBlocks)
			blockSums[i] = before + x - v;
		__syncthreads();
		if (threadIdx.x == kB - 1)
			carry = before + x;
		__syncthreads();
	}
	if (threadIdx.x == 0)
		S->nRealQuads = carry;
}
// quadIndex[i] = the node's record (array order) or kNone; nodeOfQuad[] = the inverse
__global__ void k_lay_number(const tyr_bvh_node* __restrict__ nodes, const uint32_t* __restrict__ seen, const uint32_t* __restrict__ blockSums, uint32_t* __restrict__ quadIndex,
                             uint32_t* __restrict__ nodeOfQuad, int n) {
	__shared__ uint32_t waveTot[kB / 64];
	__shared__ uint32_t carry;
	if (threadIdx.x == 0)
		carry = blockSums[blockIdx.x];
	__syncthreads();
	for (uint32_t j = 0; j < kScanPer / kB; ++j) {
		const int i = (int)(blockIdx.x * kScanPer + j * kB + threadIdx.x);
		const uint32_t v = i < n ? quad_flag(nodes, seen, i) : 0u;
		uint32_t x = v;
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t y = __shfl_up(x, o, 64);
			if ((int)(threadIdx.x & 63u) >= o)
				x += y;
		}
		if ((threadIdx.x & 63u) == 63u)
			waveTot[threadIdx.x >> 6] = x;
		__syncthreads();
		uint32_t before = carry;
		for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w)
			before += waveTot[w];
		if (i < n) {
			const uint32_t q = before + x - v;
			quadIndex[i] = v ? q : kNone;
			if (v)
				nodeOfQuad[q] = (uint32_t)i;
		}
		__syncthreads();
		if (threadIdx.x == kB - 1)
			carry = before + x;
		__syncthreads();
	}
}

// ---- what a record holds (bvh_layout.cpp real_slots / real_meta): slots 0,1 = the children of the node's first child (or that
// child itself in slot 0 when it is a leaf), slots 2,3 the same for the second child ----
struct Slots {
	uint32_t ref[4]; // old quad index, leaf reference, or kRefDone (unused)
	int32_t node[4]; // the reference-tree node behind the slot (-1: none)
};
__device__ __forceinline__ Slots real_slots(const tyr_bvh_node* __restrict__ nodes, const uint32_t* __restrict__ quadIndex, int32_t pi) {
	Slots s;
	for (int k = 0; k < 4; ++k) {
		s.ref[k] = kRefDone;
		s.node[k] = -1;
	}
	const int32_t kids[2] = { pi + 1, nodes[pi].offset };
	for (int g = 0; g < 2; ++g) {
		const tyr_bvh_node X = nodes[kids[g]];
		if (X.primitiveCount > 0) {
			s.ref[2 * g] = leaf_ref_d((uint32_t)X.offset, X.primitiveCount);
			s.node[2 * g] = kids[g];
		} else {
			const int32_t gk[2] = { kids[g] + 1, X.offset };
			for (int h = 0; h < 2; ++h) {
				const tyr_bvh_node Y = nodes[gk[h]];
				s.ref[2 * g + h] = Y.primitiveCount > 0 ? leaf_ref_d((uint32_t)Y.offset, Y.primitiveCount) : quadIndex[gk[h]];
				s.node[2 * g + h] = gk[h];
			}
		}
	}
	return s;
}
__device__ __forceinline__ uint32_t real_meta(const tyr_bvh_node* __restrict__ nodes, int32_t pi) {
	const tyr_bvh_node P = nodes[pi];
	const tyr_bvh_node X0 = nodes[pi + 1], X1 = nodes[P.offset];
	return (uint32_t)P.splitAxis | ((X0.primitiveCount > 0 ? 0u : (uint32_t)X0.splitAxis) << 2) | ((X1.primitiveCount > 0 ? 0u : (uint32_t)X1.splitAxis) << 4);
}

// ---- the top of the tree: the first kStagedNodes records in breadth-first order.  One wave: a generation of the queue (the
// records appended by the generation before) is expanded by as many lanes at once, and the children are appended in lane order,
// slot order within a lane -- the order in which the serial walk (bvh_layout.cpp `topOld`) appends them; it stops at 64 alike.
// (one thread needed 0.3 ms for its ~450 dependent loads: half of the whole pass on C3) ----
__global__ void __launch_bounds__(64) k_lay_top(const tyr_bvh_node* __restrict__ nodes, const uint32_t* __restrict__ quadIndex, const uint32_t* __restrict__ nodeOfQuad, LayState* S) {
	static_assert(kStagedNodes == 64, "one lane per record of the top");
	__shared__ uint32_t topOld[kStagedNodes];
	__shared__ uint32_t so[kStagedNodes], sn[kStagedNodes];
	const uint32_t lane = threadIdx.x;
	if (lane == 0)
		topOld[0] = 0u; // the root is interior and at depth 0: record 0
	__syncthreads();
	uint32_t head = 0, n = 1;
	while (head < n && n < kStagedNodes) {
		const uint32_t gen = n - head;
		uint32_t kids[4], c = 0;
		if (lane < gen) {
			const Slots s = real_slots(nodes, quadIndex, (int32_t)nodeOfQuad[topOld[head + lane]]);
			for (int k = 0; k < 4; ++k)
				if ((int32_t)s.ref[k] >= 0)
					kids[c++] = s.ref[k];
		}
		uint32_t incl = c; // inclusive prefix sum over the lanes
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t y = __shfl_up(incl, o, 64);
			if ((int)lane >= o)
				incl += y;
		}
		const uint32_t total = __shfl(incl, 63, 64);
		const uint32_t at = n + incl - c;
		for (uint32_t k = 0; k < c; ++k)
			if (at + k < kStagedNodes)
				topOld[at + k] = kids[k];
		__syncthreads();
		head = n;
		n = min(n + total, kStagedNodes);
	}
	// (old index, new index) sorted by old index: lane i's record goes to place #{j : old_j < old_i} (the indices are distinct)
	const uint32_t mine = lane < n ? topOld[lane] : kNone;
	uint32_t place = 0;
	for (uint32_t j = 0; j < n; ++j)
		place += topOld[j] < mine ? 1u : 0u;
	if (lane < n) {
		so[place] = mine;
		sn[place] = lane;
	}
	__syncthreads();
	if (lane < n) {
		S->topSortedOld[lane] = so[lane];
		S->topSortedNew[lane] = sn[lane];
	}
	if (lane == 0) {
		S->nTop = n;
		S->quadRootRef = 0u | ((real_meta(nodes, 0) & 63u) << kQuadOrderShift); // new_index(0) = 0
	}
}
// old index -> new index: the top's records by the table, the others keep their order behind the top
__device__ __forceinline__ uint32_t new_index(const uint32_t* so, const uint32_t* sn, uint32_t nTop, uint32_t old) {
	uint32_t lo = 0, hi = nTop; // lower_bound
	while (lo < hi) {
		const uint32_t mid = (lo + hi) >> 1;
		if (so[mid] < old)
			lo = mid + 1;
		else
			hi = mid;
	}
	if (lo < nTop && so[lo] == old)
		return sn[lo];
	return nTop + old - lo;
}

// ---- the records, each written once at its final place (bvh_layout.cpp quad_write) ----
__global__ void k_lay_quads(const tyr_bvh_node* __restrict__ nodes, const uint32_t* __restrict__ quadIndex, float4* __restrict__ quads, int nNodes, const LayState* __restrict__ S) {
	__shared__ uint32_t so[kStagedNodes], sn[kStagedNodes];
	const uint32_t nTop = S->nTop;
	if (threadIdx.x < kStagedNodes) {
		so[threadIdx.x] = threadIdx.x < nTop ? S->topSortedOld[threadIdx.x] : kNone;
		sn[threadIdx.x] = threadIdx.x < nTop ? S->topSortedNew[threadIdx.x] : 0u;
	}
	__syncthreads();
	const int i = blockIdx.x * kB + threadIdx.x;
	if (i >= nNodes)
		return;
	const uint32_t qi = quadIndex[i];
	if (qi == kNone)
		return;
	const Slots s = real_slots(nodes, quadIndex, i);
	float lo[4][3], hi[4][3];
	uint32_t refs[4];
	const float inf = __builtin_inff();
	for (int k = 0; k < 4; ++k) {
		if (s.node[k] >= 0) {
			const tyr_bvh_node c = nodes[s.node[k]];
			for (int a = 0; a < 3; ++a) {
				lo[k][a] = c.bbox.bounds[0][a];
				hi[k][a] = c.bbox.bounds[1][a];
			}
		} else { // an unused slot: both planes of every axis at +infinity (no ray passes the box tests on it)
			for (int a = 0; a < 3; ++a)
				lo[k][a] = hi[k][a] = inf;
		}
		const uint32_t r = s.ref[k];
		refs[k] = (int32_t)r < 0 ? r : (new_index(so, sn, nTop, r) | ((real_meta(nodes, s.node[k]) & 63u) << kQuadOrderShift));
	}
	float4* q = quads + 8 * (size_t)new_index(so, sn, nTop, qi);
	for (int a = 0; a < 3; ++a) {
		q[2 * a + 0] = make_float4(lo[0][a], hi[0][a], lo[1][a], hi[1][a]);
		q[2 * a + 1] = make_float4(lo[2][a], hi[2][a], lo[3][a], hi[3][a]);
	}
	q[6] = make_float4(__uint_as_float(refs[0]), __uint_as_float(refs[1]), __uint_as_float(refs[2]), __uint_as_float(refs[3]));
	q[7] = make_float4(__uint_as_float(real_meta(nodes, i)), 0.0f, 0.0f, 0.0f);
}

// ---- the deepest a traversal's stack can get: need(q) = (used slots - 1) + max over the slots that are records of need(child),
// one level of the tree per launch, deepest first (bvh_layout.cpp "quadMaxStack") ----
__global__ void k_lay_need(const tyr_bvh_node* __restrict__ nodes, const uint32_t* __restrict__ quadIndex, const uint32_t* __restrict__ list, uint32_t* __restrict__ need, LayState* S, uint32_t level) {
	const uint32_t begin = S->levelOffset[level], count = S->levelCount[level];
	for (uint32_t k = blockIdx.x * kB + threadIdx.x; k < count; k += gridDim.x * kB) {
		const uint32_t node = list[begin + k];
		const uint32_t qi = quadIndex[node];
		if (qi == kNone)
			continue;
		const Slots s = real_slots(nodes, quadIndex, (int32_t)node);
		uint32_t used = 0, deepest = 0;
		for (int j = 0; j < 4; ++j) {
			if (s.ref[j] == kRefDone)
				continue;
			++used;
			if ((int32_t)s.ref[j] >= 0)
				deepest = max(deepest, need[s.ref[j]]);
		}
		const uint32_t v = (used ? used - 1u : 0u) + deepest;
		need[qi] = v;
		if (level == 0u)
			S->maxStack = v;
	}
}

template <class T>
struct DevBufL {
	T* p = nullptr;
	hipError_t alloc(size_t count) { return hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T)); }
	T* release() {
		T* r = p;
		p = nullptr;
		return r;
	}
	~DevBufL() {
		if (p)
			(void)hipFree(p);
	}
};
inline unsigned blocks_for(size_t n) { return static_cast<unsigned>((n + kB - 1) / kB); }

} // namespace

// dNodes / dPrims: DEVICE arrays in the reference's formats.  On TYR_OK out.quads / out.tris are fresh device allocations (hipMalloc)
// the caller owns.  TYR_ERR_UNSUPPORTED: the host pass has to do this tree (see the head of this file); anything else: a device error.
int layout_on_device(const tyr_bvh_node* dNodes, int32_t nNodes, const tyr_triangle* dPrims, int32_t nPrims, DeviceTreeLayout& out, hipStream_t st) {
#define TYR_L(expr)                                                     \
	do {                                                                \
		const hipError_t e_ = (expr);                                   \
		if (e_ != hipSuccess)                                           \
			return e_ == hipErrorOutOfMemory ? TYR_ERR_OOM : static_cast<int>(e_); \
	} while (0)
	out = DeviceTreeLayout{};
	if (nNodes < 3 || nPrims <= 0 || !dNodes || !dPrims || static_cast<uint32_t>(nPrims) > kMaxPrimOffset)
		return TYR_ERR_UNSUPPORTED;
	const size_t nN = static_cast<size_t>(nNodes), nP = static_cast<size_t>(nPrims);
	const size_t nScanBlocks = (nN + kScanPer - 1) / kScanPer;
	DevBufL<LayState> dS;
	DevBufL<uint32_t> dSeen, dList, dQuadIndex, dNodeOfQuad, dNeed, dBlockSums;
	DevBufL<float4> dTris, dQuads;
	TYR_L(dS.alloc(1));
	TYR_L(dSeen.alloc(nN));
	TYR_L(dList.alloc(nN));
	TYR_L(dQuadIndex.alloc(nN));
	TYR_L(dBlockSums.alloc(nScanBlocks));
	TYR_L(dTris.alloc(3 * nP));
	TYR_L(hipMemsetAsync(dS.p, 0, sizeof(LayState), st));
	hipLaunchKernelGGL(k_lay_tris, dim3(blocks_for(nP)), dim3(kB), 0, st, dPrims, dTris.p, nPrims, dS.p);
	hipLaunchKernelGGL(k_lay_validate, dim3(blocks_for(nN)), dim3(kB), 0, st, dNodes, dSeen.p, nNodes, nPrims, dS.p);
	LayState hS; // (a few KB: read back whole)
	TYR_L(hipMemcpyAsync(&hS, dS.p, offsetof(LayState, topSortedOld), hipMemcpyDeviceToHost, st));
	TYR_L(hipStreamSynchronize(st));
	if (hS.bad || hS.notHere)
		return TYR_ERR_UNSUPPORTED;
	// ---- the levels: the root is level 0 ----
	const uint32_t one = 1u, zero = 0u, rootMark = 1u; // (sources of asynchronous copies: they live until the stream has been waited for)
	TYR_L(hipMemcpyAsync(&dS.p->levelCount[0], &one, 4, hipMemcpyHostToDevice, st));
	TYR_L(hipMemcpyAsync(dList.p, &zero, 4, hipMemcpyHostToDevice, st));
	TYR_L(hipMemcpyAsync(dSeen.p, &rootMark, 4, hipMemcpyHostToDevice, st));
	const unsigned levelGrid = std::min<unsigned>(blocks_for(nN / 2 + 1), 2048u);
	uint32_t nLevels = 0;
	for (uint32_t l0 = 0;; l0 += 16) {
		if (l0 + 16 > kMaxLevels)
			return TYR_ERR_UNSUPPORTED;
		for (uint32_t l = l0; l < l0 + 16; ++l)
			hipLaunchKernelGGL(k_lay_level, dim3(levelGrid), dim3(kB), 0, st, dNodes, dList.p, dSeen.p, dS.p, l);
		uint32_t counts[17];
		TYR_L(hipMemcpyAsync(counts, &dS.p->levelCount[l0], sizeof counts, hipMemcpyDeviceToHost, st));
		TYR_L(hipStreamSynchronize(st));
		uint32_t k = 0;
		while (k < 17 && counts[k] != 0u)
			++k;
		nLevels = l0 + k;
		if (k < 17)
			break;
	}
	TYR_L(hipMemcpyAsync(&hS, dS.p, sizeof(LayState), hipMemcpyDeviceToHost, st));
	TYR_L(hipStreamSynchronize(st));
	if (hS.bad || nLevels == 0 || static_cast<size_t>(hS.levelOffset[nLevels - 1]) + hS.levelCount[nLevels - 1] != nN) // a child named twice, or nodes nobody names
		return TYR_ERR_UNSUPPORTED;
	// ---- numbering ----
	hipLaunchKernelGGL(k_lay_scan_sums, dim3(static_cast<unsigned>(nScanBlocks)), dim3(kB), 0, st, dNodes, dSeen.p, dBlockSums.p, nNodes);
	hipLaunchKernelGGL(k_lay_scan_blocks, dim3(1), dim3(kB), 0, st, dBlockSums.p, static_cast<int>(nScanBlocks), dS.p);
	uint32_t nQuads = 0;
	TYR_L(hipMemcpyAsync(&nQuads, &dS.p->nRealQuads, 4, hipMemcpyDeviceToHost, st));
	TYR_L(hipStreamSynchronize(st));
	if (nQuads == 0 || nQuads > kQuadIndexMask)
		return TYR_ERR_UNSUPPORTED;
	TYR_L(dNodeOfQuad.alloc(nQuads));
	TYR_L(dNeed.alloc(nQuads));
	TYR_L(dQuads.alloc(8 * static_cast<size_t>(nQuads)));
	hipLaunchKernelGGL(k_lay_number, dim3(static_cast<unsigned>(nScanBlocks)), dim3(kB), 0, st, dNodes, dSeen.p, dBlockSums.p, dQuadIndex.p, dNodeOfQuad.p, nNodes);
	hipLaunchKernelGGL(k_lay_top, dim3(1), dim3(64), 0, st, dNodes, dQuadIndex.p, dNodeOfQuad.p, dS.p);
	hipLaunchKernelGGL(k_lay_quads, dim3(blocks_for(nN)), dim3(kB), 0, st, dNodes, dQuadIndex.p, dQuads.p, nNodes, dS.p);
	for (uint32_t l = nLevels; l-- > 0;) {
		if (l & 1u)
			continue; // records sit at even depth
		const unsigned g = std::max(1u, std::min<unsigned>(blocks_for(hS.levelCount[l]), 2048u));
		hipLaunchKernelGGL(k_lay_need, dim3(g), dim3(kB), 0, st, dNodes, dQuadIndex.p, dList.p, dNeed.p, dS.p, l);
	}
	TYR_L(hipMemcpyAsync(&hS, dS.p, offsetof(LayState, levelOffset), hipMemcpyDeviceToHost, st));
	TYR_L(hipStreamSynchronize(st));
	TYR_L(hipGetLastError());
	out.nQuads = nQuads;
	out.nStaged = hS.nTop;
	out.quadMaxStack = hS.maxStack;
	out.quadRootRef = hS.quadRootRef;
	std::memcpy(out.rootMin, hS.rootMin, 12);
	std::memcpy(out.rootMax, hS.rootMax, 12);
	out.quads = dQuads.release();
	out.tris = dTris.release();
	return TYR_OK;
#undef TYR_L
}

} // namespace tyr
