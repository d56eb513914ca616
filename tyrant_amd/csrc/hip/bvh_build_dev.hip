// bvh_build_dev.hip -- class BVH's binned-SAH build (bvh.h:49-108, bvh.cpp:3-225) ON THE DEVICE, emitting the reference's bytes.
//
// SURVEY.md 8f-1 names two ways out of the reference's serial builder (5.6 s per 1 M triangles in the survey's container): a
// parallel host build -- host/bvh_build.cpp, 0.11 s per 1 M on 16 threads -- and a device build; this is the second (round 5).
// What has to be reproduced is not "a good tree" but THE tree: every node's 32 bytes and the order of the primitives, because
// the traversal's visit order -- and through the epsilon of bvh.h:134 every hit -- follows from them.  Three facts make that
// possible on a GPU:
//   * everything bvh.cpp decides for a range [start, end) of its primitive-info array depends only on that range's contents:
//     bounds and centroid bounds are min / max (exact, order-free), the 14 bucket counts and boxes likewise, and the one
//     order-DEPENDENT piece of arithmetic -- the SAH costs (bvh.cpp:132-160) -- is 13 candidates over 14 buckets, done by one
//     thread in the reference's order;
//   * `std::partition` as libstdc++'s bidirectional form (bvh.cpp:171-178; host/bvh_build.cpp restates it) permutes
//     deterministically: the k-th misplaced element of the left part, counted from the left, changes places with the k-th
//     misplaced element of the right part, counted from the RIGHT -- two prefix sums and a table lookup;
//   * the array is depth-first with the first child at index + 1 (bvh.cpp:195-202), so a node's index is its parent's + 1
//     (first child) or + 1 + the first child's subtree size (second child), and the primitive order IS the final order of the
//     info array.
// The shape is the host builder's: the top of the tree level by level (here: every range of a level at once, all primitives
// in flight), subtrees of at most kTaskPrims primitives built serially -- here by one THREAD each, running the reference's
// recursion as an explicit stack, into local node arrays --, then sizes bottom-up, bases top-down, and the copy-out.
// Signed zeros: min / max by parallel reduction may pick -0 where the serial fold keeps +0.  Leaf boxes (the only boxes that
// are stored from a reduction-free fold) are computed by the task threads in the reference's order, interior boxes are
// Union(left, right) of stored boxes as in initInterior (bvh.cpp:220-225), and a range whose box has zero surface area --
// where the sign of a zero could reach a decision through the division at bvh.cpp:150 -- is handed to a task thread whole.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../../include/tyr_c.h"
#include "../host/device_build.hpp"

namespace tyr {

namespace {

constexpr int kBuckets = 14;             // bvh.h:76
constexpr int kMaxLeafPrims = 4;         // bvh.h:78
constexpr float kTraversalCost = 1.0f;   // bvh.h:81
constexpr float kIntersectionCost = 1.0f; // bvh.h:84
constexpr int kTaskPrimsDefault = 32;    // ranges of at most this many primitives are built by one thread (C3 on an MI355X: 16 / 32 / 64 / 128 / 256 -> 7.4 / 7.0 / 7.6 / 9.8 / 18.3 ms; TYR_DEVBUILD_TASK_PRIMS overrides)
constexpr int kTaskStack = 96;           // explicit recursion stack of a task thread (a subtree of kTaskPrims primitives is at most that deep; larger -- degenerate -- tasks may overflow: reported, the host falls back)
constexpr int kBlockB = 256;

struct Info { // bvh.h:88-97 BVHPrimitiveInfo
	float lo[3], hi[3], c[3];
	uint32_t idx;
};
static_assert(sizeof(Info) == 40, "Info");

// fmin / fmax as glibc evaluates them for non-NaN inputs: the first argument wins ties (host/bvh_build.cpp)
__device__ __forceinline__ float fmin_first(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float fmax_first(float a, float b) { return (b > a) ? b : a; }
struct BoxD {
	float lo[3], hi[3];
	__device__ void init() { // Bbox.h:5
		for (int k = 0; k < 3; ++k) {
			lo[k] = 1e10f;
			hi[k] = -1e10f;
		}
	}
	__device__ void add(const float* v) {
		for (int k = 0; k < 3; ++k) {
			lo[k] = fmin_first(lo[k], v[k]);
			hi[k] = fmax_first(hi[k], v[k]);
		}
	}
	__device__ void unite(const float* l, const float* h) { // Union(this, other), Bbox.cpp:3-14
		for (int k = 0; k < 3; ++k) {
			lo[k] = fmin_first(lo[k], l[k]);
			hi[k] = fmax_first(hi[k], h[k]);
		}
	}
	__device__ float surfaceArea() const { // Bbox.h:18-21
		const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
		return 2 * (dx * dy + dx * dz + dy * dz);
	}
	__device__ int largestExtent() const { // Bbox.h:28-36
		const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
		if (dx > dy && dx > dz)
			return 0;
		return (dy > dz) ? 1 : 2;
	}
};
// bvh.cpp:44-58
__device__ __forceinline__ int bucket_of(float centroid, float cb, float ct) {
	float distance = centroid - cb;
	if (ct > cb)
		distance = distance / (ct - cb);
	int b = static_cast<int>(kBuckets * distance);
	if (b == kBuckets)
		--b;
	return b;
}
// the SAH of bvh.cpp:124-168 over filled buckets: the bucket to split behind, or -1 for "make a leaf"
__device__ int sah_split(const int* count, const BoxD* bounds, const BoxD& nodeBox, int n) {
	BoxD sufBox[kBuckets];
	int sufCount[kBuckets];
	{
		BoxD acc;
		acc.init();
		int c = 0;
		for (int b = kBuckets - 1; b >= 1; --b) {
			// acc = Union(bounds[b], acc)
			BoxD t = bounds[b];
			t.unite(acc.lo, acc.hi);
			acc = t;
			c += count[b];
			sufBox[b - 1] = acc;
			sufCount[b - 1] = c;
		}
	}
	const float nodeSA = nodeBox.surfaceArea();
	float minCost = FLT_MAX;
	int minBucket = -1;
	BoxD pre;
	pre.init();
	int preCount = 0;
	for (int c = 0; c < kBuckets - 1; ++c) {
		pre.unite(bounds[c].lo, bounds[c].hi);
		preCount += count[c];
		const float cost = kTraversalCost + (static_cast<float>(preCount) * pre.surfaceArea() + static_cast<float>(sufCount[c]) * sufBox[c].surfaceArea()) / nodeSA;
		if (cost < minCost) {
			minCost = cost;
			minBucket = c;
		}
	}
	const float leafCost = kIntersectionCost * static_cast<float>(n);
	if (minBucket < 0)
		return -1;
	if (!(n > kMaxLeafPrims || minCost < leafCost))
		return -1;
	return minBucket;
}

// ---- floats as order-preserving unsigned integers (atomicMin / atomicMax) ----
__device__ __forceinline__ uint32_t f2o(float f) {
	const uint32_t u = __float_as_uint(f);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float o2f(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

// ---- the top of the tree: one record per node that the level loop or a task made ----
enum : int { kActive = 0, kTask = 2, kInterior = 3 };
struct TopNode {
	int start, end;
	int state;
	int dim;
	int left, right; // node ids (interior)
	int size;        // nodes in the subtree (tasks: their local count)
	int base;        // index of this node in the reference's array
	float lo[3], hi[3]; // the node's stored box (tasks: their local root's)
};
struct Slot { // an active range of the current level
	int node;
	int dim;        // -1: no split at this level
	int minBucket, mid;
	int childSlot[2]; // the children's slots in the NEXT level's list (-1: a task)
	float cb, ct;
	uint32_t nodeLo[3], nodeHi[3], cLo[3], cHi[3]; // ordered-uint accumulators
	uint32_t count[kBuckets];
	uint32_t bLo[kBuckets][3], bHi[kBuckets][3];   // the buckets' boxes (bvh.cpp:124-131) ...
	uint32_t bcLo[kBuckets][3], bcHi[kBuckets][3]; // ... and the bounds of their centroids: a child's node box and centroid box are unions over its buckets, so only the root asks every primitive for them (k_bounds)
};
struct Counters {
	int nNodes;   // top nodes allocated
	int nNext;    // slots of the next level
	int error;    // 1: a task thread's stack overflowed; 2: a primitive box that is not finite (TYR_ERR_INVALID)
};

__global__ void k_init_info(const tyr_bbox* __restrict__ bb, Info* __restrict__ info, int* __restrict__ slotOf, int n, int rootSlot, Counters* K) {
	const int i = blockIdx.x * kBlockB + threadIdx.x;
	if (i >= n)
		return;
	Info p;
	bool finite = true;
	for (int k = 0; k < 3; ++k) {
		p.lo[k] = bb[i].bounds[0][k];
		p.hi[k] = bb[i].bounds[1][k];
		finite = finite && isfinite(p.lo[k]) && isfinite(p.hi[k]);
	}
	if (!finite) { // refused -- and made harmless until the host has read the flag (a bucket index computed from a NaN is anything)
		K->error = 2;
		for (int k = 0; k < 3; ++k)
			p.lo[k] = p.hi[k] = 0.0f;
	}
	for (int k = 0; k < 3; ++k)
		p.c[k] = p.lo[k] * 0.5f + p.hi[k] * 0.5f; // bvh.h:96
	p.idx = static_cast<uint32_t>(i);
	info[i] = p;
	slotOf[i] = rootSlot;
}

__global__ void k_slot_reset(Slot* slots, int nSlots, int resetBounds) {
	const int s = blockIdx.x * kBlockB + threadIdx.x;
	if (s >= nSlots)
		return;
	Slot& S = slots[s];
	if (resetBounds)
		for (int k = 0; k < 3; ++k) {
			S.nodeLo[k] = S.cLo[k] = 0xffffffffu;
			S.nodeHi[k] = S.cHi[k] = 0u;
		}
	for (int b = 0; b < kBuckets; ++b) {
		S.count[b] = 0;
		for (int k = 0; k < 3; ++k) {
			S.bLo[b][k] = S.bcLo[b][k] = 0xffffffffu;
			S.bHi[b][k] = S.bcHi[b][k] = 0u;
		}
	}
}

// node box and centroid box of the ROOT range (every other range inherits them from its parent's buckets): wave reduction, then
// the block's four waves through LDS, twelve atomics per block (one word serves only ~88 atomics per microsecond: per wave
// they were 1.4 ms of a 1 M-primitive build)
__global__ void k_bounds(const Info* __restrict__ info, const int* __restrict__ slotOf, Slot* slots, int n) {
	__shared__ uint32_t sh[kBlockB / 64][12];
	const int i = blockIdx.x * kBlockB + threadIdx.x;
	const int s = i < n ? slotOf[i] : -1;
	uint32_t v[12];
	for (int k = 0; k < 3; ++k) { // neutral elements for lanes without a primitive
		v[k] = v[6 + k] = 0xffffffffu;
		v[3 + k] = v[9 + k] = 0u;
	}
	if (s >= 0) {
		const Info p = info[i];
		for (int k = 0; k < 3; ++k) {
			v[k] = f2o(p.lo[k]);
			v[3 + k] = f2o(p.hi[k]);
			v[6 + k] = f2o(p.c[k]);
			v[9 + k] = v[6 + k];
		}
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
		for (int k = 0; k < 12; ++k) {
			const uint32_t w = __shfl_xor(v[k], o, 64);
			const bool isMin = k < 3 || (k >= 6 && k < 9);
			v[k] = isMin ? (w < v[k] ? w : v[k]) : (w > v[k] ? w : v[k]);
		}
	if ((threadIdx.x & 63) == 0)
		for (int k = 0; k < 12; ++k)
			sh[threadIdx.x >> 6][k] = v[k];
	__syncthreads();
	if (threadIdx.x < 12) {
		const int k = threadIdx.x;
		const bool isMin = k < 3 || (k >= 6 && k < 9);
		uint32_t r = sh[0][k];
		for (int w = 1; w < kBlockB / 64; ++w)
			r = isMin ? (sh[w][k] < r ? sh[w][k] : r) : (sh[w][k] > r ? sh[w][k] : r);
		Slot& S = slots[0]; // the root's slot
		uint32_t* dst = k < 3 ? &S.nodeLo[k] : k < 6 ? &S.nodeHi[k - 3] : k < 9 ? &S.cLo[k - 6] : &S.cHi[k - 9];
		if (isMin)
			atomicMin(dst, r);
		else
			atomicMax(dst, r);
	}
}

// bvh.cpp:86-111: the split dimension, or "no split at this level" (identical centroids: a leaf -- built, like everything that
// is not split here, by a task thread in the reference's own order)
__global__ void k_decide_dim(Slot* slots, TopNode* nodes, int nSlots) {
	const int s = blockIdx.x * kBlockB + threadIdx.x;
	if (s >= nSlots)
		return;
	Slot& S = slots[s];
	BoxD cbox;
	for (int k = 0; k < 3; ++k) {
		cbox.lo[k] = o2f(S.cLo[k]);
		cbox.hi[k] = o2f(S.cHi[k]);
	}
	const int dim = cbox.largestExtent();
	S.cb = cbox.lo[dim];
	S.ct = cbox.hi[dim];
	S.dim = dim;
	S.childSlot[0] = S.childSlot[1] = -1;
	BoxD nb;
	for (int k = 0; k < 3; ++k) {
		nb.lo[k] = o2f(S.nodeLo[k]);
		nb.hi[k] = o2f(S.nodeHi[k]);
	}
	if (S.cb == S.ct || nb.surfaceArea() == 0.0f) { // (zero area: the sign of a zero could reach bvh.cpp:150's division -- serial, exact)
		S.dim = -1;
		nodes[S.node].state = kTask;
	}
}

// bucket of every primitive of a range that is being split, and the buckets' counts, boxes and centroid bounds (bvh.cpp:124-131).
// Accumulated in LDS per BLOCK while the whole block sits in one range (the top levels: 98 words per range would otherwise take
// a million atomics each), per WAVE while a wave does (ranges are contiguous and longer than a wave: most waves of the middle
// levels), lane by lane only in the waves that straddle a boundary.
constexpr int kBktWords = 13; // count, box lo / hi, centroid lo / hi
__device__ __forceinline__ void bucket_add(uint32_t* w, const Info& p) { // w: the bucket's 13 words (LDS or global)
	atomicAdd(&w[0], 1u);
	for (int k = 0; k < 3; ++k) {
		atomicMin(&w[1 + k], f2o(p.lo[k]));
		atomicMax(&w[4 + k], f2o(p.hi[k]));
		const uint32_t c = f2o(p.c[k]);
		atomicMin(&w[7 + k], c);
		atomicMax(&w[10 + k], c);
	}
}
__device__ __forceinline__ void bucket_flush(Slot& S, int b, const uint32_t* w) {
	atomicAdd(&S.count[b], w[0]);
	for (int k = 0; k < 3; ++k) {
		atomicMin(&S.bLo[b][k], w[1 + k]);
		atomicMax(&S.bHi[b][k], w[4 + k]);
		atomicMin(&S.bcLo[b][k], w[7 + k]);
		atomicMax(&S.bcHi[b][k], w[10 + k]);
	}
}
__global__ void k_buckets(const Info* __restrict__ info, const int* __restrict__ slotOf, Slot* slots, uint8_t* __restrict__ bkt, int n) {
	__shared__ uint32_t shB[kBuckets][kBktWords];                  // the block's
	__shared__ uint32_t shW[kBlockB / 64][kBuckets][kBktWords];    // each wave's
	__shared__ int shSlot, shUniform;
	const int i = blockIdx.x * kBlockB + threadIdx.x;
	const int s = i < n ? slotOf[i] : -1;
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	if (threadIdx.x == 0) {
		shSlot = s;
		shUniform = 1;
	}
	for (int j = threadIdx.x; j < kBuckets * kBktWords; j += kBlockB) {
		const int w = j % kBktWords;
		const uint32_t init = w == 0 ? 0u : ((w >= 1 && w <= 3) || (w >= 7 && w <= 9)) ? 0xffffffffu : 0u;
		(&shB[0][0])[j] = init;
		for (int v = 0; v < kBlockB / 64; ++v)
			(&shW[v][0][0])[j] = init;
	}
	__syncthreads();
	if (s != shSlot)
		shUniform = 0; // (benign race: every writer writes 0)
	__syncthreads();
	const bool blockUniform = shUniform != 0;
	const int s0 = __shfl(s, 0, 64);
	const bool waveUniform = __all(s == s0) != 0;
	int b = 0;
	bool live = false;
	Info p;
	if (s >= 0 && slots[s].dim >= 0) {
		const Slot& S = slots[s];
		p = info[i];
		b = bucket_of(p.c[S.dim], S.cb, S.ct);
		bkt[i] = static_cast<uint8_t>(b);
		live = true;
	}
	if (blockUniform) {
		if (live)
			bucket_add(shB[b], p);
		__syncthreads();
		if (shSlot >= 0 && threadIdx.x < kBuckets && shB[threadIdx.x][0] != 0)
			bucket_flush(slots[shSlot], threadIdx.x, shB[threadIdx.x]);
	} else if (waveUniform) {
		if (live)
			bucket_add(shW[wave][b], p);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		if (s0 >= 0 && lane < kBuckets && shW[wave][lane][0] != 0)
			bucket_flush(slots[s0], lane, shW[wave][lane]);
	} else if (live) {
		Slot& S = slots[s];
		uint32_t one[kBktWords];
		one[0] = 1u;
		for (int k = 0; k < 3; ++k) {
			one[1 + k] = f2o(p.lo[k]);
			one[4 + k] = f2o(p.hi[k]);
			one[7 + k] = one[10 + k] = f2o(p.c[k]);
		}
		bucket_flush(S, b, one);
	}
}

// bvh.cpp:132-193: the SAH, leaf or split, the children
__global__ void k_decide_split(Slot* slots, Slot* next, TopNode* nodes, Counters* K, int nSlots, int kTaskPrims) {
	const int s = blockIdx.x * kBlockB + threadIdx.x;
	if (s >= nSlots)
		return;
	Slot& S = slots[s];
	if (S.dim < 0)
		return;
	TopNode& N = nodes[S.node];
	const int n = N.end - N.start;
	int count[kBuckets];
	BoxD bounds[kBuckets];
	for (int b = 0; b < kBuckets; ++b) {
		count[b] = static_cast<int>(S.count[b]);
		if (count[b] == 0) {
			bounds[b].init(); // an empty bucket's box is the reference's initial box (Bbox.h:5)
		} else {
			for (int k = 0; k < 3; ++k) {
				bounds[b].lo[k] = o2f(S.bLo[b][k]);
				bounds[b].hi[k] = o2f(S.bHi[b][k]);
			}
		}
	}
	BoxD nb;
	for (int k = 0; k < 3; ++k) {
		nb.lo[k] = o2f(S.nodeLo[k]);
		nb.hi[k] = o2f(S.nodeHi[k]);
	}
	const int minBucket = sah_split(count, bounds, nb, n);
	if (minBucket < 0) {
		S.dim = -1;
		N.state = kTask; // a leaf: emitted by a task thread (the same decision, in the reference's own order)
		return;
	}
	int left = 0;
	for (int b = 0; b <= minBucket; ++b)
		left += count[b];
	S.minBucket = minBucket;
	S.mid = N.start + left;
	N.state = kInterior;
	N.dim = S.dim;
	const int first = atomicAdd(&K->nNodes, 2);
	N.left = first;
	N.right = first + 1;
	for (int c = 0; c < 2; ++c) {
		TopNode& C = nodes[first + c];
		C.start = c == 0 ? N.start : S.mid;
		C.end = c == 0 ? S.mid : N.end;
		C.left = C.right = -1;
		C.dim = 0;
		C.size = 0;
		C.base = 0;
		if (C.end - C.start <= kTaskPrims) {
			C.state = kTask;
			S.childSlot[c] = -1;
		} else {
			C.state = kActive;
			const int ns = atomicAdd(&K->nNext, 1);
			Slot& Nx = next[ns];
			Nx.node = first + c;
			// the child's node box and centroid box: the unions over its buckets (min / max: the same values a pass over its
			// primitives would find)
			const int b0 = c == 0 ? 0 : minBucket + 1, b1 = c == 0 ? minBucket : kBuckets - 1;
			for (int k = 0; k < 3; ++k) {
				uint32_t nl = 0xffffffffu, nh = 0u, cl = 0xffffffffu, ch = 0u;
				for (int b = b0; b <= b1; ++b) {
					if (S.count[b] == 0)
						continue;
					nl = S.bLo[b][k] < nl ? S.bLo[b][k] : nl;
					nh = S.bHi[b][k] > nh ? S.bHi[b][k] : nh;
					cl = S.bcLo[b][k] < cl ? S.bcLo[b][k] : cl;
					ch = S.bcHi[b][k] > ch ? S.bcHi[b][k] : ch;
				}
				Nx.nodeLo[k] = nl, Nx.nodeHi[k] = nh, Nx.cLo[k] = cl, Nx.cHi[k] = ch;
			}
			S.childSlot[c] = ns;
		}
	}
}

// std::partition(begin, end, bucket <= minBucket) (bvh.cpp:171-178), step 1: who is on the wrong side
__global__ void k_flags(const int* __restrict__ slotOf, const Slot* __restrict__ slots, const uint8_t* __restrict__ bkt, uint2* __restrict__ flags, int n) {
	const int i = blockIdx.x * kBlockB + threadIdx.x;
	if (i >= n)
		return;
	uint2 f = make_uint2(0u, 0u);
	const int s = slotOf[i];
	if (s >= 0 && slots[s].dim >= 0) {
		const Slot& S = slots[s];
		const bool pred = static_cast<int>(bkt[i]) <= S.minBucket;
		if (i < S.mid)
			f.x = pred ? 0u : 1u; // misplaced in the left part
		else
			f.y = pred ? 1u : 0u; // misplaced in the right part
	}
	flags[i] = f;
}

// ---- exclusive prefix sums of the two flags over the whole array (n + 1 entries) ----
constexpr int kScanPerBlock = 1024;
__global__ void k_scan_sums(const uint2* __restrict__ flags, uint2* __restrict__ blockSums, int n) {
	__shared__ uint2 sh[kBlockB];
	uint2 acc = make_uint2(0u, 0u);
	const int base = blockIdx.x * kScanPerBlock;
	for (int j = threadIdx.x; j < kScanPerBlock; j += kBlockB)
		if (base + j < n) {
			acc.x += flags[base + j].x;
			acc.y += flags[base + j].y;
		}
	sh[threadIdx.x] = acc;
	__syncthreads();
	for (int o = kBlockB / 2; o > 0; o >>= 1) {
		if (threadIdx.x < static_cast<unsigned>(o)) {
			sh[threadIdx.x].x += sh[threadIdx.x + o].x;
			sh[threadIdx.x].y += sh[threadIdx.x + o].y;
		}
		__syncthreads();
	}
	if (threadIdx.x == 0)
		blockSums[blockIdx.x] = sh[0];
}
__global__ void k_scan_blocks(uint2* blockSums, int nBlocks) { // one block: exclusive scan of the block sums, in place
	__shared__ uint2 sh[kBlockB];
	__shared__ uint2 carry;
	if (threadIdx.x == 0)
		carry = make_uint2(0u, 0u);
	__syncthreads();
	for (int base = 0; base < nBlocks; base += kBlockB) {
		const int i = base + threadIdx.x;
		const uint2 v = i < nBlocks ? blockSums[i] : make_uint2(0u, 0u);
		sh[threadIdx.x] = v;
		__syncthreads();
		for (int o = 1; o < kBlockB; o <<= 1) {
			uint2 t = make_uint2(0u, 0u);
			if (threadIdx.x >= static_cast<unsigned>(o))
				t = sh[threadIdx.x - o];
			__syncthreads();
			sh[threadIdx.x].x += t.x;
			sh[threadIdx.x].y += t.y;
			__syncthreads();
		}
		if (i < nBlocks)
			blockSums[i] = make_uint2(carry.x + sh[threadIdx.x].x - v.x, carry.y + sh[threadIdx.x].y - v.y);
		__syncthreads();
		if (threadIdx.x == 0) {
			carry.x += sh[kBlockB - 1].x;
			carry.y += sh[kBlockB - 1].y;
		}
		__syncthreads();
	}
}
__global__ void k_scan_final(const uint2* __restrict__ flags, const uint2* __restrict__ blockSums, uint2* __restrict__ excl, int n) {
	__shared__ uint2 sh[kBlockB];
	const int base = blockIdx.x * kScanPerBlock + threadIdx.x * 4;
	uint2 v[4];
	uint2 mine = make_uint2(0u, 0u);
	for (int j = 0; j < 4; ++j) {
		v[j] = base + j < n ? flags[base + j] : make_uint2(0u, 0u);
		mine.x += v[j].x;
		mine.y += v[j].y;
	}
	sh[threadIdx.x] = mine;
	__syncthreads();
	for (int o = 1; o < kBlockB; o <<= 1) {
		uint2 t = make_uint2(0u, 0u);
		if (threadIdx.x >= static_cast<unsigned>(o))
			t = sh[threadIdx.x - o];
		__syncthreads();
		sh[threadIdx.x].x += t.x;
		sh[threadIdx.x].y += t.y;
		__syncthreads();
	}
	uint2 run = make_uint2(blockSums[blockIdx.x].x + sh[threadIdx.x].x - mine.x, blockSums[blockIdx.x].y + sh[threadIdx.x].y - mine.y);
	for (int j = 0; j < 4; ++j) {
		if (base + j <= n)
			excl[base + j] = run;
		run.x += v[j].x;
		run.y += v[j].y;
	}
}

// step 2: the positions of the misplaced elements, in array order, per side
__global__ void k_lists(const uint2* __restrict__ flags, const uint2* __restrict__ excl, uint32_t* __restrict__ posML, uint32_t* __restrict__ posMR, int n) {
	const int i = blockIdx.x * kBlockB + threadIdx.x;
	if (i >= n)
		return;
	const uint2 f = flags[i];
	if (f.x)
		posML[excl[i].x] = static_cast<uint32_t>(i);
	if (f.y)
		posMR[excl[i].y] = static_cast<uint32_t>(i);
}

// step 3: the k-th misplaced element of the left part (from the left) and the k-th of the right part (from the RIGHT) change
// places; everything else stays.  Every record moves to the other buffer with the slot its new place has on the next level.
__global__ void k_scatter(const Info* __restrict__ info, Info* __restrict__ out, const int* __restrict__ slotOf, int* __restrict__ slotOut, const Slot* __restrict__ slots, const TopNode* __restrict__ nodes,
                          const uint2* __restrict__ flags, const uint2* __restrict__ excl, const uint32_t* __restrict__ posML, const uint32_t* __restrict__ posMR, int n) {
	const int i = blockIdx.x * kBlockB + threadIdx.x;
	if (i >= n)
		return;
	const int s = slotOf[i];
	int dst = i, ns = -1;
	if (s >= 0 && slots[s].dim >= 0) {
		const Slot& S = slots[s];
		const TopNode& N = nodes[S.node];
		const uint2 f = flags[i];
		if (f.x | f.y) {
			const uint2 e0 = excl[N.start], e1 = excl[N.end], e = excl[i];
			const uint32_t total = e1.x - e0.x; // = e1.y - e0.y: as many misplaced on one side as on the other
			if (f.x) {
				const uint32_t k = e.x - e0.x;              // k-th from the left
				dst = static_cast<int>(posMR[e0.y + (total - 1u - k)]); // k-th from the right
			} else {
				const uint32_t kr = total - 1u - (e.y - e0.y); // k-th from the right
				dst = static_cast<int>(posML[e0.x + kr]);
			}
		}
		ns = S.childSlot[dst < S.mid ? 0 : 1];
	}
	out[dst] = info[i];
	slotOut[dst] = ns;
}

// ---- subtrees of at most kTaskPrims primitives (and everything the level loop did not split): one thread each, the reference's
// recursion as an explicit stack, local node arrays (node 0 = the subtree's root, interior offsets and leaf offsets local) ----
__global__ void k_tasks(Info* info, TopNode* nodes, tyr_bvh_node* scratch, Counters* K, int nTop) {
	const int t = blockIdx.x * 64 + threadIdx.x;
	if (t >= nTop || nodes[t].state != kTask)
		return;
	TopNode& T = nodes[t];
	tyr_bvh_node* L = scratch + 2 * static_cast<size_t>(T.start); // at most 2 n - 1 nodes for n primitives
	int cnt = 0;
	struct Frame {
		int start, end, parent;
	};
	Frame stack[kTaskStack];
	int sp = 0;
	stack[sp++] = Frame{ T.start, T.end, -1 };
	while (sp > 0) {
		const Frame fr = stack[--sp];
		const int node = cnt++;
		tyr_bvh_node nd;
		memset(&nd, 0, sizeof nd); // value-initialised: all 32 bytes zero (bvh.cpp:11)
		if (fr.parent >= 0)
			L[fr.parent].offset = node; // secondChildOffset (bvh.cpp:203): a frame with a parent is its second child
		const int n = fr.end - fr.start;
		BoxD nodeBox;
		nodeBox.init();
		for (int i = fr.start; i < fr.end; ++i)
			nodeBox.unite(info[i].lo, info[i].hi);
		bool leaf = true;
		int dim = 0, mid = 0;
		if (n > 1) {
			BoxD cbox;
			cbox.init();
			for (int i = fr.start; i < fr.end; ++i)
				cbox.add(info[i].c);
			dim = cbox.largestExtent();
			const float cb = cbox.lo[dim], ct = cbox.hi[dim];
			if (cb != ct) {
				int count[kBuckets];
				BoxD bounds[kBuckets];
				for (int b = 0; b < kBuckets; ++b) {
					count[b] = 0;
					bounds[b].init();
				}
				for (int i = fr.start; i < fr.end; ++i) {
					const int b = bucket_of(info[i].c[dim], cb, ct);
					++count[b];
					bounds[b].unite(info[i].lo, info[i].hi);
				}
				const int minBucket = sah_split(count, bounds, nodeBox, n);
				if (minBucket >= 0) {
					// std::partition, bvh.cpp:171-178 (libstdc++'s bidirectional form, host/bvh_build.cpp decide())
					int first = fr.start, last = fr.end;
					for (;;) {
						while (first != last && bucket_of(info[first].c[dim], cb, ct) <= minBucket)
							++first;
						if (first == last)
							break;
						--last;
						while (first != last && !(bucket_of(info[last].c[dim], cb, ct) <= minBucket))
							--last;
						if (first == last)
							break;
						const Info tmp = info[first];
						info[first] = info[last];
						info[last] = tmp;
						++first;
					}
					mid = first;
					leaf = false;
				}
			}
		}
		if (leaf) { // bvh.cpp:80-84, 214-218
			for (int k = 0; k < 3; ++k) {
				nd.bbox.bounds[0][k] = nodeBox.lo[k];
				nd.bbox.bounds[1][k] = nodeBox.hi[k];
			}
			nd.offset = fr.start - T.start;
			nd.primitiveCount = static_cast<uint16_t>(n);
			L[node] = nd;
		} else {
			nd.splitAxis = static_cast<uint8_t>(dim);
			L[node] = nd;
			if (sp + 2 > kTaskStack) {
				atomicExch(&K->error, 1);
				return;
			}
			stack[sp++] = Frame{ mid, fr.end, node };  // second child: later
			stack[sp++] = Frame{ fr.start, mid, -1 };  // first child: next (index node + 1)
		}
	}
	// initInterior (bvh.cpp:220-225): Union(first child, second child), children before parents
	for (int node = cnt - 1; node >= 0; --node) {
		tyr_bvh_node& nd = L[node];
		if (nd.primitiveCount > 0)
			continue;
		const tyr_bvh_node &l = L[node + 1], &r = L[nd.offset];
		BoxD b;
		for (int k = 0; k < 3; ++k) {
			b.lo[k] = l.bbox.bounds[0][k];
			b.hi[k] = l.bbox.bounds[1][k];
		}
		b.unite(r.bbox.bounds[0], r.bbox.bounds[1]);
		for (int k = 0; k < 3; ++k) {
			nd.bbox.bounds[0][k] = b.lo[k];
			nd.bbox.bounds[1][k] = b.hi[k];
		}
	}
	T.size = cnt;
	for (int k = 0; k < 3; ++k) {
		T.lo[k] = L[0].bbox.bounds[0][k];
		T.hi[k] = L[0].bbox.bounds[1][k];
	}
}

// sizes and boxes bottom-up (nodes [lo, hi) = one level: its children were made on the next one), bases top-down
__global__ void k_sizes(TopNode* nodes, int lo, int hi) {
	const int t = lo + blockIdx.x * kBlockB + threadIdx.x;
	if (t >= hi || nodes[t].state != kInterior)
		return;
	TopNode& N = nodes[t];
	const TopNode &l = nodes[N.left], &r = nodes[N.right];
	N.size = 1 + l.size + r.size;
	BoxD b;
	for (int k = 0; k < 3; ++k) {
		b.lo[k] = l.lo[k];
		b.hi[k] = l.hi[k];
	}
	b.unite(r.lo, r.hi);
	for (int k = 0; k < 3; ++k) {
		N.lo[k] = b.lo[k];
		N.hi[k] = b.hi[k];
	}
}
__global__ void k_bases(TopNode* nodes, int lo, int hi) {
	const int t = lo + blockIdx.x * kBlockB + threadIdx.x;
	if (t >= hi || nodes[t].state != kInterior)
		return;
	const TopNode& N = nodes[t];
	nodes[N.left].base = N.base + 1;
	nodes[N.right].base = N.base + 1 + nodes[N.left].size;
}
__global__ void k_emit(const TopNode* __restrict__ nodes, const tyr_bvh_node* __restrict__ scratch, tyr_bvh_node* __restrict__ out, int nTop) {
	const int t = blockIdx.x * 64 + threadIdx.x;
	if (t >= nTop)
		return;
	const TopNode& N = nodes[t];
	if (N.state == kInterior) {
		tyr_bvh_node nd;
		memset(&nd, 0, sizeof nd);
		for (int k = 0; k < 3; ++k) {
			nd.bbox.bounds[0][k] = N.lo[k];
			nd.bbox.bounds[1][k] = N.hi[k];
		}
		nd.offset = nodes[N.right].base;
		nd.primitiveCount = 0;
		nd.splitAxis = static_cast<uint8_t>(N.dim);
		out[N.base] = nd;
	} else if (N.state == kTask) {
		const tyr_bvh_node* L = scratch + 2 * static_cast<size_t>(N.start);
		for (int i = 0; i < N.size; ++i) {
			tyr_bvh_node nd = L[i];
			nd.offset += nd.primitiveCount > 0 ? N.start : N.base; // (the primitive order is the info array's: a subtree's first primitive is its range's start)
			out[N.base + i] = nd;
		}
	}
}
__global__ void k_emit_prims(const Info* __restrict__ info, const tyr_triangle* __restrict__ prims, tyr_triangle* __restrict__ out, int n) {
	const int i = blockIdx.x * kBlockB + threadIdx.x;
	if (i < n)
		out[i] = prims[info[i].idx]; // bvh.cpp:24: primitives.swap(orderedPrims)
}

template <class T>
struct DevBuf {
	T* p = nullptr;
	hipError_t alloc(size_t count) { return hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T)); }
	~DevBuf() {
		if (p)
			(void)hipFree(p);
	}
};
inline int grid(int n, int b = kBlockB) { return (n + b - 1) / b; }
struct PoolView { // typed views into one device allocation
	char* base;
	template <class T>
	T* at(size_t off) const { return reinterpret_cast<T*>(base + off); }
};

} // namespace

DeviceBuild::~DeviceBuild() {
	if (pool)
		(void)hipFree(pool);
}

#define TYR_D(expr)                                                     \
	do {                                                                \
		const hipError_t e_ = (expr);                                   \
		if (e_ != hipSuccess)                                           \
			return e_ == hipErrorOutOfMemory ? TYR_ERR_OOM : static_cast<int>(e_); \
	} while (0)

// The build, its results LEFT ON THE DEVICE (out.nodes: the reference's node array, out.prims: the primitives in their final order;
// both inside out.pool, which out's destructor frees).  prims / bboxes are host arrays and are not touched.  Returns the node count
// (> 0) or a negative status.  seconds_out (may be null): [0] the device's work (first kernel to last, hipEvents), [1] the copies in.
// TYR_ERR_UNSUPPORTED: a task thread's stack overflowed (a degenerate range far beyond kTaskPrims): use the host builder.
int bvh_build_device_keep(int device, const tyr_triangle* prims, int32_t n, const tyr_bbox* bboxes, DeviceBuild& out, double* seconds_out) {
	if (n <= 0 || !prims || !bboxes)
		return TYR_ERR_INVALID;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
		return TYR_ERR_NO_DEVICE;
	// The caller's current device is put back on every way out, and the build runs on a stream of its own: hosts that call this also
	// use torch or other HIP code, and the legacy NULL stream synchronises implicitly with every blocking stream of the process.  The
	// guard's destructor WAITS for the stream first -- asynchronous copies below read and write locals of this function (root0, k0, s0,
	// hK, root), which must outlive them on the early returns too.
	struct DeviceAndStream {
		int prev = -1;
		hipStream_t st = nullptr;
		~DeviceAndStream() {
			if (st) {
				(void)hipStreamSynchronize(st);
				(void)hipStreamDestroy(st);
			}
			if (prev >= 0)
				(void)hipSetDevice(prev);
		}
	} scope;
	if (hipGetDevice(&scope.prev) != hipSuccess)
		scope.prev = -1;
	TYR_D(hipSetDevice(device));
	TYR_D(hipStreamCreateWithFlags(&scope.st, hipStreamNonBlocking));
	const size_t N = static_cast<size_t>(n);
	int kTaskPrims = kTaskPrimsDefault;
	if (const char* e = std::getenv("TYR_DEVBUILD_TASK_PRIMS"))
		kTaskPrims = std::max(4, std::min(std::atoi(e), 4096));
	// one allocation, carved up (nineteen hipMalloc / hipFree pairs were 35 ms of a 50 ms call around 9 ms of work)
	const size_t maxSlots = N / (kTaskPrims + 1) + 2, maxTop = 2 * N + 2, nScanBlocks = (N + 1 + kScanPerBlock - 1) / kScanPerBlock;
	struct Carve {
		size_t total = 0;
		size_t take(size_t bytes) {
			const size_t at = total;
			total += (bytes + 255) & ~size_t(255);
			return at;
		}
	} carve;
	const size_t oBB = carve.take(N * sizeof(tyr_bbox)), oPrims = carve.take(N * sizeof(tyr_triangle)), oPrimsOut = carve.take(N * sizeof(tyr_triangle));
	const size_t oInfo0 = carve.take(N * sizeof(Info)), oInfo1 = carve.take(N * sizeof(Info)), oSlotOf0 = carve.take(N * sizeof(int)), oSlotOf1 = carve.take(N * sizeof(int));
	const size_t oBkt = carve.take(N), oFlags = carve.take((N + 1) * sizeof(uint2)), oExcl = carve.take((N + 4) * sizeof(uint2)), oBlockSums = carve.take(nScanBlocks * sizeof(uint2));
	const size_t oPosML = carve.take(N * sizeof(uint32_t)), oPosMR = carve.take(N * sizeof(uint32_t)), oSlots0 = carve.take(maxSlots * sizeof(Slot)), oSlots1 = carve.take(maxSlots * sizeof(Slot));
	const size_t oNodes = carve.take(maxTop * sizeof(TopNode)), oScratch = carve.take(2 * N * sizeof(tyr_bvh_node)), oOut = carve.take(2 * N * sizeof(tyr_bvh_node)), oK = carve.take(sizeof(Counters));
	if (out.pool)
		(void)hipFree(out.pool);
	out.pool = nullptr;
	out.nodes = nullptr;
	out.prims = nullptr;
	out.nNodes = out.n = 0;
	TYR_D(hipMalloc(&out.pool, carve.total)); // (freed by out's destructor, whichever way this function is left)
	const PoolView mem{ static_cast<char*>(out.pool) };
	struct { tyr_bbox* p; } dBB{ mem.at<tyr_bbox>(oBB) };
	struct { tyr_triangle* p; } dPrims{ mem.at<tyr_triangle>(oPrims) }, dPrimsOut{ mem.at<tyr_triangle>(oPrimsOut) };
	struct { Info* p; } dInfo[2] = { { mem.at<Info>(oInfo0) }, { mem.at<Info>(oInfo1) } };
	struct { int* p; } dSlotOf[2] = { { mem.at<int>(oSlotOf0) }, { mem.at<int>(oSlotOf1) } };
	struct { uint8_t* p; } dBkt{ mem.at<uint8_t>(oBkt) };
	struct { uint2* p; } dFlags{ mem.at<uint2>(oFlags) }, dExcl{ mem.at<uint2>(oExcl) }, dBlockSums{ mem.at<uint2>(oBlockSums) };
	struct { uint32_t* p; } dPosML{ mem.at<uint32_t>(oPosML) }, dPosMR{ mem.at<uint32_t>(oPosMR) };
	struct { Slot* p; } dSlots[2] = { { mem.at<Slot>(oSlots0) }, { mem.at<Slot>(oSlots1) } };
	struct { TopNode* p; } dNodes{ mem.at<TopNode>(oNodes) };
	struct { tyr_bvh_node* p; } dScratch{ mem.at<tyr_bvh_node>(oScratch) }, dOut{ mem.at<tyr_bvh_node>(oOut) };
	struct { Counters* p; } dK{ mem.at<Counters>(oK) };
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	TYR_D(hipEventCreate(&ev0));
	TYR_D(hipEventCreate(&ev1));
	struct EvGuard {
		hipEvent_t a, b;
		~EvGuard() {
			(void)hipEventDestroy(a);
			(void)hipEventDestroy(b);
		}
	} evGuard{ ev0, ev1 };
	const auto tCopy0 = std::chrono::steady_clock::now();
	TYR_D(hipMemcpy(dBB.p, bboxes, N * sizeof(tyr_bbox), hipMemcpyHostToDevice));
	TYR_D(hipMemcpy(dPrims.p, prims, N * sizeof(tyr_triangle), hipMemcpyHostToDevice));
	double copyS = std::chrono::duration<double>(std::chrono::steady_clock::now() - tCopy0).count();
	const hipStream_t st = scope.st;
	TYR_D(hipEventRecord(ev0, st));
	// the root (sources of asynchronous copies: function scope, they live until the stream has been waited for)
	TopNode root0{};
	root0.start = 0;
	root0.end = n;
	root0.state = n <= kTaskPrims ? kTask : kActive;
	root0.left = root0.right = -1;
	TYR_D(hipMemcpyAsync(dNodes.p, &root0, sizeof root0, hipMemcpyHostToDevice, st));
	const Counters k0{ 1, 0, 0 };
	TYR_D(hipMemcpyAsync(dK.p, &k0, sizeof k0, hipMemcpyHostToDevice, st));
	Slot s0{};
	s0.node = 0;
	TYR_D(hipMemcpyAsync(dSlots[0].p, &s0, sizeof(int), hipMemcpyHostToDevice, st)); // (only `node`: k_slot_reset and k_decide_dim fill the rest)
	hipLaunchKernelGGL(k_init_info, dim3(grid(n)), dim3(kBlockB), 0, st, dBB.p, dInfo[0].p, dSlotOf[0].p, n, n <= kTaskPrims ? -1 : 0, dK.p);
	int cur = 0, nSlots = n <= kTaskPrims ? 0 : 1;
	std::vector<int> levelStart{ 0 }; // top-node ids [levelStart[l], levelStart[l + 1]) were made by level l - 1's splits (level 0: the root)
	int nTop = 1;
	Counters hK{};
	while (nSlots > 0) {
		levelStart.push_back(nTop);
		Slot* S = dSlots[cur].p;
		Slot* Snext = dSlots[cur ^ 1].p;
		const Info* in = dInfo[cur].p;
		Info* out = dInfo[cur ^ 1].p;
		const bool rootLevel = levelStart.size() == 2;
		hipLaunchKernelGGL(k_slot_reset, dim3(grid(nSlots)), dim3(kBlockB), 0, st, S, nSlots, rootLevel ? 1 : 0);
		if (rootLevel) // (every other range got its bounds from its parent's buckets, k_decide_split)
			hipLaunchKernelGGL(k_bounds, dim3(grid(n)), dim3(kBlockB), 0, st, in, dSlotOf[cur].p, S, n);
		hipLaunchKernelGGL(k_decide_dim, dim3(grid(nSlots)), dim3(kBlockB), 0, st, S, dNodes.p, nSlots);
		hipLaunchKernelGGL(k_buckets, dim3(grid(n)), dim3(kBlockB), 0, st, in, dSlotOf[cur].p, S, dBkt.p, n);
		hipLaunchKernelGGL(k_decide_split, dim3(grid(nSlots)), dim3(kBlockB), 0, st, S, Snext, dNodes.p, dK.p, nSlots, kTaskPrims);
		hipLaunchKernelGGL(k_flags, dim3(grid(n)), dim3(kBlockB), 0, st, dSlotOf[cur].p, S, dBkt.p, dFlags.p, n);
		hipLaunchKernelGGL(k_scan_sums, dim3(static_cast<unsigned>(nScanBlocks)), dim3(kBlockB), 0, st, dFlags.p, dBlockSums.p, n);
		hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(kBlockB), 0, st, dBlockSums.p, static_cast<int>(nScanBlocks));
		hipLaunchKernelGGL(k_scan_final, dim3(static_cast<unsigned>(nScanBlocks)), dim3(kBlockB), 0, st, dFlags.p, dBlockSums.p, dExcl.p, n);
		hipLaunchKernelGGL(k_lists, dim3(grid(n)), dim3(kBlockB), 0, st, dFlags.p, dExcl.p, dPosML.p, dPosMR.p, n);
		hipLaunchKernelGGL(k_scatter, dim3(grid(n)), dim3(kBlockB), 0, st, in, out, dSlotOf[cur].p, dSlotOf[cur ^ 1].p, S, dNodes.p, dFlags.p, dExcl.p, dPosML.p, dPosMR.p, n);
		TYR_D(hipMemcpyAsync(&hK, dK.p, sizeof hK, hipMemcpyDeviceToHost, st));
		TYR_D(hipStreamSynchronize(st));
		if (hK.error == 2)
			return TYR_ERR_INVALID;
		nTop = hK.nNodes;
		nSlots = hK.nNext;
		if (static_cast<size_t>(nSlots) > maxSlots || static_cast<size_t>(nTop) > maxTop)
			return TYR_ERR_DEVICE;
		hK.nNext = 0;
		TYR_D(hipMemcpyAsync(dK.p, &hK, sizeof hK, hipMemcpyHostToDevice, st));
		cur ^= 1;
		if (levelStart.size() > 4096)
			return TYR_ERR_DEVICE;
	}
	levelStart.push_back(nTop);
	Info* finalInfo = dInfo[cur].p;
	hipLaunchKernelGGL(k_tasks, dim3(grid(nTop, 64)), dim3(64), 0, st, finalInfo, dNodes.p, dScratch.p, dK.p, nTop);
	for (size_t l = levelStart.size() - 1; l-- > 0;) { // bottom-up
		const int lo = levelStart[l], hi = levelStart[l + 1];
		if (hi > lo)
			hipLaunchKernelGGL(k_sizes, dim3(grid(hi - lo)), dim3(kBlockB), 0, st, dNodes.p, lo, hi);
	}
	for (size_t l = 0; l + 1 < levelStart.size(); ++l) { // top-down (the root's base is 0)
		const int lo = levelStart[l], hi = levelStart[l + 1];
		if (hi > lo)
			hipLaunchKernelGGL(k_bases, dim3(grid(hi - lo)), dim3(kBlockB), 0, st, dNodes.p, lo, hi);
	}
	hipLaunchKernelGGL(k_emit, dim3(grid(nTop, 64)), dim3(64), 0, st, dNodes.p, dScratch.p, dOut.p, nTop);
	hipLaunchKernelGGL(k_emit_prims, dim3(grid(n)), dim3(kBlockB), 0, st, finalInfo, dPrims.p, dPrimsOut.p, n);
	TYR_D(hipEventRecord(ev1, st));
	TopNode root{};
	TYR_D(hipMemcpyAsync(&root, dNodes.p, sizeof root, hipMemcpyDeviceToHost, st));
	TYR_D(hipMemcpyAsync(&hK, dK.p, sizeof hK, hipMemcpyDeviceToHost, st));
	TYR_D(hipStreamSynchronize(st));
	TYR_D(hipGetLastError());
	if (hK.error)
		return hK.error == 2 ? TYR_ERR_INVALID : TYR_ERR_UNSUPPORTED;
	const int nNodes = root.size;
	if (nNodes <= 0 || static_cast<size_t>(nNodes) > 2 * N)
		return TYR_ERR_DEVICE;
	out.nodes = dOut.p;
	out.prims = dPrimsOut.p;
	out.nNodes = nNodes;
	out.n = n;
	if (seconds_out) {
		float ms = 0.0f;
		(void)hipEventElapsedTime(&ms, ev0, ev1);
		seconds_out[0] = ms * 1e-3;
		seconds_out[1] = copyS;
	}
	return nNodes;
}

// Returns the node count (>= 0) or a negative status.  prims / bboxes / nodes_out are HOST arrays, as tyr_bvh_build's; prims is
// reordered in place (bvh.cpp:24).  seconds_out (may be null): [0] the device's work, [1] the copies in and out.
int bvh_build_device(int device, tyr_triangle* prims, int32_t n, const tyr_bbox* bboxes, tyr_bvh_node* nodes_out, double* seconds_out) {
	if (n < 0 || (n > 0 && (!prims || !bboxes || !nodes_out)))
		return TYR_ERR_INVALID;
	if (n == 0)
		return 0; // bvh.cpp:8-10
	DeviceBuild B;
	double secs[2] = { 0.0, 0.0 };
	const int nNodes = bvh_build_device_keep(device, prims, n, bboxes, B, secs);
	if (nNodes < 0)
		return nNodes;
	const auto tCopy1 = std::chrono::steady_clock::now();
	TYR_D(hipMemcpy(nodes_out, B.nodes, static_cast<size_t>(nNodes) * sizeof(tyr_bvh_node), hipMemcpyDeviceToHost));
	TYR_D(hipMemcpy(prims, B.prims, static_cast<size_t>(n) * sizeof(tyr_triangle), hipMemcpyDeviceToHost));
	secs[1] += std::chrono::duration<double>(std::chrono::steady_clock::now() - tCopy1).count();
	if (seconds_out) {
		seconds_out[0] = secs[0];
		seconds_out[1] = secs[1];
	}
	return nNodes;
}
#undef TYR_D

} // namespace tyr
