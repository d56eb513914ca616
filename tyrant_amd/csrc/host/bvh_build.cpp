// bvh_build.cpp -- host binned-SAH BVH builder emitting the reference's flat node array.
//
// Reference: class BVH (bvh.h:49-108), BVH::BVH / computeBucket / recursiveBuild /
// initLeaf / initInterior (bvh.cpp:3-225), BBox host operations (Bbox.h:8-36,
// Bbox.cpp:3-14).  Contract kept byte for byte: 14 buckets, at most 4 primitives per
// SAH leaf, traversal cost 1, depth-first node order with the left child at index+1,
// leaf when one primitive is left / the centroid extent on the split axis is zero / the
// split does not pay (bvh.cpp:78, 103, 168); `primitives` is reordered in place.
//
// What is done differently from the reference (same output):
//   - the per-split SAH cost uses one prefix and one suffix sweep over the 14 buckets
//     instead of re-unioning every interval for every candidate (bvh.cpp:139-152);
//     unions are exact min/max, so the boxes -- and the costs -- are identical;
//   - std::partition (bvh.cpp:171-176) is written out as the bidirectional two-pointer
//     swap, the order libstdc++ and MSVC both produce, so the result does not depend on
//     the standard library;
//   - PrimitiveInfo is never copied by value into computeBucket (bvh.cpp:44).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#ifdef TYR_HOST_ONLY // the ThreadSanitizer build (make tsan): this file alone under g++ -fsanitize=thread, no HIP headers
#include "../../../include/tyr_c.h"
namespace tyr {
int bvh_build(tyr_triangle* prims, int32_t n, const tyr_bbox* bboxes, tyr_bvh_node* nodes_out, int32_t algo);
void triangle_bboxes(const tyr_triangle* prims, int32_t n, tyr_bbox* out);
void set_build_threads(int threads);
int build_threads();
} // namespace tyr
#else
#include "host.hpp"
#endif

namespace tyr {

namespace {

constexpr int kBuckets = 14;        // bvh.h:76
constexpr int kMaxLeafPrims = 4;    // bvh.h:78
constexpr float kTraversalCost = 1.0f;    // bvh.h:81
constexpr float kIntersectionCost = 1.0f; // bvh.h:84

// fmin / fmax as glibc evaluates them for non-NaN inputs: the first argument wins ties
inline float fminf_first(float a, float b) { return (b < a) ? b : a; }
inline float fmaxf_first(float a, float b) { return (b > a) ? b : a; }

struct Box {
	float lo[3], hi[3];
	Box() { // Bbox.h:5
		for (int k = 0; k < 3; ++k) {
			lo[k] = 1e10f;
			hi[k] = -1e10f;
		}
	}
	void add(const float v[3]) { // Bbox.h:8-14
		for (int k = 0; k < 3; ++k) {
			lo[k] = fminf_first(lo[k], v[k]);
			hi[k] = fmaxf_first(hi[k], v[k]);
		}
	}
	float surfaceArea() const { // Bbox.h:18-21
		const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
		return 2 * (dx * dy + dx * dz + dy * dz);
	}
	int largestExtent() const { // Bbox.h:28-36
		const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
		if (dx > dy && dx > dz)
			return 0;
		return (dy > dz) ? 1 : 2;
	}
};
inline Box unite(const Box& a, const Box& b) { // Bbox.cpp:3-14
	Box r;
	for (int k = 0; k < 3; ++k) {
		r.lo[k] = fminf_first(a.lo[k], b.lo[k]);
		r.hi[k] = fmaxf_first(a.hi[k], b.hi[k]);
	}
	return r;
}
inline Box from_abi(const tyr_bbox& b) {
	Box r;
	std::memcpy(r.lo, b.bounds[0], 12);
	std::memcpy(r.hi, b.bounds[1], 12);
	return r;
}
inline void to_abi(const Box& b, tyr_bbox& out) {
	std::memcpy(out.bounds[0], b.lo, 12);
	std::memcpy(out.bounds[1], b.hi, 12);
}

struct PrimInfo { // bvh.h:88-97
	uint32_t primitiveNumber;
	Box bbox;
	float centroid[3];
};

// What one call of recursiveBuild decides for the range [start, end) before it recurses (bvh.cpp:61-193)
struct Split {
	Box nodeBox;
	bool isLeaf;
	int dim;
	int mid;
};

// Output of one subtree, with LOCAL indices: node 0 is the subtree's root, interior offsets count from it,
// leaf offsets count from the subtree's first ordered primitive.
struct Local {
	std::vector<tyr_bvh_node> nodes;
	std::vector<tyr_triangle> ordered;
};

class Builder {
public:
	Builder(const tyr_triangle* prims, int n, const tyr_bbox* bboxes, int algo) : prims_(prims), algo_(algo), info_(static_cast<size_t>(n)) {
		for (int i = 0; i < n; ++i) {
			PrimInfo& p = info_[static_cast<size_t>(i)];
			p.primitiveNumber = static_cast<uint32_t>(i);
			p.bbox = from_abi(bboxes[i]);
			for (int k = 0; k < 3; ++k)
				p.centroid[k] = p.bbox.lo[k] * 0.5f + p.bbox.hi[k] * 0.5f; // bvh.h:96
		}
	}

	// The reference builder is serial (bvh.cpp:61-212).  Here the top of the tree fans out over a fixed pool of
	// worker threads: a range above the grain size is split by one worker (the same two-pointer partition) and
	// its two halves go back on the shared FIFO; a range below it is built serially into LOCAL buffers.  A
	// placement pass then gives every subtree its depth-first position and the workers copy the subtrees there,
	// rebasing offsets, so the node array and the primitive order are byte-identical to the serial build
	// (tests/test_host_and_abi.py compares both with the oracle's).
	int run(tyr_triangle* primsOut, tyr_bvh_node* nodesOut, int threads) {
		const int n = static_cast<int>(info_.size());
		threads = std::max(1, std::min(threads, 1 + n / kMinGrain));
		grain_ = threads > 1 ? std::max(kMinGrain, n / (threads * 8)) : n;
		const bool trace = std::getenv("TYR_BUILD_TRACE") != nullptr;
		const auto t0 = std::chrono::steady_clock::now();
		auto lap = [&](const char* what) {
			if (trace)
				std::fprintf(stderr, "[tyr_bvh_build] %-10s +%.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
		};
		Top root;
		root.start = 0;
		root.end = n;
		queue_.push_back(&root);
		pending_ = 1;
		run_workers(threads, [this] { build_worker(); });
		lap("subtrees");
		// depth-first placement: bases first (walks only the top of the tree), then parallel copies
		std::vector<Top*> tasks;
		int nNodes = 0, nPrims = 0;
		assign(root, nNodes, nPrims, tasks);
		std::atomic<size_t> next{0};
		run_workers(threads, [&] {
			for (size_t i = next.fetch_add(1); i < tasks.size(); i = next.fetch_add(1))
				copy_out(*tasks[i], nodesOut, primsOut);
		});
		finish_top(root, nodesOut);
		lap("placement");
		return nNodes;
	}

private:
	static constexpr int kMinGrain = 4096;

	// top of the tree: either a split made on the way down, or a finished subtree
	struct Top {
		int start = 0, end = 0;
		bool isTask = false;
		Local local;        // isTask
		int dim = 0;        // !isTask
		int nodeBase = 0, primBase = 0;
		std::unique_ptr<Top> left, right;
	};

	template <class F>
	static void run_workers(int threads, F&& body) {
		std::vector<std::thread> pool;
		for (int i = 1; i < threads; ++i)
			pool.emplace_back(body);
		body();
		for (std::thread& t : pool)
			t.join();
	}

	void build_worker() {
		for (;;) {
			Top* t = nullptr;
			{
				std::unique_lock<std::mutex> lock(mutex_);
				wake_.wait(lock, [this] { return !queue_.empty() || pending_ == 0; });
				if (queue_.empty())
					return; // pending_ == 0: every range has been built
				t = queue_.front(); // FIFO: the large ranges are split first, the leaf tasks balance the tail
				queue_.pop_front();
			}
			bool split = false;
			if (t->end - t->start > grain_) {
				const Split s = decide(t->start, t->end);
				if (!s.isLeaf) {
					t->dim = s.dim;
					t->left.reset(new Top());
					t->right.reset(new Top());
					t->left->start = t->start;
					t->left->end = t->right->start = s.mid;
					t->right->end = t->end;
					split = true;
				}
				// else: one huge leaf (identical centroids); decide() leaves such a range untouched, build_serial emits it
			}
			if (!split) {
				t->isTask = true;
				t->local.nodes.reserve(static_cast<size_t>(t->end - t->start));
				t->local.ordered.reserve(static_cast<size_t>(t->end - t->start));
				build_serial(t->start, t->end, t->local);
			}
			{
				std::lock_guard<std::mutex> lock(mutex_);
				if (split) {
					queue_.push_back(t->left.get());
					queue_.push_back(t->right.get());
					pending_ += 2;
				}
				--pending_;
			}
			wake_.notify_all();
		}
	}

	// bvh.cpp:44-58
	static int bucketOf(const PrimInfo& p, const float cb[3], const float ct[3], int dim) {
		float distance = p.centroid[dim] - cb[dim];
		if (ct[dim] > cb[dim])
			distance = distance / (ct[dim] - cb[dim]);
		int b = static_cast<int>(kBuckets * distance);
		if (b == kBuckets)
			--b;
		return b;
	}

	Split decide(int start, int end) { // bvh.cpp:61-193, everything before the two recursive calls
		Split s;
		s.isLeaf = true;
		s.dim = 0;
		s.mid = (start + end) / 2;
		for (int i = start; i < end; ++i)
			s.nodeBox = unite(s.nodeBox, info_[static_cast<size_t>(i)].bbox);
		const int n = end - start;
		if (n == 1)
			return s;
		Box centroidBox;
		for (int i = start; i < end; ++i)
			centroidBox.add(info_[static_cast<size_t>(i)].centroid);
		const int dim = centroidBox.largestExtent();
		s.dim = dim;
		const float* cb = centroidBox.lo;
		const float* ct = centroidBox.hi;
		if (cb[dim] == ct[dim]) // bvh.cpp:103-111
			return s;
		if (algo_ == 1) {
			// EqualCounts, bvh.cpp:115-122.  nth_element's permutation is unspecified;
			// a stable sort by centroid satisfies its postcondition deterministically.
			std::stable_sort(info_.begin() + start, info_.begin() + end, [dim](const PrimInfo& a, const PrimInfo& b) { return a.centroid[dim] < b.centroid[dim]; });
			s.isLeaf = false;
			return s;
		}
		int count[kBuckets] = {};
		Box bounds[kBuckets];
		for (int i = start; i < end; ++i) {
			const PrimInfo& p = info_[static_cast<size_t>(i)];
			const int b = bucketOf(p, cb, ct, dim);
			++count[b];
			bounds[b] = unite(bounds[b], p.bbox);
		}
		// suffix sweep: box/count of buckets (c, 13]
		Box sufBox[kBuckets];
		int sufCount[kBuckets];
		{
			Box acc;
			int c = 0;
			for (int b = kBuckets - 1; b >= 1; --b) {
				acc = unite(bounds[b], acc);
				c += count[b];
				sufBox[b - 1] = acc;
				sufCount[b - 1] = c;
			}
		}
		const float nodeSA = s.nodeBox.surfaceArea();
		float minCost = FLT_MAX;
		int minBucket = -1;
		Box pre;
		int preCount = 0;
		for (int c = 0; c < kBuckets - 1; ++c) {
			pre = unite(pre, bounds[c]);
			preCount += count[c];
			const float cost = kTraversalCost + (static_cast<float>(preCount) * pre.surfaceArea() + static_cast<float>(sufCount[c]) * sufBox[c].surfaceArea()) / nodeSA;
			if (cost < minCost) {
				minCost = cost;
				minBucket = c;
			}
		}
		const float leafCost = kIntersectionCost * static_cast<float>(n);
		if (minBucket < 0)
			return s; // every cost was NaN/inf (zero-area node box): the reference asserts here (bvh.cpp:167); a leaf keeps the tree valid
		if (!(n > kMaxLeafPrims || minCost < leafCost))
			return s;
		// std::partition(begin, end, bucket <= minBucket), bvh.cpp:171-178
		int first = start, last = end;
		for (;;) {
			while (first != last && bucketOf(info_[static_cast<size_t>(first)], cb, ct, dim) <= minBucket)
				++first;
			if (first == last)
				break;
			--last;
			while (first != last && !(bucketOf(info_[static_cast<size_t>(last)], cb, ct, dim) <= minBucket))
				--last;
			if (first == last)
				break;
			std::swap(info_[static_cast<size_t>(first)], info_[static_cast<size_t>(last)]);
			++first;
		}
		s.mid = first;
		s.isLeaf = false;
		return s;
	}

	void build_serial(int start, int end, Local& L) { // bvh.cpp:61-212 into local buffers
		const size_t node = L.nodes.size();
		L.nodes.emplace_back(); // value-initialised: all 32 bytes zero (bvh.cpp:11)
		const Split s = decide(start, end);
		if (s.isLeaf) { // bvh.cpp:80-84, 214-218
			const int first = static_cast<int>(L.ordered.size());
			for (int i = start; i < end; ++i)
				L.ordered.push_back(prims_[info_[static_cast<size_t>(i)].primitiveNumber]);
			to_abi(s.nodeBox, L.nodes[node].bbox);
			L.nodes[node].offset = first;
			L.nodes[node].primitiveCount = static_cast<uint16_t>(end - start);
			return;
		}
		build_serial(start, s.mid, L);
		const size_t second = L.nodes.size();
		L.nodes[node].offset = static_cast<int32_t>(second); // secondChildOffset, bvh.cpp:203
		build_serial(s.mid, end, L);
		// initInterior, bvh.cpp:220-225
		const Box l = from_abi(L.nodes[node + 1].bbox), r = from_abi(L.nodes[second].bbox);
		to_abi(unite(l, r), L.nodes[node].bbox);
		L.nodes[node].primitiveCount = 0;
		L.nodes[node].splitAxis = static_cast<uint8_t>(s.dim);
	}

	void assign(Top& t, int& nNodes, int& nPrims, std::vector<Top*>& tasks) {
		t.nodeBase = nNodes;
		t.primBase = nPrims;
		if (t.isTask) {
			nNodes += static_cast<int>(t.local.nodes.size());
			nPrims += static_cast<int>(t.local.ordered.size());
			tasks.push_back(&t);
			return;
		}
		++nNodes;
		assign(*t.left, nNodes, nPrims, tasks);
		assign(*t.right, nNodes, nPrims, tasks);
	}

	static void copy_out(const Top& t, tyr_bvh_node* nodesOut, tyr_triangle* primsOut) {
		for (size_t i = 0; i < t.local.nodes.size(); ++i) {
			tyr_bvh_node nd = t.local.nodes[i];
			nd.offset += nd.primitiveCount > 0 ? t.primBase : t.nodeBase;
			nodesOut[static_cast<size_t>(t.nodeBase) + i] = nd;
		}
		if (!t.local.ordered.empty())
			std::memcpy(primsOut + t.primBase, t.local.ordered.data(), t.local.ordered.size() * sizeof(tyr_triangle));
	}

	// the interior nodes above the tasks, children before parents (initInterior, bvh.cpp:220-225)
	static void finish_top(const Top& t, tyr_bvh_node* nodesOut) {
		if (t.isTask)
			return;
		finish_top(*t.left, nodesOut);
		finish_top(*t.right, nodesOut);
		tyr_bvh_node& nd = nodesOut[t.nodeBase];
		std::memset(&nd, 0, sizeof nd);
		const Box l = from_abi(nodesOut[t.left->nodeBase].bbox), r = from_abi(nodesOut[t.right->nodeBase].bbox);
		to_abi(unite(l, r), nd.bbox);
		nd.offset = t.right->nodeBase;
		nd.primitiveCount = 0;
		nd.splitAxis = static_cast<uint8_t>(t.dim);
	}

	const tyr_triangle* prims_;
	int algo_;
	int grain_ = 0;
	std::vector<PrimInfo> info_;
	std::mutex mutex_;
	std::condition_variable wake_;
	std::deque<Top*> queue_;
	int pending_ = 0;
};

int g_buildThreads = 0; // 0 = auto

} // namespace

int bvh_build(tyr_triangle* prims, int32_t n, const tyr_bbox* bboxes, tyr_bvh_node* nodes_out, int32_t algo) {
	if (n < 0 || (n > 0 && (!prims || !bboxes || !nodes_out)))
		return TYR_ERR_INVALID;
	if (algo != 1 && algo != 2)
		return TYR_ERR_INVALID; // PartitionAlgorithm::Middle is unimplemented in the reference (bvh.cpp:190-193)
	if (n == 0)
		return 0; // bvh.cpp:8-10
	for (int i = 0; i < n; ++i)
		for (int k = 0; k < 6; ++k)
			if (!std::isfinite((&bboxes[i].bounds[0][0])[k]))
				return TYR_ERR_INVALID;
	std::memset(nodes_out, 0, sizeof(tyr_bvh_node) * (static_cast<size_t>(n) * 2 - 1)); // vector::resize value-initialises, bvh.cpp:11
	const int threads = build_threads();
	Builder b(prims, n, bboxes, algo);
	// every subtree keeps its own ordered copy until all of them are built, so the result can replace `prims` in place (bvh.cpp:24)
	return b.run(prims, nodes_out, threads);
}

void set_build_threads(int threads) { g_buildThreads = threads; }
int build_threads() {
	int threads = g_buildThreads;
	if (threads <= 0) {
		if (const char* e = std::getenv("TYR_BUILD_THREADS"))
			threads = std::atoi(e);
		if (threads <= 0)
			threads = static_cast<int>(std::min(16u, std::max(1u, std::thread::hardware_concurrency())));
	}
	return threads;
}

// Scene.cpp:22-33: BBox over the three vertices; the stored form gives them as vert, vert+e1, vert+e2
void triangle_bboxes(const tyr_triangle* prims, int32_t n, tyr_bbox* out) {
	for (int i = 0; i < n; ++i) {
		const tyr_triangle& t = prims[i];
		float v1[3], v2[3];
		for (int k = 0; k < 3; ++k) {
			v1[k] = t.vert[k] + t.e1[k];
			v2[k] = t.vert[k] + t.e2[k];
		}
		Box b;
		b.add(t.vert);
		b.add(v1);
		b.add(v2);
		to_abi(b, out[i]);
	}
}

} // namespace tyr
