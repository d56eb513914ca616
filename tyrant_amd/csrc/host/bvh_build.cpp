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
#include <cstring>
#include <vector>

#include "host.hpp"

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

class Builder {
public:
	Builder(const tyr_triangle* prims, int n, const tyr_bbox* bboxes, tyr_bvh_node* nodes, int algo)
		: prims_(prims), nodes_(nodes), algo_(algo), info_(static_cast<size_t>(n)) {
		ordered_.reserve(static_cast<size_t>(n));
		for (int i = 0; i < n; ++i) {
			PrimInfo& p = info_[static_cast<size_t>(i)];
			p.primitiveNumber = static_cast<uint32_t>(i);
			p.bbox = from_abi(bboxes[i]);
			for (int k = 0; k < 3; ++k)
				p.centroid[k] = p.bbox.lo[k] * 0.5f + p.bbox.hi[k] * 0.5f; // bvh.h:96
		}
	}
	int run(tyr_triangle* primsInOut) {
		build(0, static_cast<int>(info_.size()));
		std::memcpy(primsInOut, ordered_.data(), ordered_.size() * sizeof(tyr_triangle)); // bvh.cpp:24
		return nNodes_;
	}

private:
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
	void leaf(int node, int start, int end, const Box& box) { // bvh.cpp:80-84, 214-218
		const int first = static_cast<int>(ordered_.size());
		for (int i = start; i < end; ++i)
			ordered_.push_back(prims_[info_[static_cast<size_t>(i)].primitiveNumber]);
		to_abi(box, nodes_[node].bbox);
		nodes_[node].offset = first;
		nodes_[node].primitiveCount = static_cast<uint16_t>(end - start);
	}
	void build(int start, int end) { // bvh.cpp:61-212
		const int node = nNodes_++;
		Box nodeBox;
		for (int i = start; i < end; ++i)
			nodeBox = unite(nodeBox, info_[static_cast<size_t>(i)].bbox);
		const int n = end - start;
		if (n == 1) {
			leaf(node, start, end, nodeBox);
			return;
		}
		Box centroidBox;
		for (int i = start; i < end; ++i)
			centroidBox.add(info_[static_cast<size_t>(i)].centroid);
		const int dim = centroidBox.largestExtent();
		const float* cb = centroidBox.lo;
		const float* ct = centroidBox.hi;
		if (cb[dim] == ct[dim]) { // bvh.cpp:103-111
			leaf(node, start, end, nodeBox);
			return;
		}
		int mid = (start + end) / 2;
		if (algo_ == 1) {
			// EqualCounts, bvh.cpp:115-122.  nth_element's permutation is unspecified;
			// a stable sort by centroid satisfies its postcondition deterministically.
			std::stable_sort(info_.begin() + start, info_.begin() + end, [dim](const PrimInfo& a, const PrimInfo& b) { return a.centroid[dim] < b.centroid[dim]; });
		} else {
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
			const float nodeSA = nodeBox.surfaceArea();
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
			if (minBucket < 0) {
				// every cost was NaN/inf (zero-area node box): the reference asserts here (bvh.cpp:167);
				// a leaf keeps the tree valid
				leaf(node, start, end, nodeBox);
				return;
			}
			if (n > kMaxLeafPrims || minCost < leafCost) {
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
				mid = first;
			} else {
				leaf(node, start, end, nodeBox);
				return;
			}
		}
		build(start, mid);
		const int second = nNodes_;
		nodes_[node].offset = second; // secondChildOffset, bvh.cpp:203
		build(mid, end);
		// initInterior, bvh.cpp:220-225
		Box l = from_abi(nodes_[node + 1].bbox), r = from_abi(nodes_[second].bbox);
		to_abi(unite(l, r), nodes_[node].bbox);
		nodes_[node].primitiveCount = 0;
		nodes_[node].splitAxis = static_cast<uint8_t>(dim);
	}

	const tyr_triangle* prims_;
	tyr_bvh_node* nodes_;
	int algo_;
	int nNodes_ = 0;
	std::vector<PrimInfo> info_;
	std::vector<tyr_triangle> ordered_;
};

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
	Builder b(prims, n, bboxes, nodes_out, algo);
	return b.run(prims);
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
