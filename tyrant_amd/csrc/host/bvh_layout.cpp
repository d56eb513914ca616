// bvh_layout.cpp -- the upload half of Scene::Load (Scene.cpp:55-67) for gfx950: turns the
// reference's flat depth-first 32-byte node array (bvh.h:55-68) into the private device
// layout documented in hip/traverse.hpp (128-byte quad nodes, 64-byte child-pair nodes for the counting build,
// 48-byte triangles).  The tree itself -- which primitive is in which leaf, which child is "first", every box --
// is unchanged, so the traversal visit order of bvh.h:118-161 is preserved.
//
// Round 5: every pass runs on the host builder's threads (tyr_set_build_threads).  Rounds 1-4 did this serially -- 0.17 s for
// C3's 1.1 M nodes, 2 s for C5's 12.6 M, behind a 16-thread build that takes 0.1 s / 1.1 s and in front of a 5 ms render.
// What makes it parallel is the reference's own layout: depth-first, left child = index + 1 (bvh.cpp:195-202), so the subtree
// of a node is a CONTIGUOUS index range and nothing below a cut through the top of the tree refers to anything outside its
// range.  Per-node work (triangles, boxes, references) is a plain parallel loop; what flows down the tree (depth parity: every
// interior node at even depth is the root of a quad record) or up (the deepest stack a traversal can need) is done per
// subtree range below the cut, and serially for the few nodes above it; numberings are chunked prefix sums.  A quad record is
// written ONCE, at its final position with its final references (rounds 1-4 wrote, moved and patched the array in three
// passes).  The bytes do not depend on the thread count (tests/test_host_and_abi.py; tyr_layout_probe).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <thread>
#include <vector>

#include <exception>
#include <mutex>

#include "host.hpp"

namespace tyr {

namespace {

inline uint32_t leaf_ref(uint32_t off, uint32_t cnt) { return kRefLeaf | ((cnt - 1u) << 26) | off; }
inline float bits(uint32_t u) {
	float f;
	std::memcpy(&f, &u, 4);
	return f;
}
inline bool finite3(const float* p) { return std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]); }

// f(chunk, begin, end) over [0, n) cut into at most `threads` contiguous chunks of at least `grain` items.
// No exception leaves a worker thread (that would be std::terminate, whatever the C entry point catches): a worker's exception is
// kept, every thread is joined -- also when the calling thread's share or a std::thread constructor throws -- and the first one is
// rethrown on the calling thread, where tyr_scene_upload / tyr_layout_probe / tyr_scene_build_upload turn it into TYR_ERR_OOM.
struct Par {
	int threads;
	size_t chunks(size_t n, size_t grain) const { return std::max<size_t>(1, std::min<size_t>(static_cast<size_t>(std::max(threads, 1)), n / std::max<size_t>(grain, 1))); }
	struct Joined { // joins what was started, whatever happens between the first emplace_back and the end of the scope
		std::vector<std::thread> pool;
		~Joined() {
			for (std::thread& th : pool)
				if (th.joinable())
					th.join();
		}
	};
	struct FirstError {
		std::mutex m;
		std::exception_ptr e;
		void keep() noexcept {
			std::lock_guard<std::mutex> g(m);
			if (!e)
				e = std::current_exception();
		}
	};
	template <class F>
	void run(size_t n, size_t grain, F&& f) const {
		const size_t c = chunks(n, grain);
		if (c <= 1) {
			f(size_t(0), size_t(0), n);
			return;
		}
		FirstError err;
		{
			Joined j;
			j.pool.reserve(c - 1);
			try {
				for (size_t t = 1; t < c; ++t)
					j.pool.emplace_back([&f, &err, t, c, n]() noexcept {
						try {
							f(t, n * t / c, n * (t + 1) / c);
						} catch (...) {
							err.keep();
						}
					});
				f(size_t(0), size_t(0), n / c);
			} catch (...) {
				err.keep();
			}
		} // (joined)
		if (err.e)
			std::rethrow_exception(err.e);
	}
	// run f over a list of independent tasks (dynamic: the tasks differ in size)
	template <class F>
	void tasks(size_t nTasks, F&& f) const {
		const size_t c = std::max<size_t>(1, std::min<size_t>(static_cast<size_t>(std::max(threads, 1)), nTasks));
		std::atomic<size_t> next{ 0 };
		FirstError err;
		auto body = [&]() noexcept {
			try {
				for (size_t i = next.fetch_add(1); i < nTasks; i = next.fetch_add(1))
					f(i);
			} catch (...) {
				err.keep();
				next.store(nTasks); // (the others stop at their next draw)
			}
		};
		{
			Joined j;
			try {
				j.pool.reserve(c - 1);
				for (size_t t = 1; t < c; ++t)
					j.pool.emplace_back(body);
			} catch (...) {
				err.keep();
			}
			body();
		} // (joined)
		if (err.e)
			std::rethrow_exception(err.e);
	}
};
constexpr size_t kGrain = 1 << 14;

void pair_write(float* q, const tyr_bbox& l, const tyr_bbox& r, uint32_t lref, uint32_t rref, uint32_t axis) {
	for (int k = 0; k < 3; ++k) {
		q[4 * k + 0] = l.bounds[0][k];
		q[4 * k + 1] = l.bounds[1][k];
		q[4 * k + 2] = r.bounds[0][k];
		q[4 * k + 3] = r.bounds[1][k];
	}
	q[12] = bits(lref);
	q[13] = bits(rref);
	q[14] = bits(axis);
	q[15] = 0.0f;
}
//   q0 = {s0.min.x, s0.max.x, s1.min.x, s1.max.x}   q1 = {s2.., s3..}   q2,q3 = y   q4,q5 = z
//   q6 = {ref0, ref1, ref2, ref3}                   q7 = {axisTop | axisL << 2 | axisR << 4 | synthetic << 6, 0, 0, 0}
// (q7 is not fetched by the kernels: the same bits ride in every reference TO the node)
void quad_write(float* q, const tyr_bbox* boxes /*4*/, const uint32_t* refs /*4*/, uint32_t meta) {
	for (int k = 0; k < 3; ++k)
		for (int sidx = 0; sidx < 4; ++sidx) {
			q[8 * k + 2 * sidx + 0] = boxes[sidx].bounds[0][k];
			q[8 * k + 2 * sidx + 1] = boxes[sidx].bounds[1][k];
		}
	for (int sidx = 0; sidx < 4; ++sidx)
		q[24 + sidx] = bits(refs[sidx]);
	q[28] = bits(meta);
	q[29] = q[30] = q[31] = 0.0f;
}

// a leaf longer than kMaxLeafPrims (identical centroids, bvh.cpp:103): where its records start in the two layouts
struct LongLeaf {
	int32_t node;       // index of the leaf in the reference's array
	uint32_t pairHead;  // reference of the head of its chain of synthetic pair nodes
	uint32_t quadHead;  // (old) index of the head of its chain of synthetic quad records
};
inline uint32_t leaf_chunks(uint32_t cnt) { return (cnt + kMaxLeafPrims - 1) / kMaxLeafPrims; }
inline uint32_t quads_of_chain(uint32_t chunks) { return chunks <= 4 ? 1u : 1u + (chunks - 4 + 2) / 3; } // the last record holds up to 4 chunks, the others 3 + a link

struct Range { // the subtree of `begin`: nodes [begin, end)
	int32_t begin, end;
	uint8_t parity; // depth of `begin` mod 2
};

} // namespace

int build_device_layout(const tyr_bvh_node* nodes, int32_t nNodes, const tyr_triangle* prims, int32_t nPrims, DeviceLayout& L, bool wantPairs) {
	L.pairNodes.clear();
	L.quadNodes.clear();
	L.tris.clear();
	L.nPairs = 0;
	L.nQuads = 0;
	L.nStaged = 0;
	L.quadMaxStack = 0;
	L.rootRef = kRefDone;
	L.quadRootRef = kRefDone;
	for (int k = 0; k < 3; ++k) {
		L.rootMin[k] = 0.0f;
		L.rootMax[k] = 0.0f;
	}
	if (nPrims <= 0 || nNodes <= 0)
		return TYR_OK; // Scene.cpp:49-52: empty scene, no BVH
	if (!nodes || !prims || static_cast<uint32_t>(nPrims) > kMaxPrimOffset)
		return TYR_ERR_INVALID;
	const Par par{ build_threads() };
	const size_t nN = static_cast<size_t>(nNodes);
	std::atomic<bool> bad{ false };

	// ---- triangles: 40-byte records -> 3 x dwordx4 ----
	L.tris.resize(static_cast<size_t>(nPrims) * 12);
	par.run(static_cast<size_t>(nPrims), kGrain, [&](size_t, size_t b, size_t e) {
		for (size_t i = b; i < e; ++i) {
			const tyr_triangle& t = prims[i];
			if (!finite3(t.vert) || !finite3(t.e1) || !finite3(t.e2)) {
				bad = true;
				return;
			}
			float* q = &L.tris[i * 12];
			q[0] = t.vert[0];
			q[1] = t.vert[1];
			q[2] = t.vert[2];
			q[3] = t.e1[0];
			q[4] = t.e1[1];
			q[5] = t.e1[2];
			q[6] = t.e2[0];
			q[7] = t.e2[1];
			q[8] = t.e2[2];
			q[9] = bits(static_cast<uint32_t>(t.materialType));
			q[10] = bits(static_cast<uint32_t>(t.pad_[0])); // TYR_FLAG_TRIANGLE_COLORS: palette index (the reference leaves the byte unused)
			q[11] = 0.0f;
		}
	});
	if (bad)
		return TYR_ERR_INVALID;

	// ---- validate every node on its own; count the interior ones per chunk ----
	const size_t nc = par.chunks(nN, kGrain);
	std::vector<uint32_t> chunkInterior(nc + 1, 0);
	par.run(nN, kGrain, [&](size_t t, size_t b, size_t e) {
		uint32_t cnt = 0;
		for (size_t i = b; i < e; ++i) {
			const tyr_bvh_node& n = nodes[i];
			if (!finite3(n.bbox.bounds[0]) || !finite3(n.bbox.bounds[1])) {
				bad = true;
				return;
			}
			if (n.primitiveCount > 0) {
				if (n.offset < 0 || static_cast<int64_t>(n.offset) + n.primitiveCount > nPrims) {
					bad = true;
					return;
				}
			} else {
				if (n.splitAxis > 2 || static_cast<int64_t>(n.offset) <= static_cast<int64_t>(i) + 1 || n.offset >= nNodes || i + 1 >= nN) {
					bad = true;
					return;
				}
				++cnt;
			}
		}
		chunkInterior[t + 1] = cnt;
	});
	if (bad)
		return TYR_ERR_INVALID;
	for (size_t t = 0; t < nc; ++t)
		chunkInterior[t + 1] += chunkInterior[t];
	const uint32_t nInterior = chunkInterior[nc];

	// ---- the cut: the top of the tree down to subtrees of at most `cutGrain` nodes.  A node's subtree is [node, end):
	// end = the parent's second child for a first child, the parent's end for a second child (depth-first layout) ----
	std::vector<Range> ranges;
	std::vector<Range> top; // the nodes above the cut, in array order (begin = the node; end = its subtree's end)
	{
		const int32_t cutGrain = static_cast<int32_t>(std::max<size_t>(4096, nN / (static_cast<size_t>(std::max(par.threads, 1)) * 8)));
		std::vector<Range> work{ Range{ 0, nNodes, 0 } };
		while (!work.empty()) {
			const Range r = work.back();
			work.pop_back();
			if (par.threads <= 1 || r.end - r.begin <= cutGrain || nodes[r.begin].primitiveCount > 0) {
				ranges.push_back(r);
				continue;
			}
			const int32_t second = nodes[r.begin].offset;
			if (second >= r.end) // the second child lies outside its parent's subtree: not a depth-first array
				return TYR_ERR_INVALID;
			top.push_back(r);
			work.push_back(Range{ second, r.end, static_cast<uint8_t>(r.parity ^ 1u) });
			work.push_back(Range{ r.begin + 1, second, static_cast<uint8_t>(r.parity ^ 1u) });
		}
		std::sort(top.begin(), top.end(), [](const Range& a, const Range& b) { return a.begin < b.begin; });
		std::sort(ranges.begin(), ranges.end(), [](const Range& a, const Range& b) { return a.begin < b.begin; });
	}
	// depth parity of every node, and with it the check that makes the ranges independent: both children of a node lie inside
	// the node's range, every node but the root is somebody's child exactly once
	std::vector<uint8_t> parity(nN, 0), seen(nN, 0);
	seen[0] = 1;
	for (const Range& r : top) {
		parity[static_cast<size_t>(r.begin)] = r.parity;
		const int32_t c[2] = { r.begin + 1, nodes[r.begin].offset };
		for (int32_t ci : c) {
			if (seen[static_cast<size_t>(ci)])
				return TYR_ERR_INVALID;
			seen[static_cast<size_t>(ci)] = 1;
		}
	}
	par.tasks(ranges.size(), [&](size_t k) {
		const Range& r = ranges[k];
		parity[static_cast<size_t>(r.begin)] = r.parity;
		for (int32_t i = r.begin; i < r.end; ++i) {
			if (!seen[static_cast<size_t>(i)]) { // (every node of a range has its parent in front of it in the range; the range's root was marked by the cut)
				bad = true;
				return;
			}
			const tyr_bvh_node& n = nodes[i];
			if (n.primitiveCount > 0)
				continue;
			const int32_t c[2] = { i + 1, n.offset };
			if (c[1] >= r.end) {
				bad = true;
				return;
			}
			for (int32_t ci : c) {
				if (seen[static_cast<size_t>(ci)]) {
					bad = true;
					return;
				}
				seen[static_cast<size_t>(ci)] = 1;
				parity[static_cast<size_t>(ci)] = parity[static_cast<size_t>(i)] ^ 1u;
			}
		}
	});
	if (bad)
		return TYR_ERR_INVALID;

	// ---- numberings: interior node -> pair index (array order); interior node at even depth -> quad index (array order) ----
	// (not value-initialised: the passes below write every entry, in parallel -- a serial fill of C5's 12.6 M entries costs as much as a pass)
	const std::unique_ptr<uint32_t[]> pairIndex(wantPairs ? new uint32_t[nN] : nullptr), quadIndex(new uint32_t[nN]);
	std::vector<uint32_t> chunkQuads(nc + 1, 0);
	par.run(nN, kGrain, [&](size_t t, size_t b, size_t e) {
		uint32_t p = chunkInterior[t], q = 0;
		for (size_t i = b; i < e; ++i) {
			const bool interior = nodes[i].primitiveCount == 0;
			if (wantPairs)
				pairIndex[i] = interior ? p++ : 0xFFFFFFFFu;
			if (interior && !parity[i])
				++q;
		}
		chunkQuads[t + 1] = q;
	});
	for (size_t t = 0; t < nc; ++t)
		chunkQuads[t + 1] += chunkQuads[t];
	const uint32_t nRealQuads = chunkQuads[nc];
	par.run(nN, kGrain, [&](size_t t, size_t b, size_t e) {
		uint32_t q = chunkQuads[t];
		for (size_t i = b; i < e; ++i)
			quadIndex[i] = (nodes[i].primitiveCount == 0 && !parity[i]) ? q++ : 0xFFFFFFFFu;
	});

	// ---- over-long leaves (rare): in the order the serial layout met them -- pair layout: by parent, first child first; quad
	// layout: by the quad record that refers to them, slot by slot -- each gets a chain of synthetic records behind the real ones ----
	std::vector<LongLeaf> longLeaves; // sorted by node
	std::vector<int32_t> pairOrder, quadOrder; // indices into longLeaves' nodes, in emission order
	{
		std::vector<std::vector<int32_t>> found(nc);
		par.run(nN, kGrain, [&](size_t t, size_t b, size_t e) {
			for (size_t i = b; i < e; ++i)
				if (nodes[i].primitiveCount > kMaxLeafPrims)
					found[t].push_back(static_cast<int32_t>(i));
		});
		for (const auto& f : found)
			for (int32_t i : f)
				longLeaves.push_back(LongLeaf{ i, 0u, 0u });
	}
	auto long_leaf = [&](int32_t node) -> LongLeaf& {
		return *std::lower_bound(longLeaves.begin(), longLeaves.end(), node, [](const LongLeaf& l, int32_t n) { return l.node < n; });
	};
	if (!longLeaves.empty()) {
		// parent of every long leaf: the one interior node that names it (i + 1 == leaf, or offset == leaf).  Found by a scan of the
		// interior nodes (long leaves are a handful; the scan is one compare per node against a sorted list)
		std::vector<int32_t> parentOf(longLeaves.size(), -1);
		std::vector<uint8_t> isSecond(longLeaves.size(), 0);
		if (nNodes > 1) {
			std::vector<std::vector<std::pair<size_t, std::pair<int32_t, uint8_t>>>> hits(nc);
			par.run(nN, kGrain, [&](size_t t, size_t b, size_t e) {
				for (size_t i = b; i < e; ++i) {
					if (nodes[i].primitiveCount > 0)
						continue;
					const int32_t c[2] = { static_cast<int32_t>(i) + 1, nodes[i].offset };
					for (int s = 0; s < 2; ++s) {
						if (nodes[c[s]].primitiveCount <= kMaxLeafPrims)
							continue;
						const size_t k = static_cast<size_t>(&long_leaf(c[s]) - longLeaves.data());
						hits[t].push_back({ k, { static_cast<int32_t>(i), static_cast<uint8_t>(s) } });
					}
				}
			});
			for (const auto& h : hits)
				for (const auto& x : h) {
					parentOf[x.first] = x.second.first;
					isSecond[x.first] = x.second.second;
				}
		}
		// pair layout: emission order = (parent, which child); a leaf that IS the tree has no parent and is emitted by child_ref(0)
		std::vector<size_t> order(longLeaves.size());
		for (size_t k = 0; k < order.size(); ++k)
			order[k] = k;
		std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return parentOf[a] != parentOf[b] ? parentOf[a] < parentOf[b] : isSecond[a] < isSecond[b]; });
		uint32_t nextPair = nInterior;
		for (size_t k : order) {
			const uint32_t chunks = leaf_chunks(nodes[longLeaves[k].node].primitiveCount);
			longLeaves[k].pairHead = nextPair + chunks - 2; // chunks - 1 records, the head is the last one written
			nextPair += chunks - 1;
			pairOrder.push_back(static_cast<int32_t>(k));
		}
		L.nPairs = nextPair;
		// quad layout: emission order = (the quad record that refers to the leaf, slot).  That record is the parent when the parent
		// sits at even depth (slot 0 / 2), else the grandparent (the parent is its first or second child: slots 0,1 / 2,3)
		std::vector<std::pair<std::pair<int32_t, int32_t>, size_t>> qkey(longLeaves.size());
		std::vector<int32_t> parentOfNode;
		if (nNodes > 1) {
			// the grandparent: the parent's parent.  Only needed for long leaves whose parent sits at odd depth; found by one more scan
			std::vector<int32_t> want;
			for (size_t k = 0; k < longLeaves.size(); ++k)
				if (parentOf[k] >= 0 && parity[static_cast<size_t>(parentOf[k])])
					want.push_back(parentOf[k]);
			std::sort(want.begin(), want.end());
			want.erase(std::unique(want.begin(), want.end()), want.end());
			parentOfNode.assign(want.size(), -1);
			std::vector<uint8_t> second(want.size(), 0);
			if (!want.empty()) {
				std::vector<std::vector<std::pair<size_t, std::pair<int32_t, uint8_t>>>> hits(nc);
				par.run(nN, kGrain, [&](size_t t, size_t b, size_t e) {
					for (size_t i = b; i < e; ++i) {
						if (nodes[i].primitiveCount > 0)
							continue;
						const int32_t c[2] = { static_cast<int32_t>(i) + 1, nodes[i].offset };
						for (int s = 0; s < 2; ++s) {
							const auto it = std::lower_bound(want.begin(), want.end(), c[s]);
							if (it != want.end() && *it == c[s])
								hits[t].push_back({ static_cast<size_t>(it - want.begin()), { static_cast<int32_t>(i), static_cast<uint8_t>(s) } });
						}
					}
				});
				for (const auto& h : hits)
					for (const auto& x : h) {
						parentOfNode[x.first] = x.second.first;
						second[x.first] = x.second.second;
					}
			}
			for (size_t k = 0; k < longLeaves.size(); ++k) {
				const int32_t p = parentOf[k];
				if (p < 0) {
					qkey[k] = { { -1, 0 }, k };
				} else if (!parity[static_cast<size_t>(p)]) {
					qkey[k] = { { p, 2 * isSecond[k] }, k };
				} else {
					const size_t w = static_cast<size_t>(std::lower_bound(want.begin(), want.end(), p) - want.begin());
					qkey[k] = { { parentOfNode[w], 2 * second[w] + isSecond[k] }, k };
				}
			}
		} else {
			qkey[0] = { { -1, 0 }, 0 };
		}
		std::sort(qkey.begin(), qkey.end());
		uint32_t nextQuad = nRealQuads;
		for (const auto& qk : qkey) {
			const size_t k = qk.second;
			const uint32_t nq = quads_of_chain(leaf_chunks(nodes[longLeaves[k].node].primitiveCount));
			longLeaves[k].quadHead = nextQuad + nq - 1; // the head is the last record written
			nextQuad += nq;
			quadOrder.push_back(static_cast<int32_t>(k));
		}
		L.nQuads = nextQuad;
	} else {
		L.nPairs = nInterior;
		L.nQuads = nRealQuads;
	}
	const uint32_t nQuads = L.nQuads;
	if (nQuads > kQuadIndexMask)
		return TYR_ERR_INVALID;

	// ---- pair nodes (the counting build and the BVH_DEBUG picture only) ----
	// reference of a child node; leaves longer than kMaxLeafPrims become a chain of synthetic "left first" pair nodes (axis 3)
	// that visits the same primitives in the same order
	auto child_ref = [&](int32_t ci) -> uint32_t {
		const tyr_bvh_node& c = nodes[ci];
		if (c.primitiveCount == 0)
			return wantPairs ? pairIndex[static_cast<size_t>(ci)] : 0u; // (without the pair layout only the root is ever asked for: pair 0)
		if (c.primitiveCount <= kMaxLeafPrims)
			return leaf_ref(static_cast<uint32_t>(c.offset), c.primitiveCount);
		return long_leaf(ci).pairHead;
	};
	std::memcpy(L.rootMin, nodes[0].bbox.bounds[0], 12);
	std::memcpy(L.rootMax, nodes[0].bbox.bounds[1], 12);
	L.rootRef = child_ref(0);
	if (wantPairs) {
		L.pairNodes.resize(static_cast<size_t>(L.nPairs) * 16);
		par.run(nN, kGrain, [&](size_t, size_t b, size_t e) {
			for (size_t i = b; i < e; ++i) {
				const tyr_bvh_node& n = nodes[i];
				if (n.primitiveCount > 0)
					continue;
				const int32_t li = static_cast<int32_t>(i) + 1, ri = n.offset;
				pair_write(&L.pairNodes[static_cast<size_t>(pairIndex[i]) * 16], nodes[li].bbox, nodes[ri].bbox, child_ref(li), child_ref(ri), n.splitAxis);
			}
		});
		for (int32_t k : pairOrder) { // the chains, back to front
			const LongLeaf& ll = longLeaves[static_cast<size_t>(k)];
			const tyr_bvh_node& c = nodes[ll.node];
			const uint32_t off = static_cast<uint32_t>(c.offset), cnt = c.primitiveCount, chunks = leaf_chunks(cnt);
			uint32_t ref = leaf_ref(off + (chunks - 1) * kMaxLeafPrims, cnt - (chunks - 1) * kMaxLeafPrims);
			uint32_t idx = ll.pairHead - (chunks - 2);
			for (uint32_t j = chunks - 1; j-- > 0; ++idx) {
				pair_write(&L.pairNodes[static_cast<size_t>(idx) * 16], c.bbox, c.bbox, leaf_ref(off + j * kMaxLeafPrims, kMaxLeafPrims), ref, 3u);
				ref = idx;
			}
		}
	} else {
		L.nPairs = 0;
	}

	// ---- quad nodes: a node together with both its children, 128 B ------------------------------
	// Slots 0,1 = the children of the node's LEFT child (or the left child itself in slot 0 when it is
	// a leaf), slots 2,3 = the same for the RIGHT child.  Visit order is decided by three split axes:
	// the node's (which group first) and each interior child's (which slot of the group first).
	constexpr uint32_t kEmptyRef = kRefDone; // an unused slot
	if (nodes[0].primitiveCount > 0 && nodes[0].primitiveCount <= kMaxLeafPrims) {
		L.quadRootRef = L.rootRef; // single-leaf tree: the leaf reference itself
		return TYR_OK;
	}
	// an empty slot: both planes of every axis at +infinity, and a reference that marks it unused.  No ray passes the
	// box tests on it (hip/traverse.hpp): a positive 1/d makes the entry distance +inf (never < the bound), a
	// negative one makes the exit distance -inf (never > 0), infinite 1/d likewise, and inf * x is never NaN for
	// x != 0 -- so the kernels do not spend four compares per node on "is this slot used".
	tyr_bbox emptyBox, everythingBox;
	for (int k = 0; k < 3; ++k) {
		emptyBox.bounds[0][k] = std::numeric_limits<float>::infinity();
		emptyBox.bounds[1][k] = std::numeric_limits<float>::infinity();
		// a used slot of a synthetic record (consecutive chunks of one over-long leaf, visited in slot order: bvh.h:131 -- the reference
		// tests a leaf's box once, at its parent, and then every primitive): (-inf, +inf) on every axis, which every ray enters at -inf
		everythingBox.bounds[0][k] = -std::numeric_limits<float>::infinity();
		everythingBox.bounds[1][k] = std::numeric_limits<float>::infinity();
	}
	// visit-order bits of a quad record, as they ride in every reference to it (bits 25..30): axisTop | axisL << 2 | axisR << 4;
	// a synthetic record: all three axis codes 3 = "swap nothing"
	auto real_meta = [&](int32_t pi) -> uint32_t {
		const tyr_bvh_node& P = nodes[pi];
		const tyr_bvh_node &X0 = nodes[pi + 1], &X1 = nodes[P.offset];
		return static_cast<uint32_t>(P.splitAxis) | ((X0.primitiveCount > 0 ? 0u : X0.splitAxis) << 2) | ((X1.primitiveCount > 0 ? 0u : X1.splitAxis) << 4);
	};
	// The top of the tree moves to the front of the array in breadth-first order (the persistent kernels keep the first
	// kStagedNodes records in LDS, hip/traverse.hpp); everything else keeps its order.  Old index -> new index.
	// children (old quad indices / leaf references) of an old record, without the array: from the reference's nodes
	struct Slots {
		uint32_t ref[4]; // old quad index, leaf reference, or kEmptyRef
		int32_t node[4]; // the reference-tree node behind the slot (-1: none)
	};
	auto leaf_slot_ref = [&](int32_t ci) -> uint32_t { // a leaf child: its reference, or the head of its synthetic chain
		const tyr_bvh_node& c = nodes[ci];
		return c.primitiveCount <= kMaxLeafPrims ? leaf_ref(static_cast<uint32_t>(c.offset), c.primitiveCount) : long_leaf(ci).quadHead;
	};
	auto real_slots = [&](int32_t pi) {
		Slots s{ { kEmptyRef, kEmptyRef, kEmptyRef, kEmptyRef }, { -1, -1, -1, -1 } };
		const int32_t kids[2] = { pi + 1, nodes[pi].offset };
		for (int g = 0; g < 2; ++g) {
			const tyr_bvh_node& X = nodes[kids[g]];
			if (X.primitiveCount > 0) {
				s.ref[2 * g] = leaf_slot_ref(kids[g]);
				s.node[2 * g] = kids[g];
			} else {
				const int32_t gk[2] = { kids[g] + 1, X.offset };
				for (int h = 0; h < 2; ++h) {
					s.ref[2 * g + h] = nodes[gk[h]].primitiveCount > 0 ? leaf_slot_ref(gk[h]) : quadIndex[static_cast<size_t>(gk[h])];
					s.node[2 * g + h] = gk[h];
				}
			}
		}
		return s;
	};
	// synthetic records: chunks back to front -- the last record of a chain holds up to 4 chunks, earlier ones 3 chunks + a link
	struct Synth {
		uint32_t refs[4];
		uint32_t used;
	};
	std::vector<Synth> synth(nQuads - nRealQuads);
	for (int32_t k : quadOrder) {
		const LongLeaf& ll = longLeaves[static_cast<size_t>(k)];
		const tyr_bvh_node& c = nodes[ll.node];
		const uint32_t off = static_cast<uint32_t>(c.offset), cnt = c.primitiveCount;
		std::vector<uint32_t> chunkRefs;
		for (uint32_t o = 0; o < cnt; o += kMaxLeafPrims)
			chunkRefs.push_back(leaf_ref(off + o, std::min<uint32_t>(kMaxLeafPrims, cnt - o)));
		uint32_t qi = ll.quadHead - (quads_of_chain(static_cast<uint32_t>(chunkRefs.size())) - 1), link = kEmptyRef;
		bool haveLink = false;
		size_t end = chunkRefs.size();
		while (end > 0) {
			const size_t room = haveLink ? 3 : 4;
			const size_t begin = end > room ? end - room : 0;
			Synth s{ { kEmptyRef, kEmptyRef, kEmptyRef, kEmptyRef }, 0 };
			for (size_t i = begin; i < end; ++i)
				s.refs[s.used++] = chunkRefs[i];
			if (haveLink)
				s.refs[s.used++] = link;
			synth[qi - nRealQuads] = s;
			link = qi++;
			haveLink = true;
			end = begin;
		}
	}
	const uint32_t oldRoot = nodes[0].primitiveCount > 0 ? long_leaf(0).quadHead : 0u;
	// old index of a real record -> the reference-tree node it was made from (only needed for the breadth-first top: a search)
	auto node_of_quad = [&](uint32_t qi) -> int32_t {
		// quadIndex is increasing along the array: the chunk that holds qi, then a scan
		size_t t = static_cast<size_t>(std::upper_bound(chunkQuads.begin(), chunkQuads.end(), qi) - chunkQuads.begin()) - 1;
		const size_t b = nN * t / nc, e = nN * (t + 1) / nc;
		for (size_t i = b; i < e; ++i)
			if (quadIndex[i] == qi)
				return static_cast<int32_t>(i);
		return -1;
	};
	std::vector<uint32_t> topOld{ oldRoot };
	for (size_t head = 0; head < topOld.size() && topOld.size() < kStagedNodes; ++head) {
		const uint32_t q = topOld[head];
		uint32_t refs[4];
		if (q < nRealQuads) {
			const Slots s = real_slots(node_of_quad(q));
			std::memcpy(refs, s.ref, sizeof refs);
		} else {
			std::memcpy(refs, synth[q - nRealQuads].refs, sizeof refs);
		}
		for (int sidx = 0; sidx < 4 && topOld.size() < kStagedNodes; ++sidx)
			if (static_cast<int32_t>(refs[sidx]) >= 0)
				topOld.push_back(refs[sidx]);
	}
	const uint32_t nTop = static_cast<uint32_t>(topOld.size());
	std::vector<std::pair<uint32_t, uint32_t>> topSorted(nTop); // (old index, new index)
	for (uint32_t i = 0; i < nTop; ++i)
		topSorted[i] = { topOld[i], i };
	std::sort(topSorted.begin(), topSorted.end());
	auto new_index = [&](uint32_t old) -> uint32_t {
		const auto it = std::lower_bound(topSorted.begin(), topSorted.end(), std::make_pair(old, 0u));
		if (it != topSorted.end() && it->first == old)
			return it->second;
		return nTop + old - static_cast<uint32_t>(it - topSorted.begin()); // the records that are not part of the top keep their order behind it
	};
	// a slot's final reference: leaf references as they are; a record's new index with its visit-order bits
	auto final_ref = [&](uint32_t ref, int32_t node) -> uint32_t {
		if (static_cast<int32_t>(ref) < 0)
			return ref; // leaf, unused slot
		const uint32_t order = ref < nRealQuads ? (real_meta(node) & 63u) : 63u;
		return new_index(ref) | (order << kQuadOrderShift);
	};

	L.quadNodes.resize(static_cast<size_t>(nQuads) * 32);
	par.run(nN, kGrain, [&](size_t, size_t b, size_t e) {
		for (size_t i = b; i < e; ++i) {
			const uint32_t qi = quadIndex[i];
			if (qi == 0xFFFFFFFFu)
				continue;
			const int32_t pi = static_cast<int32_t>(i);
			const Slots s = real_slots(pi);
			uint32_t refs[4];
			tyr_bbox boxes[4];
			for (int k = 0; k < 4; ++k) {
				refs[k] = final_ref(s.ref[k], s.node[k]);
				boxes[k] = s.node[k] >= 0 ? nodes[s.node[k]].bbox : emptyBox;
			}
			quad_write(&L.quadNodes[static_cast<size_t>(new_index(qi)) * 32], boxes, refs, real_meta(pi));
		}
	});
	for (uint32_t k = 0; k < nQuads - nRealQuads; ++k) {
		const Synth& s = synth[k];
		uint32_t refs[4];
		tyr_bbox boxes[4];
		for (uint32_t j = 0; j < 4; ++j) {
			refs[j] = j < s.used ? final_ref(s.refs[j], -1) : kEmptyRef;
			boxes[j] = j < s.used ? everythingBox : emptyBox;
		}
		quad_write(&L.quadNodes[static_cast<size_t>(new_index(nRealQuads + k)) * 32], boxes, refs, 1u << 6);
	}
	L.quadRootRef = final_ref(oldRoot, 0);
	L.nStaged = nTop;

	// ---- the deepest a traversal's stack can get on this tree, for any ray and any visit order: inside the subtree of one child
	// of a record, at most the record's other used slots wait on the stack -- need(q) = (used slots - 1) + max over the children
	// that are records of need(child).  The kernels' four-lanes-to-a-ray drain keeps a ray's stack in 48 LDS entries and is only
	// entered on trees that cannot need more.  Children lie behind their parents in the array (synthetic chains: in front of their
	// head), so one sweep per subtree range, back to front, then the records above the cut ----
	{
		std::vector<uint32_t> need(nQuads, 0);
		for (uint32_t k = 0; k < nQuads - nRealQuads; ++k) { // a chain's records refer to leaves and to the record written before them
			const Synth& s = synth[k];
			uint32_t deepest = 0;
			for (uint32_t j = 0; j < s.used; ++j)
				if (static_cast<int32_t>(s.refs[j]) >= 0)
					deepest = std::max(deepest, need[s.refs[j]]);
			need[nRealQuads + k] = (s.used ? s.used - 1 : 0) + deepest;
		}
		auto need_of = [&](int32_t pi) {
			const Slots s = real_slots(pi);
			uint32_t used = 0, deepest = 0;
			for (int k = 0; k < 4; ++k) {
				if (s.ref[k] == kEmptyRef)
					continue;
				++used;
				if (static_cast<int32_t>(s.ref[k]) >= 0)
					deepest = std::max(deepest, need[s.ref[k]]);
			}
			need[quadIndex[static_cast<size_t>(pi)]] = (used ? used - 1 : 0) + deepest;
		};
		par.tasks(ranges.size(), [&](size_t k) {
			const Range& r = ranges[k];
			for (int32_t i = r.end; i-- > r.begin;)
				if (quadIndex[static_cast<size_t>(i)] != 0xFFFFFFFFu)
					need_of(i);
		});
		for (size_t k = top.size(); k-- > 0;)
			if (quadIndex[static_cast<size_t>(top[k].begin)] != 0xFFFFFFFFu)
				need_of(top[k].begin);
		L.quadMaxStack = need[oldRoot];
	}
	return TYR_OK;
}

} // namespace tyr
