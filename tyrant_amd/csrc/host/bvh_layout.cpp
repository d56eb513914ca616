// bvh_layout.cpp -- the upload half of Scene::Load (Scene.cpp:55-67) for gfx950: turns the
// reference's flat depth-first 32-byte node array (bvh.h:55-68) into the private device
// layout documented in hip/traverse.hpp (64-byte child-pair nodes, 48-byte triangles).
// The tree itself -- which primitive is in which leaf, which child is "first", every box --
// is unchanged, so the traversal visit order of bvh.h:118-161 is preserved.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

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

struct Emit {
	std::vector<float>& out;
	// appends one pair node, returns its index
	uint32_t pair(const tyr_bbox& l, const tyr_bbox& r, uint32_t lref, uint32_t rref, uint32_t axis) {
		const uint32_t idx = static_cast<uint32_t>(out.size() / 16);
		out.resize(out.size() + 16);
		write(idx, l, r, lref, rref, axis);
		return idx;
	}
	void write(uint32_t idx, const tyr_bbox& l, const tyr_bbox& r, uint32_t lref, uint32_t rref, uint32_t axis) {
		float* q = &out[static_cast<size_t>(idx) * 16];
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
};

} // namespace

int build_device_layout(const tyr_bvh_node* nodes, int32_t nNodes, const tyr_triangle* prims, int32_t nPrims, DeviceLayout& L) {
	L.pairNodes.clear();
	L.tris.clear();
	L.nPairs = 0;
	L.rootRef = kRefDone;
	for (int k = 0; k < 3; ++k) {
		L.rootMin[k] = 0.0f;
		L.rootMax[k] = 0.0f;
	}
	if (nPrims <= 0 || nNodes <= 0)
		return TYR_OK; // Scene.cpp:49-52: empty scene, no BVH
	if (!nodes || !prims || static_cast<uint32_t>(nPrims) > kMaxPrimOffset)
		return TYR_ERR_INVALID;

	// triangles
	L.tris.resize(static_cast<size_t>(nPrims) * 12);
	for (int32_t i = 0; i < nPrims; ++i) {
		const tyr_triangle& t = prims[i];
		if (!finite3(t.vert) || !finite3(t.e1) || !finite3(t.e2))
			return TYR_ERR_INVALID;
		float* q = &L.tris[static_cast<size_t>(i) * 12];
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

	// pass 1: validate, and number the interior nodes in depth-first (array) order
	std::vector<uint32_t> pairIndex(static_cast<size_t>(nNodes), 0xFFFFFFFFu);
	uint32_t nInterior = 0;
	for (int32_t i = 0; i < nNodes; ++i) {
		const tyr_bvh_node& n = nodes[i];
		if (!finite3(n.bbox.bounds[0]) || !finite3(n.bbox.bounds[1]))
			return TYR_ERR_INVALID;
		if (n.primitiveCount > 0) {
			if (n.offset < 0 || static_cast<int64_t>(n.offset) + n.primitiveCount > nPrims)
				return TYR_ERR_INVALID;
		} else {
			if (n.splitAxis > 2 || n.offset <= i + 1 || n.offset >= nNodes || i + 1 >= nNodes)
				return TYR_ERR_INVALID;
			pairIndex[static_cast<size_t>(i)] = nInterior++;
		}
	}
	// every node except the root must be referenced exactly once (a depth-first tree)
	{
		std::vector<uint8_t> seen(static_cast<size_t>(nNodes), 0);
		seen[0] = 1;
		for (int32_t i = 0; i < nNodes; ++i) {
			if (nodes[i].primitiveCount == 0) {
				const int32_t c[2] = { i + 1, nodes[i].offset };
				for (int32_t ci : c) {
					if (seen[static_cast<size_t>(ci)])
						return TYR_ERR_INVALID;
					seen[static_cast<size_t>(ci)] = 1;
				}
			}
		}
		for (int32_t i = 0; i < nNodes; ++i)
			if (!seen[static_cast<size_t>(i)])
				return TYR_ERR_INVALID;
	}

	L.pairNodes.assign(static_cast<size_t>(nInterior) * 16, 0.0f);
	Emit emit{ L.pairNodes };

	// reference of a child node; leaves longer than kMaxLeafPrims become a chain of synthetic
	// "left first" pair nodes (axis 3) that visits the same primitives in the same order
	auto child_ref = [&](int32_t ci) -> uint32_t {
		const tyr_bvh_node& c = nodes[ci];
		if (c.primitiveCount == 0)
			return pairIndex[static_cast<size_t>(ci)];
		uint32_t off = static_cast<uint32_t>(c.offset), cnt = c.primitiveCount;
		if (cnt <= kMaxLeafPrims)
			return leaf_ref(off, cnt);
		// build the chain back to front
		const uint32_t chunks = (cnt + kMaxLeafPrims - 1) / kMaxLeafPrims;
		uint32_t tailOff = off + (chunks - 1) * kMaxLeafPrims;
		uint32_t ref = leaf_ref(tailOff, cnt - (chunks - 1) * kMaxLeafPrims);
		for (uint32_t k = chunks - 1; k-- > 0;) {
			const uint32_t o = off + k * kMaxLeafPrims;
			ref = emit.pair(c.bbox, c.bbox, leaf_ref(o, kMaxLeafPrims), ref, 3u);
		}
		return ref;
	};

	for (int32_t i = 0; i < nNodes; ++i) {
		const tyr_bvh_node& n = nodes[i];
		if (n.primitiveCount > 0)
			continue;
		const int32_t li = i + 1, ri = n.offset;
		const uint32_t lref = child_ref(li), rref = child_ref(ri);
		emit.write(pairIndex[static_cast<size_t>(i)], nodes[li].bbox, nodes[ri].bbox, lref, rref, n.splitAxis);
	}
	L.nPairs = static_cast<uint32_t>(L.pairNodes.size() / 16);
	std::memcpy(L.rootMin, nodes[0].bbox.bounds[0], 12);
	std::memcpy(L.rootMax, nodes[0].bbox.bounds[1], 12);
	L.rootRef = child_ref(0);
	L.nPairs = static_cast<uint32_t>(L.pairNodes.size() / 16);

	// ---- quad nodes: a node together with both its children, 128 B ------------------------------
	// Slots 0,1 = the children of the node's LEFT child (or the left child itself in slot 0 when it is
	// a leaf), slots 2,3 = the same for the RIGHT child.  Visit order is decided by three split axes:
	// the node's (which group first) and each interior child's (which slot of the group first).
	//   q0 = {s0.min.x, s0.max.x, s1.min.x, s1.max.x}   q1 = {s2.., s3..}   q2,q3 = y   q4,q5 = z
	//   q6 = {ref0, ref1, ref2, ref3}                   q7 = {axisTop | axisL << 2 | axisR << 4 | synthetic << 6, 0, 0, 0}
	// (q7 is not fetched by the kernels: the same bits ride in every reference TO this node, see the end of this function)
	L.quadNodes.clear();
	L.nQuads = 0;
	L.quadRootRef = kRefDone;
	constexpr uint32_t kEmptyRef = kRefDone; // an unused slot
	if (nodes[0].primitiveCount > 0) {
		L.quadRootRef = L.rootRef; // single-leaf tree: same leaf reference (or pair-node chain for a long leaf, see below)
	}
	std::vector<uint32_t> quadIndex(static_cast<size_t>(nNodes), 0xFFFFFFFFu);
	if (nodes[0].primitiveCount == 0) {
		// pass 1: quad roots = the root and every interior grandchild of a quad root
		std::vector<uint8_t> isRoot(static_cast<size_t>(nNodes), 0);
		std::vector<int32_t> work{ 0 };
		while (!work.empty()) {
			const int32_t pi = work.back();
			work.pop_back();
			isRoot[static_cast<size_t>(pi)] = 1;
			const int32_t kids[2] = { pi + 1, nodes[pi].offset };
			for (int32_t x : kids) {
				if (nodes[x].primitiveCount > 0)
					continue;
				const int32_t gk[2] = { x + 1, nodes[x].offset };
				for (int32_t y : gk)
					if (nodes[y].primitiveCount == 0)
						work.push_back(y);
			}
		}
		uint32_t nq = 0;
		for (int32_t i = 0; i < nNodes; ++i)
			if (isRoot[static_cast<size_t>(i)])
				quadIndex[static_cast<size_t>(i)] = nq++; // depth-first array order
		L.quadNodes.assign(static_cast<size_t>(nq) * 32, 0.0f);
	}
	auto quad_write = [&](uint32_t qi, const tyr_bbox* boxes /*4*/, const uint32_t* refs /*4*/, uint32_t meta) {
		if (L.quadNodes.size() < (static_cast<size_t>(qi) + 1) * 32)
			L.quadNodes.resize((static_cast<size_t>(qi) + 1) * 32, 0.0f);
		float* q = &L.quadNodes[static_cast<size_t>(qi) * 32];
		for (int k = 0; k < 3; ++k) {
			for (int sidx = 0; sidx < 4; ++sidx) {
				q[8 * k + 2 * sidx + 0] = boxes[sidx].bounds[0][k];
				q[8 * k + 2 * sidx + 1] = boxes[sidx].bounds[1][k];
			}
		}
		for (int sidx = 0; sidx < 4; ++sidx)
			q[24 + sidx] = bits(refs[sidx]);
		q[28] = bits(meta);
		q[29] = q[30] = q[31] = 0.0f;
	};
	// an empty slot: both planes of every axis at +infinity, and a reference that marks it unused.  No ray passes the
	// box tests on it (hip/traverse.hpp): a positive 1/d makes the entry distance +inf (never < the bound), a
	// negative one makes the exit distance -inf (never > 0), infinite 1/d likewise, and inf * x is never NaN for
	// x != 0 -- so the kernels do not spend four compares per node on "is this slot used".
	tyr_bbox emptyBox;
	for (int k = 0; k < 3; ++k) {
		emptyBox.bounds[0][k] = std::numeric_limits<float>::infinity();
		emptyBox.bounds[1][k] = std::numeric_limits<float>::infinity();
	}
	// leaf reference in the quad layout; leaves longer than kMaxLeafPrims become synthetic quads
	// (bit 6 of meta) whose slots are consecutive chunks, visited in slot order (bvh.h:131: the reference tests a
	// leaf's box once, at its parent, and then every primitive).  The kernel has no special case for them: a used
	// slot's box is (-inf, +inf) on every axis, which every ray passes with entry distance -inf (inf * x is never NaN
	// for the 1/d the kernel allows, see the empty slot above), and the order bits say "no swap" (axis code 3 three
	// times: bit 3 of a ray's sign bits is clear).
	tyr_bbox everythingBox;
	for (int k = 0; k < 3; ++k) {
		everythingBox.bounds[0][k] = -std::numeric_limits<float>::infinity();
		everythingBox.bounds[1][k] = std::numeric_limits<float>::infinity();
	}
	auto quad_leaf_ref = [&](const tyr_bvh_node& c) -> uint32_t {
		uint32_t off = static_cast<uint32_t>(c.offset), cnt = c.primitiveCount;
		if (cnt <= kMaxLeafPrims)
			return leaf_ref(off, cnt);
		// chunks back to front: the last quad of the chain holds up to 4 chunks, earlier ones 3 chunks + a link
		std::vector<uint32_t> chunkRefs;
		for (uint32_t o = 0; o < cnt; o += kMaxLeafPrims)
			chunkRefs.push_back(leaf_ref(off + o, std::min<uint32_t>(kMaxLeafPrims, cnt - o)));
		uint32_t link = kEmptyRef;
		bool haveLink = false;
		size_t end = chunkRefs.size();
		while (end > 0) {
			const size_t room = haveLink ? 3 : 4;
			const size_t begin = end > room ? end - room : 0;
			uint32_t refs[4] = { kEmptyRef, kEmptyRef, kEmptyRef, kEmptyRef };
			tyr_bbox boxes[4] = { emptyBox, emptyBox, emptyBox, emptyBox };
			size_t k = 0;
			for (size_t i = begin; i < end; ++i, ++k) {
				refs[k] = chunkRefs[i];
				boxes[k] = everythingBox;
			}
			if (haveLink) {
				refs[k] = link;
				boxes[k] = everythingBox;
			}
			const uint32_t qi = static_cast<uint32_t>(L.quadNodes.size() / 32);
			quad_write(qi, boxes, refs, 1u << 6);
			link = qi;
			haveLink = true;
			end = begin;
		}
		return link;
	};
	if (nodes[0].primitiveCount > 0) {
		L.quadRootRef = quad_leaf_ref(nodes[0]);
	} else {
		const uint32_t nRealQuads = static_cast<uint32_t>(L.quadNodes.size() / 32);
		(void)nRealQuads;
		for (int32_t pi = 0; pi < nNodes; ++pi) {
			const uint32_t qi = quadIndex[static_cast<size_t>(pi)];
			if (qi == 0xFFFFFFFFu)
				continue;
			const tyr_bvh_node& P = nodes[pi];
			uint32_t refs[4] = { kEmptyRef, kEmptyRef, kEmptyRef, kEmptyRef };
			tyr_bbox boxes[4] = { emptyBox, emptyBox, emptyBox, emptyBox };
			uint32_t axes[2] = { 0, 0 };
			const int32_t kids[2] = { pi + 1, P.offset };
			for (int g = 0; g < 2; ++g) {
				const tyr_bvh_node& X = nodes[kids[g]];
				if (X.primitiveCount > 0) {
					refs[2 * g] = quad_leaf_ref(X);
					boxes[2 * g] = X.bbox;
				} else {
					axes[g] = X.splitAxis;
					const int32_t gk[2] = { kids[g] + 1, X.offset };
					for (int h = 0; h < 2; ++h) {
						const tyr_bvh_node& Y = nodes[gk[h]];
						refs[2 * g + h] = Y.primitiveCount > 0 ? quad_leaf_ref(Y) : quadIndex[static_cast<size_t>(gk[h])];
						boxes[2 * g + h] = Y.bbox;
					}
				}
			}
			quad_write(qi, boxes, refs, static_cast<uint32_t>(P.splitAxis) | (axes[0] << 2) | (axes[1] << 4));
		}
		L.quadRootRef = 0;
	}
	L.nQuads = static_cast<uint32_t>(L.quadNodes.size() / 32);
	// The deepest a traversal's stack can get on this tree, for any ray and any visit order: inside the subtree of one child
	// of a node, at most the node's other used slots wait on the stack -- need(q) = max over interior children c of
	// (used slots of q - 1) + need(c), and (used slots - 1) for a node of leaves.  The kernels' four-lanes-to-a-ray drain
	// keeps a ray's stack in 48 LDS entries and is only entered on trees that cannot need more.
	L.quadMaxStack = 0;
	if (L.nQuads > 0 && static_cast<int32_t>(L.quadRootRef) >= 0) {
		auto ref_of = [&](uint32_t qi, int sidx) {
			uint32_t r;
			std::memcpy(&r, &L.quadNodes[static_cast<size_t>(qi) * 32 + 24 + sidx], 4);
			return r;
		};
		std::vector<uint32_t> need(L.nQuads, 0);
		std::vector<uint8_t> state(L.nQuads, 0); // 0 new, 1 children pushed, 2 done
		std::vector<uint32_t> todo{ L.quadRootRef };
		while (!todo.empty()) {
			const uint32_t q = todo.back();
			if (state[q] == 0) {
				state[q] = 1;
				for (int sidx = 0; sidx < 4; ++sidx) {
					const uint32_t r = ref_of(q, sidx);
					if (static_cast<int32_t>(r) >= 0 && r < L.nQuads && state[r] == 0)
						todo.push_back(r);
				}
			} else {
				todo.pop_back();
				if (state[q] == 2)
					continue;
				state[q] = 2;
				uint32_t used = 0, deepest = 0;
				for (int sidx = 0; sidx < 4; ++sidx) {
					const uint32_t r = ref_of(q, sidx);
					if (r == kEmptyRef)
						continue;
					++used;
					if (static_cast<int32_t>(r) >= 0 && r < L.nQuads)
						deepest = std::max(deepest, need[r]);
				}
				need[q] = (used ? used - 1 : 0) + deepest;
			}
		}
		L.quadMaxStack = need[L.quadRootRef];
	}
	// The top of the tree moves to the front of the array in breadth-first order: the persistent kernels keep the
	// first kStagedNodes records in LDS (hip/traverse.hpp).  Everything else keeps its depth-first order.
	L.nStaged = 0;
	if (L.nQuads > 0 && static_cast<int32_t>(L.quadRootRef) >= 0) {
		auto ref_at = [&](uint32_t qi, int sidx) {
			uint32_t r;
			std::memcpy(&r, &L.quadNodes[static_cast<size_t>(qi) * 32 + 24 + sidx], 4);
			return r;
		};
		std::vector<uint32_t> top{ L.quadRootRef };
		for (size_t head = 0; head < top.size() && top.size() < kStagedNodes; ++head)
			for (int sidx = 0; sidx < 4 && top.size() < kStagedNodes; ++sidx) {
				const uint32_t r = ref_at(top[head], sidx);
				if (static_cast<int32_t>(r) >= 0)
					top.push_back(r);
			}
		std::vector<uint32_t> newIndex(L.nQuads, 0xFFFFFFFFu);
		for (size_t i = 0; i < top.size(); ++i)
			newIndex[top[i]] = static_cast<uint32_t>(i);
		uint32_t next = static_cast<uint32_t>(top.size());
		for (uint32_t qi = 0; qi < L.nQuads; ++qi)
			if (newIndex[qi] == 0xFFFFFFFFu)
				newIndex[qi] = next++;
		std::vector<float> moved(L.quadNodes.size());
		for (uint32_t qi = 0; qi < L.nQuads; ++qi) {
			float* dst = &moved[static_cast<size_t>(newIndex[qi]) * 32];
			std::memcpy(dst, &L.quadNodes[static_cast<size_t>(qi) * 32], 32 * sizeof(float));
			for (int sidx = 0; sidx < 4; ++sidx) {
				uint32_t r;
				std::memcpy(&r, dst + 24 + sidx, 4);
				if (static_cast<int32_t>(r) >= 0) {
					r = newIndex[r];
					std::memcpy(dst + 24 + sidx, &r, 4);
				}
			}
		}
		L.quadNodes.swap(moved);
		L.quadRootRef = newIndex[L.quadRootRef];
		L.nStaged = static_cast<uint32_t>(top.size());
	}
	// An interior reference carries the visit-order bits of the node it points to in bits 25..30 (axisTop | axisL << 2 |
	// axisR << 4, all three 3 for a synthetic chain), so the kernel knows them before the node arrives and fetches
	// seven vectors per node instead of eight.
	if (L.nQuads > kQuadIndexMask)
		return TYR_ERR_INVALID;
	auto with_order_bits = [&](uint32_t ref) -> uint32_t {
		if (static_cast<int32_t>(ref) < 0)
			return ref; // leaf, unused slot
		uint32_t meta;
		std::memcpy(&meta, &L.quadNodes[static_cast<size_t>(ref) * 32 + 28], 4);
		const uint32_t order = (meta & 64u) ? 63u : (meta & 63u);
		return ref | (order << kQuadOrderShift);
	};
	for (uint32_t qi = 0; qi < L.nQuads; ++qi) {
		for (int sidx = 0; sidx < 4; ++sidx) {
			uint32_t ref;
			std::memcpy(&ref, &L.quadNodes[static_cast<size_t>(qi) * 32 + 24 + sidx], 4);
			ref = with_order_bits(ref);
			std::memcpy(&L.quadNodes[static_cast<size_t>(qi) * 32 + 24 + sidx], &ref, 4);
		}
	}
	L.quadRootRef = with_order_bits(L.quadRootRef);
	return TYR_OK;
}

} // namespace tyr
