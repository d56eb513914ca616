// traverse.hpp -- BVH traversal on gfx950: closest hit (reference: CachedBVH::intersect,
// bvh.h:118-161) and any hit (CachedBVH::intersectSimple, bvh.h:213-256), with
// BBox::intersect (Bbox.h:38-62) and Triangle::intersect (loader.h:21-46).
//
// Device layout (private to the library, built by host/bvh_layout.cpp from the
// reference's flat 32-byte node array):
//
//   PairNode, 64 B = 4 x dwordx4, one per INTERIOR node of the reference tree, holding
//   the boxes of BOTH children, slab-major so one 16-byte load feeds one axis:
//       q0 = { left.min.x, left.max.x, right.min.x, right.max.x }
//       q1 = {      ..y                                         }
//       q2 = {      ..z                                         }
//       q3 = { leftRef, rightRef, splitAxis, 0 }
//   left = the reference's node index+1, right = its secondChildOffset.
//   ChildRef: bit 31 = leaf; leaf: [30:26] = primitiveCount-1, [25:0] = primitiveOffset;
//             interior: index of the child's PairNode.
//   Triangles: 48 B = 3 x dwordx4 { vert.xyz, e1.x | e1.yz, e2.xy | e2.z, material, 0, 0 }.
//
// Why this shape: the reference fetches a 32-byte node, tests ONE box, then fetches the
// next node -- every box test is a dependent memory round trip.  A PairNode fetch is one
// round trip for two box tests, leaves need no node fetch at all, and a child whose box
// fails is never pushed or fetched.
//
// Visit-order equivalence (SURVEY.md section 7 "Traversal-order-dependent hits"): the
// reference tests a node's box when the node is VISITED, against the ray's current
// closest distance.  Here a child's box is tested when its PARENT is visited; the test
// is `overlap && tMin < dist && tMax > 0` (Bbox.h:61) and only `tMin < dist` depends on
// dist, which can only shrink, so (1) a child rejected early would also be rejected at
// visit time, and (2) for a pushed child, re-checking `tMin < dist` at pop time with the
// stored tMin reproduces the reference's decision exactly.  Near child first by
// dirIsNeg[splitAxis] (bvh.h:146-152), leaf primitives in array order (bvh.h:131), so
// the sequence of triangle tests -- and therefore the epsilon-dependent accept rule of
// bvh.h:134 -- is identical.
#pragma once

#include <type_traits>

#include <hip/hip_runtime.h>

#include "vecmath.hpp"

namespace tyr {

constexpr uint32_t kRefLeaf = 0x80000000u;
constexpr uint32_t kRefDone = 0xFFFFFFFFu; // "leaf" with every bit set: never produced by the layout pass
constexpr uint32_t kRefPop = 0xFFFFFFFEu;  // flat state machine: this lane must pop its stack next (a leaf needs count-1 < 31)
constexpr int kStackSize = 64;             // bvh.h:124 nodesToVisit[64]
constexpr uint32_t kMaxLeafPrims = 31; // count-1 <= 30 keeps 0xFFFFFFFE / 0xFFFFFFFF out of the leaf encoding
constexpr uint32_t kMaxPrimOffset = 1u << 26;
// interior reference of the quad layout: node index in bits 0..24, the node's visit-order bits in 25..30
// Top of the tree staged in LDS by the persistent traversal kernel: the first kStagedNodes records
// of the quad array are the tree's top levels in breadth-first order (host/bvh_layout.cpp).  In LDS they are kept
// vector-major -- vector v of node n at [v][n] -- so that lanes reading different nodes spread over 16 bank
// groups (node-major, every node would start on bank 0 or 32); lanes reading the same node are a broadcast.
constexpr uint32_t kStagedNodes = 64;
constexpr uint32_t kQuadOrderShift = 25;
constexpr uint32_t kQuadIndexMask = (1u << kQuadOrderShift) - 1u;

struct DevScene {
	const float4* quads; // QuadNode array, 8 float4 each (the production traversal)
	uint32_t quadRootRef;
	uint32_t nQuads;
	uint32_t nStaged;    // the first nStaged quad nodes are the top of the tree in breadth-first order (<= kStagedNodes)
	uint32_t quadMaxStack; // upper bound of a traversal's stack depth on this tree, any ray (host/bvh_layout.cpp)
	const float4* nodes; // PairNode array, 4 float4 each (the counting build and variants 0/1)
	const float4* tris;  // 3 float4 each
	float rootMin[3];
	float rootMax[3];
	uint32_t rootRef;    // kRefDone when the scene has no triangles (Scene.cpp:49-52)
	uint32_t nPairs;
	uint32_t nPrims;
};

#ifdef __HIPCC__

struct RayConst {
	f3 o, d, inv;
	bool nx, ny, nz;
};

__device__ __forceinline__ RayConst make_ray(f3 o, f3 d) {
	RayConst r;
	r.o = o;
	r.d = d;
	r.inv = mk3(1.f / d.x, 1.f / d.y, 1.f / d.z); // bvh.h:120
	r.nx = r.inv.x < 0;                           // bvh.h:121
	r.ny = r.inv.y < 0;
	r.nz = r.inv.z < 0;
	return r;
}

// Bbox.h:38-62 with the sign-selected bounds already picked.  Returns the pass/fail of the
// dist-independent part AND `tMin < lowest`; tMinOut is the entry distance used for the
// pop-time re-check.
__device__ __forceinline__ bool slab_test(const RayConst& r, float lox, float hix, float loy, float hiy, float loz, float hiz, float lowest, float& tMinOut) {
	float tMin = (lox - r.o.x) * r.inv.x;
	float tMax = (hix - r.o.x) * r.inv.x;
	const float tyMin = (loy - r.o.y) * r.inv.y;
	const float tyMax = (hiy - r.o.y) * r.inv.y;
	bool ok = !(tMin > tyMax || tyMin > tMax);
	if (tyMin > tMin)
		tMin = tyMin;
	if (tyMax < tMax)
		tMax = tyMax;
	const float tzMin = (loz - r.o.z) * r.inv.z;
	const float tzMax = (hiz - r.o.z) * r.inv.z;
	ok = ok && !(tMin > tzMax || tzMin > tMax);
	if (tzMin > tMin)
		tMin = tzMin;
	if (tzMax < tMax)
		tMax = tzMax;
	tMinOut = tMin;
	return ok && (tMin < lowest) && (tMax > 0);
}

// loader.h:21-46: Moller-Trumbore, back faces culled (det < 1e-7), 0 = miss
struct TriData {
	float4 a, b, c; // vert.xyz e1.x | e1.yz e2.xy | e2.z material pad pad
};
__device__ __forceinline__ TriData triangle_load(const float4* __restrict__ tris, uint32_t prim) {
	TriData d;
	d.a = tris[3 * prim + 0];
	d.b = tris[3 * prim + 1];
	d.c = tris[3 * prim + 2];
	return d;
}
__device__ __forceinline__ float triangle_test(const TriData& d, const RayConst& r);
__device__ __forceinline__ float triangle_test(const float4* __restrict__ tris, uint32_t prim, const RayConst& r) { return triangle_test(triangle_load(tris, prim), r); }
__device__ __forceinline__ float triangle_test(const TriData& d, const RayConst& r) {
	const float4 a = d.a, b = d.b, c = d.c;
	const f3 vert = mk3(a.x, a.y, a.z);
	const f3 e1 = mk3(a.w, b.x, b.y);
	const f3 e2 = mk3(b.z, b.w, c.x);
	const f3 pvec = cross(r.d, e2);
	const float det = dot(e1, pvec);
	if (det < 0.0000001f)
		return 0.0f;
	const float invDet = 1 / det;
	const f3 tvec = r.o - vert;
	const float u = dot(tvec, pvec) * invDet;
	if (u < 0 || u > 1)
		return 0.0f;
	const f3 qvec = cross(tvec, e1);
	const float v = dot(r.d, qvec) * invDet;
	if (v < 0 || u + v > 1)
		return 0.0f;
	return dot(e2, qvec) * invDet;
}

// The same test without a branch: every operation of loader.h:21-46 in its order, the three early-outs folded into one
// select at the end (identical results: a lane that would have left early computes values nobody reads -- 1 / det may be
// inf or NaN there, which no comparison of the select lets through).  For code whose cost is its SCALAR instructions
// (the four-lanes-to-a-ray drain): three early-outs are three exec-mask save / branch / restore sequences per test.
__device__ __forceinline__ float triangle_test_select(const TriData& d, const RayConst& r) {
	const float4 a = d.a, b = d.b, c = d.c;
	const f3 vert = mk3(a.x, a.y, a.z);
	const f3 e1 = mk3(a.w, b.x, b.y);
	const f3 e2 = mk3(b.z, b.w, c.x);
	const f3 pvec = cross(r.d, e2);
	const float det = dot(e1, pvec);
	const bool out0 = det < 0.0000001f;
	const float invDet = 1 / det;
	const f3 tvec = r.o - vert;
	const float u = dot(tvec, pvec) * invDet;
	const bool out1 = u < 0 || u > 1;
	const f3 qvec = cross(tvec, e1);
	const float v = dot(r.d, qvec) * invDet;
	const bool out2 = v < 0 || u + v > 1;
	const float t = dot(e2, qvec) * invDet;
	return (out0 || out1 || out2) ? 0.0f : t;
}

// Per-lane traversal stack: the first LDS_DEPTH entries in LDS, the rest in a private (scratch)
// array, the top entry cached in registers (a push followed by a pop never touches memory).
//
// Why LDS: on CDNA, vmcnt retires loads AND stores in issue order, so a scratch push (a store)
// sits in front of the next node fetch's wait and a scratch pop is a vector-memory round trip on
// the critical path "pop -> node address -> fetch".  LDS traffic is counted by lgkmcnt instead and
// returns in ~64 cycles.  Layout [depth][thread]: a lane always hits its own pair of banks whatever
// its depth (depth * 256 threads * 8 B is a whole number of 256-byte bank rows), so divergent
// depths never conflict.
//
// The arrays live OUTSIDE this struct (the kernel declares them and binds pointers): with the
// arrays as members, the compiler kept the whole object -- n, hasTop, the cached top -- in scratch
// (scratch_store_dword in the descent loop of the first build), because a struct holding a
// dynamically indexed array is not split into registers.
template <int LDS_DEPTH>
struct TravStack {
	static constexpr int kBlockThreads = 256;
	static constexpr int kSpill = kStackSize - LDS_DEPTH; // entries of the private arrays
	uint2* lds;          // this thread's column: entry d at lds[d * kBlockThreads]
	uint32_t* spillRef;  // kSpill entries
	float* spillT;
	int n;               // entries in memory
	uint32_t topRef;
	float topT;
	bool hasTop;
	bool overflow;
	__device__ __forceinline__ void bind(uint2* ldsColumn, uint32_t* refs, float* ts) {
		lds = ldsColumn;
		spillRef = refs;
		spillT = ts;
	}
	__device__ __forceinline__ void reset() {
		n = 0;
		hasTop = false;
		overflow = false;
	}
	__device__ __forceinline__ void push(uint32_t r, float t) {
		if (hasTop) {
			if (LDS_DEPTH > 0 && n < LDS_DEPTH) {
				lds[n * kBlockThreads] = make_uint2(topRef, __float_as_uint(topT));
				++n;
			} else if (n < kStackSize - 1) {
				spillRef[n - LDS_DEPTH] = topRef;
				spillT[n - LDS_DEPTH] = topT;
				++n;
			} else {
				overflow = true; // the reference's 64-entry array would be overrun here (bvh.h:124)
			}
		}
		topRef = r;
		topT = t;
		hasTop = true;
	}
	__device__ __forceinline__ bool pop(uint32_t& r, float& t) {
		if (hasTop) {
			r = topRef;
			t = topT;
			hasTop = false;
			return true;
		}
		if (n == 0)
			return false;
		--n;
		if (LDS_DEPTH > 0 && n < LDS_DEPTH) {
			const uint2 e = lds[n * kBlockThreads];
			r = e.x;
			t = __uint_as_float(e.y);
		} else {
			r = spillRef[n - LDS_DEPTH];
			t = spillT[n - LDS_DEPTH];
		}
		return true;
	}
};

// The flat traversal kernels' stack: the same storage, but no register-cached top and a wave-level fast path.
// PMC on the tuned kernel: 2.25 G scalar against 2.05 G vector instructions per frame -- the scalar unit (one per
// CU, shared by the four SIMDs) was as busy as the vector units, most of it exec-mask bookkeeping around the
// three pushes of a quad step: each push nested "is there a cached top", "LDS or private", "full?".  Here a wave
// first asks once whether ANY of its lanes could leave the LDS part during this step; if not (the common case) a
// push is one predicated ds_write_b64 and a pop one ds_read_b64.
template <int LDS_DEPTH, bool WITH_T = true>
struct LdsStack {
	// WITH_T = false: entries are the reference alone (4 bytes).  Any-hit traversal tests boxes against a bound that
	// never shrinks, so the pop-time "entry distance < bound" re-check of the closest-hit traversal is always true
	// and the entry distance need not be kept: half the LDS per lane and ds_*_b32 instead of ds_*_b64.
	static constexpr int kBlockThreads = 256;
	typedef typename std::conditional<WITH_T, uint2, uint32_t>::type entry_t;
	// An explicit LDS pointer: held as a generic pointer, the 4-byte variant's pop merged its LDS read and its
	// private-memory read into ONE flat_load through a selected base -- a flat access (with a wait on vmcnt AND
	// lgkmcnt behind it) for every pop of the any-hit traversal, instead of a ds_read_b32.
#if defined(__HIP_DEVICE_COMPILE__)
	typedef __attribute__((address_space(3))) entry_t* lds_column_t;
#else
	typedef entry_t* lds_column_t; // host pass of the same source: never executed
#endif
	lds_column_t lds;    // this thread's column: entry d at lds[d * kBlockThreads]
	uint32_t* spillRef;  // kStackSize - LDS_DEPTH entries
	float* spillT;
	int n;
	bool overflow;
	__device__ __forceinline__ void bind(entry_t* ldsColumn, uint32_t* refs, float* ts) {
		lds = (lds_column_t)ldsColumn;
		spillRef = refs;
		spillT = ts;
	}
	__device__ __forceinline__ void reset() {
		n = 0;
		overflow = false;
	}
	__device__ __forceinline__ void lds_put(int d, uint32_t r, float t) {
		if constexpr (WITH_T)
			lds[d * kBlockThreads] = make_uint2(r, __float_as_uint(t));
		else
			lds[d * kBlockThreads] = r;
	}
	__device__ __forceinline__ void lds_get(int d, uint32_t& r, float& t) {
		if constexpr (WITH_T) {
			const uint2 e = lds[d * kBlockThreads];
			r = e.x;
			t = __uint_as_float(e.y);
		} else {
			r = lds[d * kBlockThreads];
			t = -__builtin_inff();
		}
	}
	__device__ __forceinline__ void push(uint32_t r, float t) {
		if (LDS_DEPTH > 0 && n < LDS_DEPTH) {
			lds_put(n, r, t);
			++n;
		} else if (n < kStackSize) {
			spillRef[n - LDS_DEPTH] = r;
			if (WITH_T)
				spillT[n - LDS_DEPTH] = t;
			++n;
		} else {
			overflow = true; // the reference's 64-entry array would be overrun here (bvh.h:124)
		}
	}
	__device__ __forceinline__ bool pop(uint32_t& r, float& t) {
		if (n == 0)
			return false;
		// wave-level: does any popping lane sit in the private part?
		if (LDS_DEPTH > 0 && __builtin_amdgcn_ballot_w64(n > LDS_DEPTH) == 0ull) {
			--n;
			lds_get(n, r, t);
			return true;
		}
		--n;
		if (LDS_DEPTH > 0 && n < LDS_DEPTH) {
			lds_get(n, r, t);
		} else {
			r = spillRef[n - LDS_DEPTH];
			t = WITH_T ? spillT[n - LDS_DEPTH] : -__builtin_inff();
		}
		return true;
	}
	// up to three pushes of one quad step (a first, c last); mX = the lanes that push entry X
	__device__ __forceinline__ void push3(unsigned long long ma, uint32_t ra, float ta, unsigned long long mb, uint32_t rb, float tb, unsigned long long mc, uint32_t rc, float tc) {
		if (LDS_DEPTH >= 3 && __builtin_amdgcn_ballot_w64(n > LDS_DEPTH - 3) == 0ull) {
			if (__builtin_amdgcn_inverse_ballot_w64(ma)) {
				lds_put(n, ra, ta);
				++n;
			}
			if (__builtin_amdgcn_inverse_ballot_w64(mb)) {
				lds_put(n, rb, tb);
				++n;
			}
			if (__builtin_amdgcn_inverse_ballot_w64(mc)) {
				lds_put(n, rc, tc);
				++n;
			}
			return;
		}
		if (__builtin_amdgcn_inverse_ballot_w64(ma))
			push(ra, ta);
		if (__builtin_amdgcn_inverse_ballot_w64(mb))
			push(rb, tb);
		if (__builtin_amdgcn_inverse_ballot_w64(mc))
			push(rc, tc);
	}
};

struct PairTest {
	uint32_t nearRef, farRef;
	float nearT, farT;
	bool nearHit, farHit;
	bool synthetic; // axis 3: continuation of an over-long leaf, not a node of the reference tree
};

__device__ __forceinline__ PairTest test_pair(const float4* __restrict__ nodes, uint32_t idx, const RayConst& r, float dist) {
	const float4 qx = nodes[4 * idx + 0];
	const float4 qy = nodes[4 * idx + 1];
	const float4 qz = nodes[4 * idx + 2];
	const float4 qr = nodes[4 * idx + 3];
	const uint32_t leftRef = __float_as_uint(qr.x), rightRef = __float_as_uint(qr.y), axis = __float_as_uint(qr.z);
	float tL, tR;
	// bounds[dirIsNeg] is the entry plane, bounds[1 - dirIsNeg] the exit plane (Bbox.h:39-42)
	const bool hL = slab_test(r, r.nx ? qx.y : qx.x, r.nx ? qx.x : qx.y, r.ny ? qy.y : qy.x, r.ny ? qy.x : qy.y, r.nz ? qz.y : qz.x, r.nz ? qz.x : qz.y, dist, tL);
	const bool hR = slab_test(r, r.nx ? qx.w : qx.z, r.nx ? qx.z : qx.w, r.ny ? qy.w : qy.z, r.ny ? qy.z : qy.w, r.nz ? qz.w : qz.z, r.nz ? qz.z : qz.w, dist, tR);
	// bvh.h:146-152: dirIsNeg[splitAxis] -> second child first.  axis 3 (synthetic chain) = left first.
	const bool rightFirst = (axis == 0) ? r.nx : (axis == 1) ? r.ny : (axis == 2) ? r.nz : false;
	PairTest p;
	p.nearRef = rightFirst ? rightRef : leftRef;
	p.farRef = rightFirst ? leftRef : rightRef;
	p.nearT = rightFirst ? tR : tL;
	p.farT = rightFirst ? tL : tR;
	p.nearHit = rightFirst ? hR : hL;
	p.farHit = rightFirst ? hL : hR;
	p.synthetic = (axis == 3);
	if (p.synthetic) {
		// synthetic split of an over-long leaf: the reference tests every primitive of a leaf once
		// the leaf's box passed (bvh.h:131), so neither half is re-tested here or at pop time
		p.nearHit = true;
		p.farHit = true;
		p.farT = -__builtin_inff();
	}
	return p;
}


// ---- fast exact pair test for rays whose 1/d components are all finite ---------------------
// With finite inv there is no NaN anywhere in Bbox.h:38-62 (NaN needs 0 * inf), and multiplying
// by inv is monotonic, so the sign-selected planes of Bbox.h:39-42 are min/max of the two
// products; the two early-outs plus the running max/min of Bbox.h:44-59 are then exactly
//      tMin = max3(near_x, near_y, near_z)   tMax = min3(far_x, far_y, far_z)
//      hit  = tMin <= tMax  &&  tMin < lowest  &&  tMax > 0
// (all nine near_i <= far_j comparisons of the reference collapse to max(near) <= min(far)).
// That is ~25 VALU per box instead of ~45.  Rays with a zero direction component (inv = inf)
// keep the generic path above.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bool slab_fast(const RayConst& r, float lox, float hix, float loy, float hiy, float loz, float hiz, float lowest, float& tMinOut) {
	// both planes of an axis in one packed operation (v_pk_add_f32 / v_pk_mul_f32: two IEEE binary32 results per
	// instruction, each rounded exactly like the scalar form -- contraction is off for the whole file)
	const float ox = r.o.x, oy = r.o.y, oz = r.o.z, ix = r.inv.x, iy = r.inv.y, iz = r.inv.z;
	v2f x = { lox, hix }, y = { loy, hiy }, z = { loz, hiz };
	x = (x - ox) * ix;
	y = (y - oy) * iy;
	z = (z - oz) * iz;
	const float tMin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x.x, x.y), __builtin_fminf(y.x, y.y)), __builtin_fminf(z.x, z.y));
	const float tMax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x.x, x.y), __builtin_fmaxf(y.x, y.y)), __builtin_fmaxf(z.x, z.y));
	tMinOut = tMin;
	return (tMin <= tMax) && (tMin < lowest) && (tMax > 0);
}
// the same test with its answer as a wave-wide lane mask: the ballot of each comparison IS that comparison's
// result register, and the three are combined by scalar ands -- ballot(a && b && c) made the compiler turn the
// combined mask into a 0/1 VGPR and compare that against zero again, for each of the four boxes of every node
typedef unsigned long long lanemask;
__device__ __forceinline__ lanemask slab_fast_mask(const RayConst& r, float lox, float hix, float loy, float hiy, float loz, float hiz, float lowest, float& tMinOut) {
	const float ox = r.o.x, oy = r.o.y, oz = r.o.z, ix = r.inv.x, iy = r.inv.y, iz = r.inv.z;
	v2f x = { lox, hix }, y = { loy, hiy }, z = { loz, hiz };
	x = (x - ox) * ix;
	y = (y - oy) * iy;
	z = (z - oz) * iz;
	const float tMin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x.x, x.y), __builtin_fminf(y.x, y.y)), __builtin_fminf(z.x, z.y));
	const float tMax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x.x, x.y), __builtin_fmaxf(y.x, y.y)), __builtin_fmaxf(z.x, z.y));
	tMinOut = tMin;
	return __builtin_amdgcn_ballot_w64(tMin <= tMax) & __builtin_amdgcn_ballot_w64(tMin < lowest) & __builtin_amdgcn_ballot_w64(tMax > 0);
}
__device__ __forceinline__ bool ray_is_regular(const RayConst& r) {
	// |inv| < inf for all three components (also false for NaN)
	return (fabsf(r.inv.x) < __builtin_inff()) && (fabsf(r.inv.y) < __builtin_inff()) && (fabsf(r.inv.z) < __builtin_inff());
}
__device__ __forceinline__ PairTest test_pair_fast(const float4* __restrict__ nodes, uint32_t idx, const RayConst& r, float dist) {
	const float4 qx = nodes[4 * idx + 0];
	const float4 qy = nodes[4 * idx + 1];
	const float4 qz = nodes[4 * idx + 2];
	const float4 qr = nodes[4 * idx + 3];
	const uint32_t leftRef = __float_as_uint(qr.x), rightRef = __float_as_uint(qr.y), axis = __float_as_uint(qr.z);
	float tL, tR;
	const bool hL = slab_fast(r, qx.x, qx.y, qy.x, qy.y, qz.x, qz.y, dist, tL);
	const bool hR = slab_fast(r, qx.z, qx.w, qy.z, qy.w, qz.z, qz.w, dist, tR);
	const bool rightFirst = (axis == 0) ? r.nx : (axis == 1) ? r.ny : (axis == 2) ? r.nz : false;
	PairTest p;
	p.nearRef = rightFirst ? rightRef : leftRef;
	p.farRef = rightFirst ? leftRef : rightRef;
	p.nearT = rightFirst ? tR : tL;
	p.farT = rightFirst ? tL : tR;
	p.nearHit = rightFirst ? hR : hL;
	p.farHit = rightFirst ? hL : hR;
	p.synthetic = (axis == 3);
	if (p.synthetic) {
		p.nearHit = true;
		p.farHit = true;
		p.farT = -__builtin_inff();
	}
	return p;
}


// ---- quad nodes (128 B): a node and both its children in one fetch -----------------------------
// Layout: host/bvh_layout.cpp.  One step tests the four GRANDCHILD boxes (or a leaf child in slot 0 /
// 2 of its group) instead of child, then grandchild: half the dependent memory round trips.
//
// Exactness.  Boxes are exact unions (min/max) and rounding is monotone, so a child's slab interval
// [tMin, tMax] is nested in its parent's: if the child passes Bbox.h:38-62 for some bound, so does the
// parent for any bound at least as large.  Skipping the intermediate node's test therefore never
// changes WHICH leaves are reached; and the order in which they are reached is still the reference's
// depth-first order: group of the near child first (dirIsNeg[axis of the node], bvh.h:146-152), then
// inside each group its near slot first (dirIsNeg[axis of that child]).  Every reached box is
// re-checked against the current distance when popped, as for pair nodes.
// Hit flags travel as WAVE-WIDE lane masks (the ballot of the per-lane flag, a scalar register pair): as `bool`s
// they met in a phi where the finite-1/d and the generic box tests rejoin and were materialised as 0/1 vector
// registers, re-compared before every use; as masks the near/far reordering is scalar and/or work.
__device__ __forceinline__ lanemask lanes_where(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool lane_in(lanemask m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
__device__ __forceinline__ lanemask mask_select(lanemask sel, lanemask a, lanemask b) { return (sel & a) | (~sel & b); }

struct QuadHits { // in visit order
	uint32_t ref[4];
	float t[4];
	lanemask hit[4];
};

template <bool FAST>
__device__ __forceinline__ lanemask slab_any(const RayConst& r, float lox, float hix, float loy, float hiy, float loz, float hiz, float lowest, float& tOut) {
	if (FAST)
		return slab_fast_mask(r, lox, hix, loy, hiy, loz, hiz, lowest, tOut);
	return lanes_where(slab_test(r, r.nx ? hix : lox, r.nx ? lox : hix, r.ny ? hiy : loy, r.ny ? loy : hiy, r.nz ? hiz : loz, r.nz ? loz : hiz, lowest, tOut));
}

// FAST: every lane of the wave has a finite 1/d (the caller decides once per wave, not per box: a
// per-lane choice made the compiler emit both paths with exec juggling around each of the four tests).
// ORDERED: closest-hit needs the reference's visit order; any-hit (bvh.h:213-256) does not depend on it.
template <bool FAST, bool ORDERED, bool STAGED = false, int STAGE_STRIDE = (int)kStagedNodes>
__device__ __forceinline__ QuadHits test_quad(const float4* __restrict__ quads, uint32_t ref, const RayConst& r, float dist, const float4* staged = nullptr, uint32_t nStaged = 0) {
	const uint32_t idx = ref & kQuadIndexMask;
	const float4* q = quads + 8 * idx;
	const uint32_t meta = ref >> kQuadOrderShift; // bit 31 of an interior reference is clear
	float4 x01, x23, y01, y23, z01, z23, rf;
	if (STAGED && idx < nStaged) {
		// an explicit LDS pointer: as a generic pointer the compiler selected between the two bases and issued
		// 28 flat_load_dword per node
#if defined(__HIP_DEVICE_COMPILE__)
		typedef __attribute__((address_space(3))) const float4* lds_f4;
		const lds_f4 c = (lds_f4)staged + idx;
#else
		const float4* c = staged + idx; // host pass of the same source: never executed
#endif
		x01 = c[0 * STAGE_STRIDE], x23 = c[1 * STAGE_STRIDE], y01 = c[2 * STAGE_STRIDE], y23 = c[3 * STAGE_STRIDE];
		z01 = c[4 * STAGE_STRIDE], z23 = c[5 * STAGE_STRIDE], rf = c[6 * STAGE_STRIDE];
		// keeps this tail different from the global branch's: otherwise the two sets of loads are merged into
		// loads through one generic pointer (flat_load_dword x 28)
		__asm__ volatile("" : "+v"(rf.x));
	} else {
		x01 = q[0], x23 = q[1], y01 = q[2], y23 = q[3], z01 = q[4], z23 = q[5], rf = q[6];
	}
	const uint32_t r0 = __float_as_uint(rf.x), r1 = __float_as_uint(rf.y), r2 = __float_as_uint(rf.z), r3 = __float_as_uint(rf.w);
	float t0, t1, t2, t3;
	const lanemask H0 = slab_any<FAST>(r, x01.x, x01.y, y01.x, y01.y, z01.x, z01.y, dist, t0);
	const lanemask H1 = slab_any<FAST>(r, x01.z, x01.w, y01.z, y01.w, z01.z, z01.w, dist, t1);
	const lanemask H2 = slab_any<FAST>(r, x23.x, x23.y, y23.x, y23.y, z23.x, z23.y, dist, t2);
	const lanemask H3 = slab_any<FAST>(r, x23.z, x23.w, y23.z, y23.w, z23.z, z23.w, dist, t3);
	// No special cases: an unused slot's box is at +infinity and never hit, and the slots of a synthetic node (the
	// consecutive chunks of one over-long leaf, bvh.h:131) have boxes from -inf to +inf that every ray enters at
	// -inf, with order bits that swap nothing (host/bvh_layout.cpp).
	if (!ORDERED) {
		QuadHits o;
		o.ref[0] = r0, o.ref[1] = r1, o.ref[2] = r2, o.ref[3] = r3;
		o.t[0] = t0, o.t[1] = t1, o.t[2] = t2, o.t[3] = t3;
		o.hit[0] = H0, o.hit[1] = H1, o.hit[2] = H2, o.hit[3] = H3;
		return o;
	}
	// "is the ray's direction negative along this split axis": a bit extract from the three sign bits (constant for
	// the ray, so hoisted out of the descent loop).  Written as `axis == 0 ? nx : axis == 1 ? ny : nz` the compiler
	// built each of the three answers from nested exec-mask branches -- 75 scalar instructions and a dozen
	// mask -> VGPR -> mask copies per trip, in a loop whose scalar pipe is as busy as its vector pipe.
	const uint32_t signBits = (r.nx ? 1u : 0u) | (r.ny ? 2u : 0u) | (r.nz ? 4u : 0u);
	const uint32_t aT = meta & 3u, aL = (meta >> 2) & 3u, aR = (meta >> 4) & 3u;
	const bool bT = ((signBits >> aT) & 1u) != 0u; // axis code 3 (synthetic nodes): bit 3 is clear, nothing is swapped
	const bool bL = ((signBits >> aL) & 1u) != 0u;
	const bool bR = ((signBits >> aR) & 1u) != 0u;
	const lanemask BT = lanes_where(bT), BL = lanes_where(bL), BR = lanes_where(bR);
	// near slot first inside each group
	const uint32_t lr0 = bL ? r1 : r0, lr1 = bL ? r0 : r1;
	const float lt0 = bL ? t1 : t0, lt1 = bL ? t0 : t1;
	const lanemask LH0 = mask_select(BL, H1, H0), LH1 = mask_select(BL, H0, H1);
	const uint32_t rr0 = bR ? r3 : r2, rr1 = bR ? r2 : r3;
	const float rt0 = bR ? t3 : t2, rt1 = bR ? t2 : t3;
	const lanemask RH0 = mask_select(BR, H3, H2), RH1 = mask_select(BR, H2, H3);
	// near group first
	QuadHits o;
	o.ref[0] = bT ? rr0 : lr0;
	o.ref[1] = bT ? rr1 : lr1;
	o.ref[2] = bT ? lr0 : rr0;
	o.ref[3] = bT ? lr1 : rr1;
	o.t[0] = bT ? rt0 : lt0;
	o.t[1] = bT ? rt1 : lt1;
	o.t[2] = bT ? lt0 : rt0;
	o.t[3] = bT ? lt1 : rt1;
	o.hit[0] = mask_select(BT, RH0, LH0);
	o.hit[1] = mask_select(BT, RH1, LH1);
	o.hit[2] = mask_select(BT, LH0, RH0);
	o.hit[3] = mask_select(BT, LH1, RH1);
	return o;
}

struct VisitCount {
	uint32_t nodes, tris;
};

// Closest hit.  `dist` / `prim` are updated like ray.distance / ray.identifier (bvh.h:135-136).
// COUNT: also count nodes visited / triangles tested by the reference's rule (bvh.h:164-209:
// one per loop iteration = every node fetched, including those whose box test fails).
template <bool COUNT, class Stack>
__device__ __forceinline__ bool bvh_closest(const DevScene& sc, const RayConst& r, float& dist, int& prim, Stack& st, VisitCount& vc) {
	bool hit = false;
	st.reset();
	uint32_t ref;
	{
		float t0;
		const bool ok = slab_test(r, r.nx ? sc.rootMax[0] : sc.rootMin[0], r.nx ? sc.rootMin[0] : sc.rootMax[0], r.ny ? sc.rootMax[1] : sc.rootMin[1], r.ny ? sc.rootMin[1] : sc.rootMax[1],
			r.nz ? sc.rootMax[2] : sc.rootMin[2], r.nz ? sc.rootMin[2] : sc.rootMax[2], dist, t0);
		if (COUNT)
			vc.nodes += 1;
		ref = ok ? sc.rootRef : kRefDone;
	}
	while (ref != kRefDone) {
		// interior nodes: keep descending until this lane holds a leaf (or is done)
		while ((int)ref >= 0) {
			const PairTest p = test_pair(sc.nodes, ref, r, dist);
			if (COUNT && !p.synthetic)
				vc.nodes += 2; // closest hit never exits early: both children are visited by the reference
			if (p.nearHit) {
				if (p.farHit)
					st.push(p.farRef, p.farT);
				ref = p.nearRef;
			} else if (p.farHit) {
				ref = p.farRef;
			} else {
				ref = kRefDone;
				uint32_t pr;
				float pt;
				while (st.pop(pr, pt)) {
					if (pt < dist) { // the pop-time half of Bbox.h:61
						ref = pr;
						break;
					}
				}
			}
		}
		if (ref == kRefDone)
			break;
		// leaf: bvh.h:129-140
		const uint32_t off = ref & (kMaxPrimOffset - 1);
		const uint32_t cnt = ((ref >> 26) & 31u) + 1u;
		for (uint32_t i = 0; i < cnt; ++i) {
			const float t = triangle_test(sc.tris, off + i, r);
			if (COUNT)
				vc.tris += 1;
			if (t > kEpsilon && t < dist && ((dist - t) > kEpsilon)) {
				prim = (int)(off + i);
				dist = t;
				hit = true;
			}
		}
		ref = kRefDone;
		uint32_t pr;
		float pt;
		while (st.pop(pr, pt)) {
			if (pt < dist) {
				ref = pr;
				break;
			}
		}
	}
	return hit;
}

// Any hit within `closest` (bvh.h:213-256).  The bound never shrinks, so the result does not
// depend on visit order; the same near-first order is kept so COUNT reproduces the reference's
// visit counts (nodes popped before the early return).
template <bool COUNT, class Stack>
__device__ __forceinline__ bool bvh_any(const DevScene& sc, const RayConst& r, float closest, Stack& st, VisitCount& vc) {
	st.reset();
	uint32_t ref;
	{
		float t0;
		const bool ok = slab_test(r, r.nx ? sc.rootMax[0] : sc.rootMin[0], r.nx ? sc.rootMin[0] : sc.rootMax[0], r.ny ? sc.rootMax[1] : sc.rootMin[1], r.ny ? sc.rootMin[1] : sc.rootMax[1],
			r.nz ? sc.rootMax[2] : sc.rootMin[2], r.nz ? sc.rootMin[2] : sc.rootMax[2], closest, t0);
		if (COUNT)
			vc.nodes += 1;
		ref = ok ? sc.rootRef : kRefDone;
	}
	const float kFailed = __builtin_inff(); // COUNT only: a far child that failed its box test but is still "visited" when popped
	while (ref != kRefDone) {
		while ((int)ref >= 0) {
			const PairTest p = test_pair(sc.nodes, ref, r, closest);
			if (COUNT && !p.synthetic) {
				vc.nodes += 1; // the near child is visited next
				st.push(p.farRef, p.farHit ? p.farT : kFailed);
				ref = p.nearHit ? p.nearRef : kRefDone;
			} else {
				if (p.nearHit) {
					if (p.farHit)
						st.push(p.farRef, p.farT);
					ref = p.nearRef;
				} else if (p.farHit) {
					ref = p.farRef;
				} else {
					ref = kRefDone;
				}
			}
			if (ref == kRefDone) {
				uint32_t pr;
				float pt;
				while (st.pop(pr, pt)) {
					if (COUNT)
						vc.nodes += 1;
					if (pt < closest) {
						ref = pr;
						break;
					}
				}
			}
		}
		if (ref == kRefDone)
			break;
		const uint32_t off = ref & (kMaxPrimOffset - 1);
		const uint32_t cnt = ((ref >> 26) & 31u) + 1u;
		for (uint32_t i = 0; i < cnt; ++i) {
			const float t = triangle_test(sc.tris, off + i, r);
			if (COUNT)
				vc.tris += 1;
			if (t > kEpsilon && ((closest - t) > kEpsilon))
				return true; // bvh.h:232-236
		}
		ref = kRefDone;
		uint32_t pr;
		float pt;
		while (st.pop(pr, pt)) {
			if (COUNT)
				vc.nodes += 1;
			if (pt < closest) {
				ref = pr;
				break;
			}
		}
	}
	return false;
}

#endif // __HIPCC__

} // namespace tyr
