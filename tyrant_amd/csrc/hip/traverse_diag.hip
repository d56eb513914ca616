// traverse_diag.hip -- the earlier traversal kernels, kept for A/B runs and for the variant parity tests
// (libtyrant_hip_diag.so, -DTYR_DIAG; never part of libtyrant_hip.so): variant 0 = one thread per queue slot,
// variant 1 = persistent waves with lane refill.  DESIGN.md 4.4 has the measurements that retired them.
#include "device_common.hpp"

namespace tyr {

// traversal-stack storage of one thread: LDS column + private arrays, bound to a TravStack<STACK_LDS>
#define TYR_DECLARE_STACK(st)                                                        \
	__shared__ uint2 smem_[STACK_LDS ? STACK_LDS * kBlock : 1];                      \
	uint32_t spillRef_[kStackSize - STACK_LDS];                                      \
	float spillT_[kStackSize - STACK_LDS];                                           \
	TravStack<STACK_LDS> st;                                                         \
	st.bind(smem_ + threadIdx.x, spillRef_, spillT_);                                \
	st.reset();


// ======================================================================================
// extend, kernel.cu:331-343 via intersect_scene, kernel.cu:125-142
// ======================================================================================
template <bool COUNT, int STACK_LDS>
__global__ void __launch_bounds__(kBlock) k_extend(const FrameParams P) {
	TYR_DECLARE_STACK(st)
	const uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
	const uint32_t nLive = P.k->n_live;
	VisitCount vc{ 0, 0 };
	bool overflow = false;
	if (slot < nLive) {
		const float4 a = P.work.o_dx[slot];
		const float2 b = P.work.dyz[slot];
		const f3 o = mk3(a.x, a.y, a.z), d = mk3(a.w, b.x, b.y);
		float dist = kVeryFar;
		uint32_t id = 0;
#pragma unroll
		for (int i = TYR_NUM_SPHERES; i--;) {
			const float t = sphere_intersect(P.spheres[i], o, d);
			if (t && t < dist) {
				dist = t;
				id = kHitSphere | (uint32_t)i;
			}
		}
		if (P.scene.rootRef != kRefDone) {
			const RayConst r = make_ray(o, d);
			int prim = 0;
			if (bvh_closest<COUNT>(P.scene, r, dist, prim, st, vc))
				id = (uint32_t)prim;
			overflow = st.overflow;
		}
		P.work.hit[slot] = make_float2(dist, __uint_as_float(id));
	}
	if (overflow)
		atomicOr(&P.k->device_error, kErrStackOverflow);
	if (COUNT) {
		wave_add_u64(&P.k->nodes_extend, vc.nodes);
		wave_add_u64(&P.k->tris_extend, vc.tris);
	}
}

// ======================================================================================
// connect, kernel.cu:630-646 via intersect_scene_simple, kernel.cu:162-174
// ======================================================================================
template <bool COUNT, int STACK_LDS>
__global__ void __launch_bounds__(kBlock) k_connect(const FrameParams P) {
	TYR_DECLARE_STACK(st)
	const uint32_t index = blockIdx.x * kBlock + threadIdx.x;
	const uint32_t n = P.kc->shadow_cnt;
	VisitCount vc{ 0, 0 };
	bool overflow = false;
	uint32_t visible = 0;
	if (index < n) {
		const float4 a = P.shadow.o_dx[index];
		const float4 b = P.shadow.dyz_cd_ix[index];
		const f3 o = mk3(a.x, a.y, a.z), d = mk3(a.w, b.x, b.y);
		const float closest = b.z;
		bool occluded = false;
		if (P.scene.rootRef != kRefDone) {
			const RayConst r = make_ray(o, d);
			occluded = bvh_any<COUNT>(P.scene, r, closest, st, vc);
			overflow = st.overflow;
		}
		if (!occluded) {
#pragma unroll
			for (int i = TYR_NUM_SPHERES; i--;) {
				const float t = sphere_intersect(P.spheres[i], o, d);
				if (t && (t + kEpsilon) < closest) {
					occluded = true;
					break;
				}
			}
		}
		if (!occluded) {
			const float4 c = P.shadow.color[index];
			float* px = reinterpret_cast<float*>(&P.blit[__float_as_int(b.w)]);
			if (c.x != 0.0f)
				atomicAdd(px + 0, c.x);
			if (c.y != 0.0f)
				atomicAdd(px + 1, c.y);
			if (c.z != 0.0f)
				atomicAdd(px + 2, c.z);
			visible = 1;
		}
	}
	if (overflow)
		atomicOr(&P.k->device_error, kErrStackOverflow);
	wave_add_u64(&P.k->n_shadow_visible, visible);
	if (COUNT) {
		wave_add_u64(&P.k->nodes_connect, vc.nodes);
		wave_add_u64(&P.k->tris_connect, vc.tris);
	}
}


// ======================================================================================
// Persistent traversal (variant 1): waves stay resident and every lane that finishes its
// ray takes the next queue slot from a device-wide ticket.
//
// Why: with one thread per slot, rocprofv3 on MI355X shows k_extend issuing ~13,000 VALU
// instructions per wave at 15 % lane utilisation (SQ_THREAD_CYCLES_VALU / (64 *
// SQ_ACTIVE_INST_VALU)): a wave runs as long as its longest ray while rays that miss the
// root box idle from the first instruction.  The kernel is VALU-issue bound, not memory
// bound (L2 hit rate 96 %), so the lever is lanes doing work.  Refilling is done by the
// wave as a whole (ballot, one atomicAdd per refill, ranks by popcount) once at least
// `refillMinIdle` lanes are free, so the ~100-instruction ray set-up is not paid for one
// lane at a time.  The seven sphere tests (kernel.cu:129-136) move into a coherent
// one-thread-per-slot pre-pass: they are the same for every ray and would otherwise run
// under a partial mask inside the refill.
// Results do not depend on which lane traces which ray: each ray's answer goes to its
// own slot.
// ======================================================================================

template <bool COUNT, int STACK_LDS>
__global__ void __launch_bounds__(kBlock) k_extend_persistent(const FrameParams P) {
	TYR_DECLARE_STACK(st)
	const uint32_t lane = lane_id();
	const unsigned long long below = (1ull << lane) - 1ull;
	const uint32_t nLive = P.k->n_live;
	const DevScene& sc = P.scene;
	RayConst r = {};
	float dist = 0.0f;
	uint32_t ref = kRefDone, slot = 0;
	int prim = 0;
	bool hitTri = false, live = false, overflow = false;
	VisitCount vc{ 0, 0 };
	uint32_t dbg[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }; // COUNT only

#define TYR_DBG(i)                                                     \
	if (COUNT) {                                                       \
		const unsigned long long m_ = __ballot(1);                     \
		if (lane == (uint32_t)__ffsll((long long)m_) - 1) {            \
			dbg[i] += 1;                                               \
			dbg[i + 1] += __popcll(m_);                                \
		}                                                              \
	}
	bool exhausted = (sc.rootRef == kRefDone); // no triangles: the pre-pass already wrote every answer
	uint32_t chunkNext = 0, chunkEnd = 0;

	for (;;) {
		const unsigned long long idleMask = __ballot(!live);
		const uint32_t nIdle = __popcll(idleMask);
		if (!exhausted && nIdle >= P.refillMinIdle) {
			if (chunkNext >= chunkEnd) {
				// one returning atomic per `ticketChunk` rays: a single device-wide word serves only ~88 dequeues/us
				// (MI355X_MICROARCH.md "dequeue"), which capped this kernel at ~0.9 ms when every refill paid one
				uint32_t base = 0;
				if (lane == 0)
					base = atomicAdd(&P.k->extend_ticket, P.ticketChunk);
				base = __shfl(base, 0, 64);
				chunkNext = base < nLive ? base : nLive;
				chunkEnd = (base + P.ticketChunk) < nLive ? (base + P.ticketChunk) : nLive;
				exhausted = (chunkNext >= chunkEnd);
			}
			const uint32_t take = (chunkEnd - chunkNext) < nIdle ? (chunkEnd - chunkNext) : nIdle;
			const uint32_t base = chunkNext;
			chunkNext += take;
			if (!live) {
				const uint32_t rank = __popcll(idleMask & below);
				const uint32_t s = base + rank;
				if (rank < take) {
					TYR_DBG(6)
					const float4 a = P.work.o_dx[s];
					const float2 b = P.work.dyz[s];
					const float2 h = P.work.hit[s];
					r = make_ray(mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y));
					dist = h.x;
					slot = s;
					hitTri = false;
					live = true;
					st.reset();
					ref = root_ref(sc, r, dist);
					if (COUNT)
						vc.nodes += 1;
				}
			}
		}
		if (__ballot(live) == 0ull) {
			if (exhausted)
				break;
			continue;
		}
		// ---- one macro step: descend until every live lane holds a leaf (or is done), then the leaves ----
		while ((int)ref >= 0) {
			TYR_DBG(0)
			const PairTest p = test_pair(sc.nodes, ref, r, dist);
			if (COUNT && !p.synthetic)
				vc.nodes += 2;
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
					TYR_DBG(2)
					if (pt < dist) {
						ref = pr;
						break;
					}
				}
			}
		}
		if (ref != kRefDone) {
			const uint32_t off = ref & (kMaxPrimOffset - 1);
			const uint32_t cnt = ((ref >> 26) & 31u) + 1u;
			for (uint32_t i = 0; i < cnt; ++i) {
				TYR_DBG(4)
				const float t = triangle_test(sc.tris, off + i, r);
				if (COUNT)
					vc.tris += 1;
				if (t > kEpsilon && t < dist && ((dist - t) > kEpsilon)) {
					prim = (int)(off + i);
					dist = t;
					hitTri = true;
				}
			}
			ref = kRefDone;
			uint32_t pr;
			float pt;
			while (st.pop(pr, pt)) {
				TYR_DBG(2)
				if (pt < dist) {
					ref = pr;
					break;
				}
			}
		}
		if (live && ref == kRefDone) {
			// this ray is finished (bvh.h:155-156): a triangle hit replaces the sphere answer (kernel.cu:138-140)
			if (hitTri)
				P.work.hit[slot] = make_float2(dist, __uint_as_float((uint32_t)prim));
			overflow = overflow || st.overflow;
			live = false;
		}
	}
	if (overflow)
		atomicOr(&P.k->device_error, kErrStackOverflow);
	if (COUNT) {
		wave_add_u64(&P.k->nodes_extend, vc.nodes);
		wave_add_u64(&P.k->tris_extend, vc.tris);
		for (int i = 0; i < 8; ++i)
			wave_add_u64(&P.k->debug[i], dbg[i]);
	}
#undef TYR_DBG
}

template <bool COUNT, int STACK_LDS>
__global__ void __launch_bounds__(kBlock) k_connect_persistent(const FrameParams P) {
	TYR_DECLARE_STACK(st)
	const uint32_t lane = lane_id();
	const unsigned long long below = (1ull << lane) - 1ull;
	const uint32_t nRays = P.kc->shadow_cnt;
	const DevScene& sc = P.scene;
	const bool haveBvh = (sc.rootRef != kRefDone);
	RayConst r = {};
	float closest = 0.0f;
	uint32_t ref = kRefDone, index = 0;
	bool live = false, occluded = false, overflow = false;
	VisitCount vc{ 0, 0 };
	uint32_t visible = 0;
	bool exhausted = false;
	const float kFailed = __builtin_inff();

	for (;;) {
		const unsigned long long idleMask = __ballot(!live);
		const uint32_t nIdle = __popcll(idleMask);
		if (!exhausted && nIdle >= P.refillMinIdle) {
			const uint32_t leader = __ffsll((long long)idleMask) - 1;
			uint32_t base = 0;
			if (lane == leader)
				base = atomicAdd(&P.kc->ticket, nIdle);
			base = __shfl(base, leader, 64);
			exhausted = (base + nIdle >= nRays);
			if (!live) {
				const uint32_t s = base + __popcll(idleMask & below);
				if (s < nRays) {
					const float4 a = P.shadow.o_dx[s];
					const float4 b = P.shadow.dyz_cd_ix[s];
					const float sphereOccluded = reinterpret_cast<const float*>(&P.shadow.color[s])[3];
					index = s;
					closest = b.z;
					occluded = (sphereOccluded != 0.0f);
					live = true;
					st.reset();
					ref = kRefDone;
					// COUNT keeps the reference's order (BVH first for every ray, kernel.cu:165) so the visit counts are its counts
					if (haveBvh && (COUNT || !occluded)) {
						r = make_ray(mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y));
						ref = root_ref(sc, r, closest);
						if (COUNT)
							vc.nodes += 1;
					}
				}
			}
		}
		if (__ballot(live) == 0ull) {
			if (exhausted)
				break;
			continue;
		}
		while ((int)ref >= 0) {
			const PairTest p = test_pair(sc.nodes, ref, r, closest);
			if (COUNT && !p.synthetic) {
				vc.nodes += 1;
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
		if (ref != kRefDone) {
			const uint32_t off = ref & (kMaxPrimOffset - 1);
			const uint32_t cnt = ((ref >> 26) & 31u) + 1u;
			bool found = false;
			for (uint32_t i = 0; i < cnt && !found; ++i) {
				const float t = triangle_test(sc.tris, off + i, r);
				if (COUNT)
					vc.tris += 1;
				found = (t > kEpsilon && ((closest - t) > kEpsilon)); // bvh.h:232-236
			}
			ref = kRefDone;
			if (found) {
				occluded = true;
			} else {
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
		if (live && ref == kRefDone) {
			if (!occluded) { // kernel.cu:640-644
				const float4 c = P.shadow.color[index];
				const float4 b = P.shadow.dyz_cd_ix[index];
				float* px = reinterpret_cast<float*>(&P.blit[__float_as_int(b.w)]);
				if (c.x != 0.0f)
					atomicAdd(px + 0, c.x);
				if (c.y != 0.0f)
					atomicAdd(px + 1, c.y);
				if (c.z != 0.0f)
					atomicAdd(px + 2, c.z);
				visible += 1;
			}
			overflow = overflow || st.overflow;
			live = false;
		}
	}
	if (overflow)
		atomicOr(&P.k->device_error, kErrStackOverflow);
	wave_add_u64(&P.k->n_shadow_visible, visible);
	if (COUNT) {
		wave_add_u64(&P.k->nodes_connect, vc.nodes);
		wave_add_u64(&P.k->tris_connect, vc.tris);
	}
}

template <bool COUNT, int STACK_LDS>
static void launch_extend_diag_t(const FrameParams& P, uint32_t maxLive, uint32_t nSurvivors, const Tuning& t, int numCUs, LaunchCache& lc, hipStream_t stream) {
	if (t.traversalVariant == 0) {
		hipLaunchKernelGGL((k_extend<COUNT, STACK_LDS>), dim3(blocks_for(maxLive)), dim3(kBlock), 0, stream, P);
		return;
	}
	launch_extend_spheres(P, nSurvivors, stream);
	hipLaunchKernelGGL((k_extend_persistent<COUNT, STACK_LDS>), dim3(persistent_blocks(k_extend_persistent<COUNT, STACK_LDS>, maxLive, t, numCUs, lc.perCU[COUNT ? kLcDiagExtendCount : kLcDiagExtend][stack_slot(STACK_LDS)])), dim3(kBlock), 0, stream, P);
}
template <bool COUNT, int STACK_LDS>
static void launch_connect_diag_t(const FrameParams& P, uint32_t maxShadow, const Tuning& t, int numCUs, LaunchCache& lc, hipStream_t stream) {
	if (t.traversalVariant == 0) {
		hipLaunchKernelGGL((k_connect<COUNT, STACK_LDS>), dim3(blocks_for(maxShadow)), dim3(kBlock), 0, stream, P);
		return;
	}
	launch_connect_spheres(P, maxShadow, stream);
	hipLaunchKernelGGL((k_connect_persistent<COUNT, STACK_LDS>), dim3(persistent_blocks(k_connect_persistent<COUNT, STACK_LDS>, maxShadow, t, numCUs, lc.perCU[COUNT ? kLcDiagConnectCount : kLcDiagConnect][stack_slot(STACK_LDS)])), dim3(kBlock), 0, stream, P);
}

#define TYR_DISPATCH_STACK(FN, COUNT, ...)          \
	switch (t.stackLdsDepth) {                      \
	case 0: FN<COUNT, 0>(__VA_ARGS__); break;       \
	case 8: FN<COUNT, 8>(__VA_ARGS__); break;       \
	case 10: FN<COUNT, 10>(__VA_ARGS__); break;     \
	case 16: FN<COUNT, 16>(__VA_ARGS__); break;     \
	case 24: FN<COUNT, 24>(__VA_ARGS__); break;     \
	default: FN<COUNT, 12>(__VA_ARGS__); break;     \
	}

void launch_extend_diag(const FrameParams& P, uint32_t maxLive, uint32_t nSurvivors, bool countVisits, const Tuning& t, int numCUs, LaunchCache& lc, hipStream_t stream) {
	if (countVisits) {
		TYR_DISPATCH_STACK(launch_extend_diag_t, true, P, maxLive, nSurvivors, t, numCUs, lc, stream)
	} else {
		TYR_DISPATCH_STACK(launch_extend_diag_t, false, P, maxLive, nSurvivors, t, numCUs, lc, stream)
	}
}
void launch_connect_diag(const FrameParams& P, uint32_t maxShadow, bool countVisits, const Tuning& t, int numCUs, LaunchCache& lc, hipStream_t stream) {
	if (countVisits) {
		TYR_DISPATCH_STACK(launch_connect_diag_t, true, P, maxShadow, t, numCUs, lc, stream)
	} else {
		TYR_DISPATCH_STACK(launch_connect_diag_t, false, P, maxShadow, t, numCUs, lc, stream)
	}
}

} // namespace tyr
