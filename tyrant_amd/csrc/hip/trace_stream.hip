// trace_stream.hip -- the traversal kernel of the STREAMED TAIL (kernels.hpp "the STREAMED TAIL of a render"): extend
// (kernel.cu:331-343) and connect (kernel.cu:630-646) of ALL remaining iterations of a render in one persistent launch.
//
// The per-lane state machine is k_trace_flat's (traverse_flat.hip: quad nodes, the LDS stack, one pop + one quad test per
// lane per trip, leaves between descents) -- same boxes, same order, same accept rule, same answers.  What differs is
// where rays come from and where answers go:
//   * a wave that has lanes to spare draws a 64-slot chunk ticket from the StreamIter it is working on -- first the
//     iteration's work rays (class 0), then the shadow rays of the iteration before -- and takes the chunk once it is READY:
//     fill[chunk] == 64, or the iteration is closed (then the final segment counts say what the chunk holds).  A ticket
//     whose chunk is not ready is kept; the wave goes on with the rays it has.  When an iteration has nothing left for it,
//     the wave moves on to the next: the stragglers of iteration j and the first rays of iteration j + 1 share the grid;
//   * a finished work ray is reported to done[tile] (k_shade_stream shades a tile once all of its rays are reported),
//     a finished shadow ray to its iteration's shadowDone;
//   * the kernel ends at the first iteration that closes without a ray of either kind.
// Every wait is bounded (kStreamTimeoutTicks); a wave only ever waits while it holds no ray.
#include "device_common.hpp"

namespace tyr {

#ifndef TYR_STREAM_STACK
#define TYR_STREAM_STACK 12
#endif

// the lane states of the flat traversal (traverse_flat.hip), as wave-wide masks: one ballot per comparison
static __device__ __forceinline__ unsigned long long lanes_traversing_stream(uint32_t ref) { return __builtin_amdgcn_ballot_w64((int)ref >= 0) | __builtin_amdgcn_ballot_w64(ref == kRefPop); }
static __device__ __forceinline__ unsigned long long lanes_at_leaf_stream(uint32_t ref) { return __builtin_amdgcn_ballot_w64((ref & kRefLeaf) != 0u) & __builtin_amdgcn_ballot_w64(ref < kRefPop); }

template <int STACK_LDS>
__global__ void __launch_bounds__(kBlock, 5) k_trace_stream(const FrameParams P) {
	TYR_DECLARE_FLAT_STACK(st, true)
	__shared__ float4 stagedNodes[7 * kStagedNodes];
	const uint32_t nStaged = P.scene.nStaged;
	for (uint32_t i = threadIdx.x; i < 7 * nStaged; i += kBlock) {
		const uint32_t v = i / nStaged, n = i - v * nStaged;
		stagedNodes[v * kStagedNodes + n] = P.scene.quads[8 * n + v];
	}
	__syncthreads();
	const uint32_t lane = lane_id();
	const unsigned long long below = (1ull << lane) - 1ull;
	const DevScene& sc = P.scene;
	StreamState* const S = P.stream;
	const uint32_t nChunks = P.classStride >> 6; // chunks of one class of a ray queue = chunks of a shadow queue

	// ---- per-lane ray state (k_trace_flat's) ----
	float rox = 0.f, roy = 0.f, roz = 0.f, rdx = 0.f, rdy = 0.f, rdz = 0.f, rix = 0.f, riy = 0.f, riz = 0.f;
	bool regular = true, allRegular = true;
	float dist = 0.0f;
	uint32_t ref = kRefDone;
	uint32_t slot = 0; // queue slot (class 0) of an extend ray, shadow-queue index of a shadow ray; bit 31: parity of its iteration
	int prim = 0;
	bool hitTri = false, live = false, overflow = false;
	bool isShadow = false, occluded = false;
	uint32_t visible = 0;
	constexpr uint32_t kNone = 0xffffffffu;
	uint32_t pendVisible = kNone; // a shadow ray that came through: its colour goes to its pixel at the wave's next refill (slot | parity)
	uint32_t pendDone = kNone;    // a finished work ray not yet reported to done[tile] (slot | parity): reported at the next refill, behind a vmcnt(0)
	uint32_t shadowFinished[2] = { 0u, 0u }; // (wave-uniform) finished shadow rays of either parity, not yet added to their iteration's shadowDone
	bool reportShadows = false;             // (wave-uniform) ... and they are due: the wave has moved on to another iteration

	// ---- per-wave feed state (uniform) ----
	uint32_t j = 0, kind = 0, word = blockIdx.x % kSegs, tried = 0;
	uint32_t cBase = 0, cNext = 0, cEnd = 0, cPar = 0, cKind = 0; // the chunk in hand: slots cBase + [cNext, cEnd)
	bool havePend = false; // a ticket whose chunk was not ready yet
	uint32_t pendRow = 0;
	bool finished = false, failed = false;
	uint32_t holdTrips = 0;              // after a refill that found nothing ready: descent trips before the next try
	unsigned long long tIdle = 0ull;     // since when this wave has had neither a ray nor a ready chunk
	uint32_t idlePolls = 0;

	auto rfl = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
	// true when the chunk in hand is non-empty afterwards
	auto acquire = [&]() -> bool {
		for (;;) { // every turn returns, or moves on to another word / kind / iteration (at most 2 * 8 * kStreamMaxIters turns)
			if (cNext != cEnd)
				return true;
			if (finished)
				return false;
			StreamIter* const I = &S->it[j];
			const uint32_t par = j & 1u;
			if (!havePend) {
				uint32_t r = 0;
				if (lane == 0)
					r = atomicAdd(&I->tick[kind][word * kSegStride], 1u);
				pendRow = rfl(r);
				havePend = true;
			}
			uint32_t* const fill = kind == 0u ? (par ? P.fillNext : P.fillWork) : (par ? P.fillShadow : P.fillShadowPrev);
			const uint32_t c = pendRow * kSegs + word; // the chunk's place among its queue's chunks
			uint32_t f = 0, cl = 0, cnt = 0;
			if (lane == 0) {
				if (c < nChunks)
					f = ld_sc1_u32(&fill[c]);
				if (f != 64u) {
					// (the fill count was read BEFORE this look: while the iteration is open its queue cannot have been handed on)
					cl = ld_sc1_u32(&I->closed);
					if (cl)
						cnt = ld_sc1_u32(kind == 0u ? &I->segWork[0][word] : &I->segShadowPrev[word]);
				}
			}
			f = rfl(f), cl = rfl(cl), cnt = rfl(cnt);
			uint32_t valid = 64u;
			if (f != 64u) {
				if (!cl)
					return false; // neither full nor final: the ticket is kept
				const uint32_t first = pendRow * 64u;
				valid = cnt > first ? (cnt - first < 64u ? cnt - first : 64u) : 0u;
				if (valid == 0u) { // this segment holds nothing more of this kind for this iteration
					havePend = false;
					word = (word + 1u) % kSegs;
					if (++tried == kSegs) {
						tried = 0;
						if (kind == 0u) {
							kind = 1u;
						} else {
							// the end of the render: an iteration that closed without a ray of either kind
							uint32_t nl = 0, ns = 0;
							if (lane == 0) {
								nl = ld_sc1_u32(&I->nLive);
								ns = ld_sc1_u32(&I->nShadowPrev);
							}
							nl = rfl(nl), ns = rfl(ns);
							if ((nl | ns) == 0u || j + 1u >= kStreamMaxIters) {
								finished = true;
								return false;
							}
							kind = 0u;
							++j;
							reportShadows = true; // (iter_of() needs the counts of iteration j - 2's parity out before rays of j's come in)
						}
					}
					continue;
				}
			}
			if (lane == 0 && c < nChunks)
				st_sc1_u32(&fill[c], 0u); // the counter is its reader's to reset (the chunk's next use is two iterations away)
			havePend = false;
			cBase = c * 64u, cNext = 0u, cEnd = valid, cPar = par, cKind = kind;
			return true;
		}
	};
	// the iteration a ray of parity `par` belongs to: rays of iteration j - 2 or older cannot be out any more when a wave
	// works on iteration j (shade(j - 1), which made j's rays, ran behind the scan of j - 2's)
	auto iter_of = [&](uint32_t par) { return (j & 1u) == par ? j : j - 1u; };
	const ShadowQ& sq0 = P.shadowPrev; // the shadow rays traced beside the tail's first iteration (parity 0), then alternating
	const ShadowQ& sq1 = P.shadow;
	// what the wave owes its consumers, paid at a refill (and once after the loop): pixel colours of shadow rays that came
	// through (kernel.cu:640-644), done[tile] for finished work rays, shadowDone for finished shadow rays
	auto settle = [&]() {
		if (__ballot(pendVisible != kNone) != 0ull) {
			float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
			int px = 0;
			if (pendVisible != kNone) {
				const uint32_t idx = pendVisible & 0x7fffffffu;
				const ShadowQ& q = (pendVisible >> 31) ? sq1 : sq0;
				c = ld_sc1_f4(&q.color[idx]);
				px = __float_as_int(ld_sc1_f2(reinterpret_cast<const float2*>(&q.dyz_cd_ix[idx]) + 1).y);
			}
			accumulate_pixels_wave(P.blit, px, mk3(c.x, c.y, c.z), 0);
			pendVisible = kNone;
		}
		unsigned long long owing = __ballot(pendDone != kNone);
		if (owing != 0ull) {
			__asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's answers (stored a pass ago at least) have arrived
			// one add per 64-slot chunk, not per ray: a chunk's rays are all this wave's, and 256 adds to one word are a serial
			// queue (~12 ns each)
			while (owing != 0ull) {
				const uint32_t first = (uint32_t)__ffsll((long long)owing) - 1u;
				const uint32_t chunk = (uint32_t)__builtin_amdgcn_readlane((int)(pendDone >> 6), (int)first); // (the parity rides in bit 25)
				const unsigned long long same = __ballot(pendDone != kNone && (pendDone >> 6) == chunk);
				if (lane == first) {
					uint32_t* const done = (pendDone >> 31) ? P.doneNext : P.doneWork;
					__hip_atomic_fetch_add(&done[(pendDone & 0x7fffffffu) >> 8], (uint32_t)__popcll(same), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
				owing &= ~same;
			}
			pendDone = kNone;
		}
		// finished shadow rays: when the wave has 64 to report, holds no ray, or has moved on to another iteration
		const bool dry = __ballot(live) == 0ull;
#pragma unroll
		for (uint32_t p = 0; p < 2u; ++p) {
			if (shadowFinished[p] != 0u && (shadowFinished[p] >= 64u || dry || reportShadows)) {
				if (lane == 0)
					__hip_atomic_fetch_add(&S->it[iter_of(p)].shadowDone[(blockIdx.x % kSegs) * kSegStride], shadowFinished[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				shadowFinished[p] = 0u;
			}
		}
		reportShadows = false;
	};

	for (;;) {
		if (failed)
			break;
		if (holdTrips != 0u)
			--holdTrips;
		// ---- refill free lanes from whatever chunk is ready ----
		const unsigned long long idleMask = __ballot(!live);
		const uint32_t nIdle = __popcll(idleMask);
		if (!finished && nIdle >= P.refillMinIdle && holdTrips == 0u) {
			const uint32_t rank = __popcll(idleMask & below);
			uint32_t s = 0, got = 0;
			bool fed = false, fedShadow = false;
			uint32_t fedPar = 0;
			while (got < nIdle) {
				if (!acquire())
					break;
				const uint32_t avail = cEnd - cNext, room = nIdle - got;
				const uint32_t take = avail < room ? avail : room;
				if (!live && rank >= got && rank < got + take) {
					s = cBase + cNext + (rank - got);
					fed = true;
					fedShadow = cKind != 0u;
					fedPar = cPar;
				}
				cNext += take;
				got += take;
			}
			settle();
			if (got == 0u) {
				if (__ballot(live) == 0ull) {
					if (finished)
						break;
					// nothing to do and nothing ready: wait a little (bounded by the wall clock)
					if (tIdle == 0ull)
						tIdle = __builtin_amdgcn_s_memrealtime();
					// (a long sleep: four idle waves per SIMD polling at a microsecond's pace took most of the issue slots the
					// shade wave beside them -- the producer they are waiting for -- needed: 3-5x slower tiles, measured)
					__builtin_amdgcn_s_sleep(127);
					if ((++idlePolls & 7u) == 0u) {
						uint32_t err = 0;
						if (lane == 0)
							err = ld_sc1_u32(&P.k->device_error);
						if (rfl(err) != 0u || __builtin_amdgcn_s_memrealtime() - tIdle > kStreamTimeoutTicks) {
							if (lane == 0)
								atomicOr(&P.k->device_error, kErrNoProgress);
							failed = true;
						}
					}
					continue;
				}
				holdTrips = 8u; // go on with the rays in hand for a few trips before asking again
			} else {
				tIdle = 0ull;
			}
			if (fed) {
				float4 a;
				float by, bz, bound;
				bool blocked = false;
				if (!fedShadow) { // a work ray: closest hit
					const RayQ& q = fedPar ? P.next : P.work;
					a = ld_sc1_f4(&q.o_dx[s]);
					const float2 b = ld_sc1_f2(&q.dyz[s]);
					const float2 h = ld_sc1_f2(&q.hit[s]);
					by = b.x, bz = b.y, bound = h.x; // the sphere half's distance (its producer's work) bounds the search
				} else { // a shadow ray: any hit within closestDistance
					const ShadowQ& q = fedPar ? sq1 : sq0;
					a = ld_sc1_f4(&q.o_dx[s]);
					const float4 b = ld_sc1_f4(&q.dyz_cd_ix[s]);
					blocked = ld_sc1_f2(reinterpret_cast<const float2*>(&q.color[s]) + 1).y != 0.0f; // the sphere half's verdict
					by = b.x, bz = b.y, bound = b.z;
				}
				slot = s | (fedPar << 31);
				isShadow = fedShadow;
				const RayConst nr = make_ray(mk3(a.x, a.y, a.z), mk3(a.w, by, bz));
				rox = nr.o.x, roy = nr.o.y, roz = nr.o.z, rdx = nr.d.x, rdy = nr.d.y, rdz = nr.d.z, rix = nr.inv.x, riy = nr.inv.y, riz = nr.inv.z;
				regular = ray_is_regular(nr);
				dist = bound;
				hitTri = false;
				occluded = blocked;
				st.reset();
				ref = blocked ? kRefDone : root_ref(sc, nr, dist);
				if (ref != kRefDone)
					ref = sc.quadRootRef;
				// a ray that ends here (a work ray that misses the root box: its producer's answer stands; a shadow ray a
				// sphere blocks) is reported below like any other finished ray
				live = true;
			}
			if (!finished && (uint32_t)__popcll(__ballot(live && ref != kRefDone)) < P.minTraversing)
				if (__ballot(live && ref == kRefDone) == 0ull && got != 0u)
					continue; // mostly short rays: top the wave up again first
		}
		if (__ballot(live) == 0ull) {
			if (finished)
				break;
			holdTrips = 0u;
			continue;
		}
		allRegular = (__ballot(live && !regular) == 0ull);
		const RayConst r = { mk3(rox, roy, roz), mk3(rdx, rdy, rdz), mk3(rix, riy, riz), rix < 0, riy < 0, riz < 0 }; // bvh.h:120-121
		// ---- descent: one pop attempt + one quad test per lane per trip ----
		for (;;) {
			const uint32_t nTrav = __popcll(lanes_traversing_stream(ref));
			if (nTrav == 0)
				break;
			if (holdTrips != 0u)
				--holdTrips;
			if (nTrav < P.minTraversing) {
				const bool anyLeaf = lanes_at_leaf_stream(ref) != 0ull;
				const bool canRefill = !finished && holdTrips == 0u && (uint32_t)__popcll(__ballot(!live || ref == kRefDone)) >= P.refillMinIdle;
				if (anyLeaf || canRefill)
					break;
			}
			if (ref == kRefPop) {
				uint32_t pr;
				float pt;
				if (st.pop(pr, pt)) {
					if (pt < dist) // the pop-time half of Bbox.h:61
						ref = pr;
				} else {
					ref = kRefDone;
				}
			}
			if ((int)ref >= 0) {
				const QuadHits q = allRegular ? test_quad<true, true, true>(sc.quads, ref, r, dist, stagedNodes, nStaged) : test_quad<false, true, true>(sc.quads, ref, r, dist, stagedNodes, nStaged);
				const lanemask any01 = q.hit[0] | q.hit[1], any012 = any01 | q.hit[2];
				st.push3(q.hit[3] & any012, q.ref[3], q.t[3], q.hit[2] & any01, q.ref[2], q.t[2], q.hit[1] & q.hit[0], q.ref[1], q.t[1]);
				ref = lane_in(q.hit[0]) ? q.ref[0] : lane_in(q.hit[1]) ? q.ref[1] : lane_in(q.hit[2]) ? q.ref[2] : lane_in(q.hit[3]) ? q.ref[3] : kRefPop;
			}
		}
		// ---- leaves: bvh.h:129-140 (closest hit) / bvh.h:229-238 (any hit) ----
		if ((ref & kRefLeaf) && ref < kRefPop) {
			const uint32_t off = ref & (kMaxPrimOffset - 1);
			const uint32_t cnt = ((ref >> 26) & 31u) + 1u;
			bool found = false;
			TriData tri = triangle_load(sc.tris, off);
			for (uint32_t i = 0; i < cnt && !found; ++i) {
				const TriData cur = tri; // the next primitive of the leaf is on its way while this one is tested
				if (i + 1 < cnt)
					tri = triangle_load(sc.tris, off + i + 1);
				const float t = triangle_test(cur, r);
				if (isShadow) {
					found = (t > kEpsilon && ((dist - t) > kEpsilon)); // bvh.h:232-236
				} else if (t > kEpsilon && t < dist && ((dist - t) > kEpsilon)) {
					prim = (int)(off + i);
					dist = t;
					hitTri = true;
				}
			}
			occluded = occluded || found;
			ref = found ? kRefDone : kRefPop;
		}
		// ---- finished rays ----
		{
			const unsigned long long fin = __ballot(live && ref == kRefDone && isShadow);
			if (fin != 0ull) { // (scalar bookkeeping: the wave's count per parity)
				const unsigned long long odd = __ballot((slot >> 31) != 0u);
				shadowFinished[0] += (uint32_t)__popcll(fin & ~odd);
				shadowFinished[1] += (uint32_t)__popcll(fin & odd);
			}
		}
		if (live && ref == kRefDone) {
			const uint32_t par = slot >> 31, idx = slot & 0x7fffffffu;
			if (isShadow) {
				if (!occluded) { // kernel.cu:640-644, added to the pixel at the wave's next refill
					pendVisible = slot;
					visible += 1;
				}
			} else {
				if (hitTri) // a triangle hit replaces the sphere answer (kernel.cu:138-140); write-through: k_shade_stream reads it beside us
					st_sc1_f2(&(par ? P.next : P.work).hit[idx], make_float2(dist, __uint_as_float((uint32_t)prim)));
				pendDone = slot;
			}
			overflow = overflow || st.overflow;
			live = false;
		}
	}
	if (!failed)
		settle();
	if (overflow)
		atomicOr(&P.k->device_error, kErrStackOverflow);
	wave_add_u64(&P.k->n_shadow_visible, visible);
}

void launch_trace_stream(const FrameParams& P, int blocksPerCU, int numCUs, hipStream_t stream) {
	const uint32_t blocks = (uint32_t)(blocksPerCU < 1 ? 1 : blocksPerCU) * (uint32_t)numCUs;
	hipLaunchKernelGGL((k_trace_stream<TYR_STREAM_STACK>), dim3(blocks), dim3(kBlock), 0, stream, P);
}

} // namespace tyr
