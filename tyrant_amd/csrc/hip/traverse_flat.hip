// traverse_flat.hip -- extend (kernel.cu:331-343 via intersect_scene, kernel.cu:125-142) and connect
// (kernel.cu:630-646 via intersect_scene_simple, kernel.cu:162-174) as flat per-lane state machines on persistent
// waves.  Production = k_trace_flat: quad nodes on a persistent grid; the counting build (TYR_FLAG_COUNT_VISITS) = the
// same state machine on pair nodes (k_extend_count / k_connect_count), which reproduces the reference's visit counts
// (bvh.h:164-209).  
#include "device_common.hpp"
#include "scan_wave.hpp"

namespace tyr {

// Register budget of the flat traversal kernels, as waves per SIMD.  Five (<= 96 VGPRs) measured 7-10 % faster in
// extend than the four the allocator picks by itself (99 VGPRs); six (80 VGPRs) spills 32 registers in the descent
// loop and is 40 % slower.  Deeper LDS stacks cap the occupancy below five anyway.
// Threads per block of k_trace_flat (template parameter): 256 -- five blocks per CU, five waves per SIMD -- or 768 = three
// 256-thread parts that share ONE copy of the staged nodes: two such blocks per CU are SIX waves per SIMD with all 64 staged
// nodes (2 x (73,728 B of stacks + 7,168 + 4) = the CU's 160 KB in 1,280-byte granules; 75 vector registers under that
// bound, no spill).  The sixth wave feeds a fat launch faster (-2.3 % per C3 render at 16.6 M slots) and lengthens the drain
// of a thin one (+0.7 % at 2 Mi slots): launch_trace_kernel picks by the launch's item count (Tuning::wideBlockMinItems).
constexpr uint32_t kTraceBlockWide = 768;
#define TYR_FLAT_WAVES_PER_EU (STACK_LDS <= 8 ? 6 : STACK_LDS <= 12 ? 5 : STACK_LDS <= 16 ? 3 : 2)

// ======================================================================================
// Flat traversal: persistent waves, lane refill, and NO nested divergent loops.
//
// Measured on the first persistent kernel (nested while loops) with the counting build (tools/loop_occupancy.py, C2 at 1080p): the
// node-test loop ran at 19.7 % lane occupancy and the nested pop loop at 6 %, because (a) a lane
// that reaches a leaf or finishes its ray waits until the LAST lane of the wave stops descending,
// and (b) `while (pop) {...}` inside the divergent "both children missed" branch runs four lanes
// wide while sixty wait.  Here every lane is a small state machine --
//      interior ref | leaf ref | kRefPop (must pop) | kRefDone --
// and one trip of the descent loop does at most ONE pop attempt and ONE pair test per lane, so
// lanes in different states advance together.  The descent loop is left as soon as fewer than
// `minTraversing` lanes are still descending and there is other work for the wave (leaves to
// intersect, or enough free lanes for a refill).
// ======================================================================================
__device__ __forceinline__ bool ref_is_leaf(uint32_t ref) { return (ref & kRefLeaf) && ref < kRefPop; }
__device__ __forceinline__ bool ref_is_traversing(uint32_t ref) { return ((int)ref >= 0) || ref == kRefPop; }
// the same as wave-wide masks, one ballot per comparison (the ballot of a compound condition goes through a 0/1
// VGPR and a second comparison, see slab_fast_mask)
__device__ __forceinline__ unsigned long long lanes_traversing(uint32_t ref) { return __builtin_amdgcn_ballot_w64((int)ref >= 0) | __builtin_amdgcn_ballot_w64(ref == kRefPop); }
__device__ __forceinline__ unsigned long long lanes_at_leaf(uint32_t ref) { return __builtin_amdgcn_ballot_w64((ref & kRefLeaf) != 0u) & __builtin_amdgcn_ballot_w64(ref < kRefPop); }

#ifdef TYR_QUAD_STATS
constexpr bool kLoopStats = true; // diagnostic build: the production (quad) kernel fills tyr_counters.debug too, tools/loop_occupancy.py
#else
constexpr bool kLoopStats = false;
#endif
#if defined(TYR_QUAD_STATS) || defined(TYR_LAUNCH_ANATOMY)
constexpr bool kAnatomy = true; // three s_memrealtime stamps per wave (start, queue used up, exit): tools/launch_tail.py
#else
constexpr bool kAnatomy = false;
#endif
#define TYR_DBG(i)                                                     \
	if (COUNT || kLoopStats) {                                         \
		const unsigned long long m_ = __ballot(1);                     \
		if (lane == (uint32_t)__ffsll((long long)m_) - 1) {            \
			dbg[i] += 1;                                               \
			dbg[i + 1] += __popcll(m_);                                \
		}                                                              \
	}

// Variant 4 work distribution: a PERSISTENT grid (as many blocks as stay resident) whose waves each own a private
// range of queue slots and draw the next chunk from one of kTicketWords device-wide tickets when it runs out.
// Chunk c of the queue belongs to ticket word c % kTicketWords; a wave starts at word blockIdx % kTicketWords
// (its XCD under round-robin placement) and moves on to the next word when one is used up, so the last chunks
// are shared by whoever is free.  Compared with block-owned ranges (variant 3) there is no per-block tail: the
// four waves of a block never wait for the block's longest ray, only the end of the launch has partly filled waves.
// -DTYR_GUARD_PASSES (make EXTRA_HIPFLAGS=...): an exit condition every wave of the flat traversal kernels reaches
// whatever the feed logic does -- an outer pass (refill + descent + leaves) takes at least ~0.1 us and a launch a
// few milliseconds, so 2^24 passes are never seen by a working build; a wave that gets there gives up and reports
// kErrNoProgress instead of holding the GPU.  For work on the refill / exit logic (one mistake there is a hung GPU);
// off in the shipped build, where the counter and its branch cost 2 % of extend (measured), and the logic is what
// the soak and fuzz runs of profiles/ exercised.
#ifdef TYR_GUARD_PASSES
constexpr bool kGuardPasses = true;
#else
constexpr bool kGuardPasses = false;
#endif
constexpr uint32_t kMaxPasses = 1u << 24;

struct ChunkFeed {
	uint32_t next, end;   // this wave's private range of queue slots (wave-uniform)
	uint32_t word, tried; // ticket word in use, words found empty so far
	uint32_t chunk;       // slots per draw
	__device__ __forceinline__ void init(uint32_t nItems, uint32_t chunkWanted) {
		next = end = 0;
		word = blockIdx.x % kTicketWords;
		tried = 0;
		// thin queues: smaller chunks, so that the rays spread over more CUs (never below one wave's worth).
		// What the sweeps said (profiles/r01_chunk_feed_sweep.txt): a draw must be ONE round trip -- with a look
		// at the word before every atomic, launches of short rays (the primary rays) were 30-60 % slower than
		// block-owned ranges; with that gone, 64- and 128-slot chunks beat larger ones by 2-4 %.  Guided
		// (shrinking) draws over 64-slot granules lost to fixed chunks.
		const uint32_t waves = gridDim.x * (blockDim.x / 64u);
		chunk = chunkWanted;
		while (chunk > 64 && (unsigned long long)waves * chunk > nItems)
			chunk >>= 1;
	}
	// true when [next, end) is non-empty afterwards
	__device__ __forceinline__ bool refill(uint32_t* tickets, uint32_t nItems, uint32_t lane) {
		while (next == end && tried < kTicketWords) {
			uint32_t t = 0;
			if (lane == 0) {
				uint32_t* w = tickets + word * 32;
				// once a word has been found empty, look before drawing: at the end of a launch every wave walks all the
				// words, and plain reads are served in parallel (an atomic on one word is not).  Before that, draw
				// straight away -- a look first would double the round trip of every draw.
				bool draw = true;
				if (tried != 0) {
					t = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					draw = (unsigned long long)(t * kTicketWords + word) * chunk < nItems;
				}
				if (draw)
					t = atomicAdd(w, 1u);
			}
			t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
			const unsigned long long start = (unsigned long long)(t * kTicketWords + word) * chunk;
			if (start < nItems) {
				next = (uint32_t)start;
				end = (start + chunk < nItems) ? (uint32_t)(start + chunk) : nItems;
			} else {
				word = (word + 1) % kTicketWords;
				++tried;
			}
		}
		return next != end;
	}
};

// a closest-hit ray is finished: a triangle hit replaces the sphere answer of the pre-pass (kernel.cu:138-140)
__device__ __forceinline__ void finish_extend_ray(float2* hit, uint32_t slot, bool hitTri, float dist, int prim) {
	if (hitTri)
		hit[slot] = make_float2(dist, __uint_as_float((uint32_t)prim));
}

// slots of the queue that each block of a persistent grid owns outright (a multiple of 64; 0 for thin queues)
__device__ __forceinline__ uint32_t static_range(uint32_t nItems, uint32_t sixteenths) {
	const unsigned long long share = (unsigned long long)nItems * sixteenths / 16ull;
	return (uint32_t)(share / gridDim.x) & ~63u;
}

// Which queue slot the `local`-th slot of a block's fixed range is.  Contiguous ranges (block b = slots
// [b * perBlock, +perBlock)) hand whole image regions to single blocks -- the queue is in scan-line order, and so are the
// survivors inside every generation -- and a block that drew the dense part of the frame is still inside its fixed range
// when the ticketed rest of the queue has long been used up by the others: the launch anatomy (tools/launch_tail.py)
// showed ~330 us between the first wave finding the tickets gone and the last wave's exit, also for the short primary
// rays.  Interleaved, block b owns the 64-slot chunks b, b + G, b + 2 G, ... of the fixed part: every block gets the
// same mix.  (perBlock is a multiple of 64.)
__device__ __forceinline__ uint32_t static_slot(uint32_t s, uint32_t blockBegin, bool interleave) {
	if (!interleave)
		return s;
	const uint32_t local = s - blockBegin;
	return (((local >> 6) * gridDim.x + blockIdx.x) << 6) + (local & 63u);
}

// ======================================================================================
// The counting build (TYR_FLAG_COUNT_VISITS): extend and connect as launches of their own on PAIR nodes -- the only
// layout that reproduces the reference's visit counts (intersect_debug's rule, bvh.h:164-209: every fetched node counts,
// a leaf's triangles count one each) -- with the same flat per-lane state machine as the production kernel.  Block b owns
// the physical slots [b * raysPerBlock, +raysPerBlock) of the queue and hands them to the free lanes of its four waves
// through a counter in LDS; slots that hold no record are skipped (slot_valid).  Not a hot path: n-bar of the roofline
// comes from ONE untimed render of this build.
// ======================================================================================
template <int STACK_LDS>
__global__ void __launch_bounds__(kBlock, TYR_FLAT_WAVES_PER_EU) k_extend_count(const FrameParams P) {
	constexpr bool COUNT = true; // (TYR_DBG)
	TYR_DECLARE_FLAT_STACK(st, true)
	const uint32_t lane = lane_id();
	const unsigned long long below = (1ull << lane) - 1ull;
	const uint32_t nSlots = queue_extent(P.segWork); // class 0; the few slots that hold no record are skipped
	const DevScene& sc = P.scene;
	float rox = 0.f, roy = 0.f, roz = 0.f, rdx = 0.f, rdy = 0.f, rdz = 0.f, rix = 0.f, riy = 0.f, riz = 0.f;
	bool regular = true, allRegular = true;
	float dist = 0.0f;
	uint32_t ref = kRefDone, slot = 0;
	int prim = 0;
	bool hitTri = false, live = false, overflow = false;
	VisitCount vc{ 0, 0 };
	uint32_t inTree = 0; // rays of this lane that passed the root box
	uint32_t dbg[16] = {};
	__shared__ uint32_t blockNext;
	const uint32_t blockBegin = blockIdx.x * P.raysPerBlock;
	const uint32_t blockEnd = (blockBegin + P.raysPerBlock) < nSlots ? (blockBegin + P.raysPerBlock) : nSlots;
	if (threadIdx.x == 0)
		blockNext = blockBegin;
	__syncthreads();
	bool exhausted = (sc.rootRef == kRefDone) || blockBegin >= nSlots;
	uint32_t passes = 0; // see kMaxPasses

	for (;;) {
		if (kGuardPasses && ++passes > kMaxPasses)
			break;
		// ---- refill free lanes from the block's range ----
		const unsigned long long idleMask = __ballot(!live);
		const uint32_t nIdle = __popcll(idleMask);
		if (!exhausted && nIdle >= P.refillMinIdle) {
			const uint32_t rank = __popcll(idleMask & below);
			uint32_t base = 0;
			if (lane == 0)
				base = atomicAdd(&blockNext, nIdle); // LDS
			base = __shfl(base, 0, 64);
			const uint32_t avail = base < blockEnd ? blockEnd - base : 0u;
			const uint32_t take = avail < nIdle ? avail : nIdle;
			exhausted = (base + nIdle >= blockEnd);
			const uint32_t s = base + rank;
			bool fed = !live && rank < take;
			if (fed)
				fed = slot_valid(P.segWork, s);
			if (fed) {
				TYR_DBG(6)
				const float4 a = P.work.o_dx[s];
				const float2 b = P.work.dyz[s];
				const float2 h = P.work.hit[s];
				const RayConst nr = make_ray(mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y));
				rox = nr.o.x, roy = nr.o.y, roz = nr.o.z, rdx = nr.d.x, rdy = nr.d.y, rdz = nr.d.z, rix = nr.inv.x, riy = nr.inv.y, riz = nr.inv.z;
				regular = ray_is_regular(nr);
				dist = h.x; // the sphere pre-pass's distance bounds the search
				slot = s;
				hitTri = false;
				st.reset();
				ref = root_ref(sc, nr, dist);
				// a ray that misses the root box (or is already stopped short of it by a sphere) is finished here
				live = (ref != kRefDone);
				vc.nodes += 1;
				inTree += live ? 1u : 0u;
			}
			if (!exhausted && (uint32_t)__popcll(__ballot(live)) < P.minTraversing)
				continue;
		}
		if (__ballot(live) == 0ull) {
			if (exhausted)
				break;
			continue;
		}
		allRegular = (__ballot(live && !regular) == 0ull);
		const RayConst r = { mk3(rox, roy, roz), mk3(rdx, rdy, rdz), mk3(rix, riy, riz), rix < 0, riy < 0, riz < 0 }; // bvh.h:120-121
		// ---- descent: one pop attempt + one pair test per lane per trip ----
		for (;;) {
			const uint32_t nTrav = __popcll(lanes_traversing(ref));
			if (nTrav == 0)
				break;
			if (nTrav < P.minTraversing) {
				const bool anyLeaf = lanes_at_leaf(ref) != 0ull;
				const bool canRefill = !exhausted && (uint32_t)__popcll(__ballot(!live || ref == kRefDone)) >= P.refillMinIdle;
				if (anyLeaf || canRefill)
					break;
			}
			if (ref == kRefPop) {
				TYR_DBG(2)
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
				TYR_DBG(0)
				const PairTest p = allRegular ? test_pair_fast(sc.nodes, ref, r, dist) : test_pair(sc.nodes, ref, r, dist);
				if (!p.synthetic)
					vc.nodes += 2;
				if (p.nearHit) {
					if (p.farHit)
						st.push(p.farRef, p.farT);
					ref = p.nearRef;
				} else if (p.farHit) {
					ref = p.farRef;
				} else {
					ref = kRefPop;
				}
			}
		}
		// ---- leaves: bvh.h:129-140 ----
		if (ref_is_leaf(ref)) {
			const uint32_t off = ref & (kMaxPrimOffset - 1);
			const uint32_t cnt = ((ref >> 26) & 31u) + 1u;
			TriData tri = triangle_load(sc.tris, off);
			for (uint32_t i = 0; i < cnt; ++i) {
				TYR_DBG(4)
				const TriData cur = tri;
				if (i + 1 < cnt)
					tri = triangle_load(sc.tris, off + i + 1);
				const float t = triangle_test(cur, r);
				vc.tris += 1;
				if (t > kEpsilon && t < dist && ((dist - t) > kEpsilon)) {
					prim = (int)(off + i);
					dist = t;
					hitTri = true;
				}
			}
			ref = kRefPop;
		}
		// ---- finished rays ----
		if (live && ref == kRefDone) {
			finish_extend_ray(P.work.hit, slot, hitTri, dist, prim);
			overflow = overflow || st.overflow;
			live = false;
		}
	}
	if (overflow)
		atomicOr(&P.k->device_error, kErrStackOverflow);
	if (kGuardPasses && passes > kMaxPasses)
		atomicOr(&P.k->device_error, kErrNoProgress);
	// the rays of class 1 never come here: the reference tests the root box for each of them once and stops (bvh.h:127)
	if (blockIdx.x == 0 && threadIdx.x == 0)
		atomicAdd(&P.k->nodes_extend, (unsigned long long)queue_records(P.segWork + kClassWords));
	wave_add_u64(&P.k->nodes_extend, vc.nodes);
	wave_add_u64(&P.k->tris_extend, vc.tris);
	wave_add_u64(&P.k->rays_in_tree_extend, inTree);
	for (int i = 0; i < 13; ++i)
		wave_add_u64(&P.k->debug[i], dbg[i]);
}

template <int STACK_LDS>
__global__ void __launch_bounds__(kBlock, TYR_FLAT_WAVES_PER_EU) k_connect_count(const FrameParams P) {
	TYR_DECLARE_FLAT_STACK(st, true) // (the counting path marks failed boxes through the entry distance)
	const uint32_t lane = lane_id();
	const unsigned long long below = (1ull << lane) - 1ull;
	const uint32_t nSlots = queue_extent(P.kc->seg); // physical slots of the shadow queue
	const DevScene& sc = P.scene;
	const bool haveBvh = (sc.rootRef != kRefDone);
	float rox = 0.f, roy = 0.f, roz = 0.f, rdx = 0.f, rdy = 0.f, rdz = 0.f, rix = 0.f, riy = 0.f, riz = 0.f;
	bool regular = true, allRegular = true;
	float closest = 0.0f;
	uint32_t ref = kRefDone, index = 0;
	bool live = false, occluded = false, overflow = false;
	VisitCount vc{ 0, 0 };
	uint32_t inTree = 0;
	uint32_t visible = 0;
	// kernel.cu:640-644, deferred: a lane whose ray came through unoccluded notes the slot; the wave adds all such colours
	// to their pixels at its next refill (and once after the loop), transposed (accumulate_pixels_wave)
	constexpr uint32_t kNoPending = 0xffffffffu;
	uint32_t pendIdx = kNoPending;
	auto flush_visible = [&]() {
		float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
		int px = 0;
		if (pendIdx != kNoPending) {
			c = P.shadow.color[pendIdx];
			px = __float_as_int(P.shadow.dyz_cd_ix[pendIdx].w);
		}
		accumulate_pixels_wave(P.blit, px, mk3(c.x, c.y, c.z), 0);
		pendIdx = kNoPending;
	};
	__shared__ uint32_t blockNext;
	const uint32_t blockBegin = blockIdx.x * P.raysPerBlock;
	const uint32_t blockEnd = (blockBegin + P.raysPerBlock) < nSlots ? (blockBegin + P.raysPerBlock) : nSlots;
	if (threadIdx.x == 0)
		blockNext = blockBegin;
	__syncthreads();
	bool exhausted = blockBegin >= nSlots;
	const float kFailed = __builtin_inff();
	uint32_t passes = 0; // see kMaxPasses

	for (;;) {
		if (kGuardPasses && ++passes > kMaxPasses)
			break;
		const unsigned long long idleMask = __ballot(!live);
		const uint32_t nIdle = __popcll(idleMask);
		if (!exhausted && nIdle >= P.refillMinIdle) {
			const uint32_t rank = __popcll(idleMask & below);
			uint32_t base = 0;
			if (lane == 0)
				base = atomicAdd(&blockNext, nIdle); // LDS
			base = __shfl(base, 0, 64);
			const uint32_t avail = base < blockEnd ? blockEnd - base : 0u;
			const uint32_t take = avail < nIdle ? avail : nIdle;
			exhausted = (base + nIdle >= blockEnd);
			const uint32_t s = base + rank;
			bool fed = !live && rank < take;
			if (__ballot(pendIdx != kNoPending) != 0ull)
				flush_visible();
			if (fed)
				fed = slot_valid(P.kc->seg, s);
			if (fed) {
				const float4 a = P.shadow.o_dx[s];
				const float4 b = P.shadow.dyz_cd_ix[s];
				const float sphereOccluded = reinterpret_cast<const float*>(&P.shadow.color[s])[3];
				index = s;
				closest = b.z;
				occluded = (sphereOccluded != 0.0f);
				live = true;
				st.reset();
				ref = kRefDone;
				if (haveBvh) { // (the counting rule traverses the BVH whatever the spheres said: kernel.cu:164-172 tests it first)
					const RayConst nr = make_ray(mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y));
					rox = nr.o.x, roy = nr.o.y, roz = nr.o.z, rdx = nr.d.x, rdy = nr.d.y, rdz = nr.d.z, rix = nr.inv.x, riy = nr.inv.y, riz = nr.inv.z;
					regular = ray_is_regular(nr);
					ref = root_ref(sc, nr, closest);
					vc.nodes += 1;
					inTree += (ref != kRefDone) ? 1u : 0u;
				}
			}
		}
		if (__ballot(live) == 0ull) {
			if (exhausted)
				break;
			continue;
		}
		allRegular = (__ballot(live && !regular) == 0ull);
		const RayConst r = { mk3(rox, roy, roz), mk3(rdx, rdy, rdz), mk3(rix, riy, riz), rix < 0, riy < 0, riz < 0 }; // bvh.h:120-121
		for (;;) {
			const uint32_t nTrav = __popcll(lanes_traversing(ref));
			if (nTrav == 0)
				break;
			if (nTrav < P.minTraversing) {
				const bool anyLeaf = lanes_at_leaf(ref) != 0ull;
				const bool canRefill = !exhausted && (uint32_t)__popcll(__ballot(!live || ref == kRefDone)) >= P.refillMinIdle;
				if (anyLeaf || canRefill)
					break;
			}
			if (ref == kRefPop) {
				uint32_t pr;
				float pt;
				if (st.pop(pr, pt)) {
					vc.nodes += 1; // the reference fetches the popped node before testing its box (bvh.h:222-224)
					if (pt < closest)
						ref = pr;
				} else {
					ref = kRefDone;
				}
			}
			if ((int)ref >= 0) {
				const PairTest p = allRegular ? test_pair_fast(sc.nodes, ref, r, closest) : test_pair(sc.nodes, ref, r, closest);
				if (!p.synthetic) {
					vc.nodes += 1;
					st.push(p.farRef, p.farHit ? p.farT : kFailed);
					ref = p.nearHit ? p.nearRef : kRefPop;
				} else if (p.nearHit) {
					if (p.farHit)
						st.push(p.farRef, p.farT);
					ref = p.nearRef;
				} else if (p.farHit) {
					ref = p.farRef;
				} else {
					ref = kRefPop;
				}
			}
		}
		if (ref_is_leaf(ref)) {
			const uint32_t off = ref & (kMaxPrimOffset - 1);
			const uint32_t cnt = ((ref >> 26) & 31u) + 1u;
			bool found = false;
			TriData tri = triangle_load(sc.tris, off);
			for (uint32_t i = 0; i < cnt && !found; ++i) {
				const TriData cur = tri; // next record in flight while this one is tested
				if (i + 1 < cnt)
					tri = triangle_load(sc.tris, off + i + 1);
				const float t = triangle_test(cur, r);
				vc.tris += 1;
				found = (t > kEpsilon && ((closest - t) > kEpsilon)); // bvh.h:232-236
			}
			if (found) {
				occluded = true;
				ref = kRefDone;
			} else {
				ref = kRefPop;
			}
		}
		if (live && ref == kRefDone) {
			if (!occluded) {
				pendIdx = index;
				visible += 1;
			}
			overflow = overflow || st.overflow;
			live = false;
		}
	}
	flush_visible();
	if (overflow)
		atomicOr(&P.k->device_error, kErrStackOverflow);
	if (kGuardPasses && passes > kMaxPasses)
		atomicOr(&P.k->device_error, kErrNoProgress);
	wave_add_u64(&P.k->n_shadow_visible, visible);
	wave_add_u64(&P.k->nodes_connect, vc.nodes);
	wave_add_u64(&P.k->tris_connect, vc.tris);
	wave_add_u64(&P.k->rays_in_tree_connect, inTree);
}

// LDS executes a wave's instructions in order; this only keeps the COMPILER from moving a lane's LDS read above another
// lane's write of the same wave (no instruction is emitted)
__device__ __forceinline__ void wave_lds_order() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// values of the other lanes of this lane's aligned group of four (DPP quad_perm: no LDS, no memory)
template <int CTRL>
__device__ __forceinline__ uint32_t quad_perm_u(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t quad_or(uint32_t v) {
	v |= quad_perm_u<0xB1>(v); // [1,0,3,2]
	v |= quad_perm_u<0x4E>(v); // [2,3,0,1]
	return v;
}
template <int K>
__device__ __forceinline__ float quad_bcast_f(float v) { return __uint_as_float(quad_perm_u<K * 0x55>(__float_as_uint(v))); }
// wide_drain is a function of its own (below): its pointer arguments arrive as GENERIC pointers, and a load through one is
// a flat load -- it counts on vmcnt AND lgkmcnt, so that every wait for a stack entry in LDS also waits for the records in
// flight.  These say what the kernel knows: the scene and the queues are global memory.
#define TYR_GLOBAL __attribute__((address_space(1)))
typedef float v2f_t __attribute__((ext_vector_type(2)));
typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float2 gload_f2(const float* p) {
	const v2f_t v = *(const TYR_GLOBAL v2f_t*)p;
	return make_float2(v.x, v.y);
}
__device__ __forceinline__ float4 gload_f4(const float4* p) {
	const v4f_t v = *(const TYR_GLOBAL v4f_t*)p;
	return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t gload_u(const float* p) { return *(const TYR_GLOBAL uint32_t*)p; }
__device__ __forceinline__ TriData triangle_gload(const float4* tris, uint32_t prim) {
	TriData d;
	d.a = gload_f4(tris + 3 * prim + 0);
	d.b = gload_f4(tris + 3 * prim + 1);
	d.c = make_float4(__uint_as_float(gload_u(reinterpret_cast<const float*>(tris + 3 * prim + 2))), 0.0f, 0.0f, 0.0f); // (the tests read c.x only)
	return d;
}
constexpr uint32_t kWideRays = 16;        // a wave switches to four lanes per ray once it holds at most this many
constexpr int kWideStackEntries = 4 * 12; // the four lanes' LDS columns of a group, as one stack

// ==== the drain, four lanes to a ray ====
		// Once the queue is used up a wave finishes its last rays a few lanes wide while every step still costs a full wave's
	// instructions: ~330 per quad step (four box tests, ordering, up to three pushes, the leaf's primitives one trip each),
	// five such waves to a SIMD -- the first third of a launch's drain is bound by exactly that.  Here the wave's (at most
	// 16) rays are dealt one to each aligned group of four lanes: a lane tests ONE child box of the quad node (and one
	// primitive of a leaf), the four answers meet through DPP quad_perm moves, every lane of the group derives the same
	// visit order test_quad derives, and the group's stack is its four LDS columns taken as one.  Same boxes, same order,
	// same accept rule, same answers -- at a quarter of the instructions per step.
// (A function of its own, not inlined: inside k_trace_flat its scalar registers competed with the feed loop's -- 35 instead
// of 16 spilled there, 2.5 % of a render whether or not a wave ever got here.)
#ifdef TYR_LAUNCH_ANATOMY
#define TYR_WIDE_STEPS_PARAM , uint32_t& wideSteps
#define TYR_WIDE_STEPS_ARG , wideSteps
#else
#define TYR_WIDE_STEPS_PARAM
#define TYR_WIDE_STEPS_ARG
#endif
struct WideState {
	float rox, roy, roz, rdx, rdy, rdz, rix, riy, riz, dist;
	uint32_t ref, slot, flags; // flags: 1 regular, 2 hitTri, 4 isShadow, 8 occluded, 16 live
	int prim, n;
};
template <int STACK_LDS>
__device__ __attribute__((noinline, cold)) uint32_t wide_drain(const float4* __restrict__ quads, const float4* __restrict__ tris, const float4* __restrict__ shadowColor, const float4* __restrict__ shadowDyzCdIx,
                                                         float2* __restrict__ workHit, float4* __restrict__ blit, typename LdsStack<STACK_LDS, true>::entry_t* smem_, WideState w, uint32_t passes TYR_WIDE_STEPS_PARAM) {
	const uint32_t lane = lane_id();
	const unsigned long long below = (1ull << lane) - 1ull;
	float rox = w.rox, roy = w.roy, roz = w.roz, rdx = w.rdx, rdy = w.rdy, rdz = w.rdz, rix = w.rix, riy = w.riy, riz = w.riz, dist = w.dist;
	uint32_t ref = w.ref, slot = w.slot;
	int prim = w.prim;
	bool regular = (w.flags & 1u) != 0u, hitTri = (w.flags & 2u) != 0u, isShadow = (w.flags & 4u) != 0u, occluded = (w.flags & 8u) != 0u;
	const bool live = (w.flags & 16u) != 0u;
	uint32_t visible = 0;
	struct { int n; } st = { w.n };
	const uint32_t sub = lane & 3u, grp = lane >> 2;
	const unsigned long long lm = __ballot(live);
	const uint32_t nl = (uint32_t)__popcll(lm);
	// the k-th live lane tells lane k its number (ds_permute: a scatter through the LDS crossbar, no LDS memory -- the
	// block has none to spare: one more allocation granule and only four blocks fit a CU); idle lanes aim at lane 63,
	// which no group asks (nl <= 16)
	const uint32_t told = (uint32_t)__builtin_amdgcn_ds_permute(live ? (int)(__popcll(lm & below) << 2) : 63 * 4, (int)lane);
	bool gActive = grp < nl;
	const uint32_t asked = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(grp << 2), (int)told);
	const uint32_t srcLane = gActive ? asked : lane;
	const int pull = (int)(srcLane << 2);
	auto pull_f = [&](float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(pull, (int)__float_as_uint(v))); };
	auto pull_u = [&](uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute(pull, (int)v); };
	rox = pull_f(rox), roy = pull_f(roy), roz = pull_f(roz);
	rdx = pull_f(rdx), rdy = pull_f(rdy), rdz = pull_f(rdz);
	rix = pull_f(rix), riy = pull_f(riy), riz = pull_f(riz);
	dist = pull_f(dist);
	ref = pull_u(ref);
	slot = pull_u(slot);
	prim = (int)pull_u((uint32_t)prim);
	const uint32_t fl = pull_u((regular ? 1u : 0u) | (hitTri ? 2u : 0u) | (isShadow ? 4u : 0u) | (occluded ? 8u : 0u));
	regular = (fl & 1u) != 0u, hitTri = (fl & 2u) != 0u, isShadow = (fl & 4u) != 0u, occluded = (fl & 8u) != 0u;
	int n = (int)pull_u((uint32_t)st.n);
	// the ray's stack (at most STACK_LDS entries, all in its old lane's LDS column) moves into the group's four columns:
	// entry e at row e / 4 of lane e % 4
	typedef typename LdsStack<STACK_LDS, true>::entry_t entry_t;
	entry_t* const column0 = smem_ + (threadIdx.x >> 8) * (STACK_LDS * kBlock) + (threadIdx.x & 255u & ~63u); // row 0 of this wave's lane 0 (a block is one or more 256-thread parts, each with its own [STACK_LDS][256] stack array)
	entry_t moved[STACK_LDS / 4];
#pragma unroll
	for (int row = 0; row < STACK_LDS / 4; ++row) {
		const int e = 4 * row + (int)sub;
		moved[row] = (gActive && e < n) ? column0[e * kBlock + srcLane] : entry_t{};
	}
	wave_lds_order();
#pragma unroll
	for (int row = 0; row < STACK_LDS / 4; ++row)
		if (gActive && 4 * row + (int)sub < n)
			column0[row * kBlock + lane] = moved[row];
	wave_lds_order();
	entry_t* const gstack = column0 + (lane & ~3u); // entry e: gstack[(e >> 2) * kBlock + (e & 3)]
	const RayConst r = { mk3(rox, roy, roz), mk3(rdx, rdy, rdz), mk3(rix, riy, riz), rix < 0, riy < 0, riz < 0 };
	const uint32_t signBits = (r.nx ? 1u : 0u) | (r.ny ? 2u : 0u) | (r.nz ? 4u : 0u);
	bool wideOverflow = false;
	// The loop below is written with SELECTS, not branches.  A SIMD issues one scalar instruction per four cycles whichever of
	// its waves it comes from; the first form of this loop (lane-varying `if`s: pop / node / leaf / four accepts per leaf /
	// fast or generic box test) compiled to ~300 scalar instructions per step -- exec-mask save, branch, restore around
	// every block -- so that five waves per SIMD took 5 x 300 x 4 = 6000 cycles = 2.9 us per step, which is what the
	// per-wave anatomy measured (DESIGN.md section 4.4 "the drain, wave by wave"): the drain of every traversal launch was
	// bound by the scalar pipe, not by memory latency and not by the vector pipe.  Here every lane runs every block a wave
	// needs at all (wave-uniform branches on ballots), state changes are selects, loads of lanes that have nothing to load
	// go to record 0.  A lane that turns a node into a leaf tests the leaf in the same trip.
	const bool allRegular = __ballot(gActive && !regular) == 0ull; // (the generic box test is exact for regular rays too: one path for the wave)
	while (__ballot(gActive) != 0ull) {
		if (kGuardPasses && ++passes > kMaxPasses)
			break;
#ifdef TYR_LAUNCH_ANATOMY
		wideSteps += 1;
#endif
		// ---- pop: the group's top entry (entry 0 when it has none) ----
		{
			const bool popping = gActive && ref == kRefPop;
			const bool has = n > 0;
			const int top = has ? n - 1 : 0;
			const entry_t e = gstack[(top >> 2) * kBlock + (top & 3)];
			const uint32_t popped = !has ? kRefDone : (__uint_as_float(e.y) < dist ? e.x : kRefPop); // the pop-time half of Bbox.h:61
			ref = popping ? popped : ref;
			n = (popping && has) ? n - 1 : n;
		}
		// ---- one quad node: this lane's child box ----
		const bool atNode = gActive && (int)ref >= 0;
		if (__ballot(atNode) != 0ull) {
			const uint32_t idx = atNode ? (ref & kQuadIndexMask) : 0u, meta = ref >> kQuadOrderShift;
			const float* qf = reinterpret_cast<const float*>(quads + 8 * idx);
			const uint32_t at = (sub >> 1) * 4u + (sub & 1u) * 2u;
			const float2 bx = gload_f2(qf + at);
			const float2 by = gload_f2(qf + 8 + at);
			const float2 bz = gload_f2(qf + 16 + at);
			const uint32_t cref = gload_u(qf + 24 + sub);
			float t;
			bool h;
			if (allRegular)
				h = slab_fast(r, bx.x, bx.y, by.x, by.y, bz.x, bz.y, dist, t);
			else
				h = slab_test(r, r.nx ? bx.y : bx.x, r.nx ? bx.x : bx.y, r.ny ? by.y : by.x, r.ny ? by.x : by.y, r.nz ? bz.y : bz.x, r.nz ? bz.x : bz.y, dist, t);
			h = h && atNode;
			// this slot's place in the reference's visit order (test_quad: near slot first inside each group, near group first)
			const uint32_t aT = meta & 3u, aL = (meta >> 2) & 3u, aR = (meta >> 4) & 3u;
			const uint32_t bT = (signBits >> aT) & 1u, bG = (signBits >> ((sub >> 1) ? aR : aL)) & 1u;
			const uint32_t rank = 2u * ((sub >> 1) ^ bT) + ((sub & 1u) ^ bG);
			const uint32_t hr = quad_or(h ? (1u << rank) : 0u); // the group's hits, in visit order
			const uint32_t first = (uint32_t)__ffs((int)(hr | 16u)) - 1u; // (4 when nothing was hit: no lane's rank)
			// the others go onto the stack farthest first, so that the nearest pops first (push3's order)
			const int e = n + (int)__popc(hr >> (rank + 1u));
			if (h && rank != first && e < kWideStackEntries)
				gstack[(e >> 2) * kBlock + (e & 3)] = make_uint2(cref, __float_as_uint(t));
			int n2 = n + (int)__popc(hr) - (hr != 0u ? 1 : 0);
			wideOverflow = wideOverflow || (atNode && n2 > kWideStackEntries);
			n2 = n2 > kWideStackEntries ? kWideStackEntries : n2;
			const uint32_t nearest = quad_or((h && rank == first) ? cref : 0u);
			ref = atNode ? (hr == 0u ? kRefPop : nearest) : ref;
			n = atNode ? n2 : n;
		}
		// ---- a leaf: four primitives per round, accepted in array order (bvh.h:129-140 / 229-238) ----
		const bool atLeaf = gActive && ref_is_leaf(ref);
		if (__ballot(atLeaf) != 0ull) {
			const uint32_t off = ref & (kMaxPrimOffset - 1);
			const uint32_t cnt = atLeaf ? ((ref >> 26) & 31u) + 1u : 0u;
			bool found = false;
			for (uint32_t base = 0; __ballot(base < cnt) != 0ull; base += 4u) {
				const uint32_t i = base + sub;
				const bool mine = i < cnt;
				const TriData td = triangle_gload(tris, mine ? off + i : 0u);
				float tm = triangle_test_select(td, r);
				tm = mine ? tm : 0.0f;
				const float t0 = quad_bcast_f<0>(tm), t1 = quad_bcast_f<1>(tm), t2 = quad_bcast_f<2>(tm), t3 = quad_bcast_f<3>(tm);
				const float tk[4] = { t0, t1, t2, t3 };
#pragma unroll
				for (uint32_t k = 0; k < 4u; ++k) {
					const float t = tk[k];
					const bool in = (base + k < cnt) && t > kEpsilon && ((dist - t) > kEpsilon);
					found = found || (in && isShadow);                 // bvh.h:232-236
					const bool closer = in && !isShadow && t < dist;    // bvh.h:133-137
					prim = closer ? (int)(off + base + k) : prim;
					hitTri = hitTri || closer;
					dist = closer ? t : dist;
				}
			}
			occluded = occluded || found;
			ref = atLeaf ? (found ? kRefDone : kRefPop) : ref;
		}
		if (gActive && ref == kRefDone) {
			if (sub == 0u) {
				if (isShadow) {
					if (!occluded) { // kernel.cu:640-644
						const float4 c = shadowColor[slot];
						accumulate_pixel(blit, __float_as_int(shadowDyzCdIx[slot].w), mk3(c.x, c.y, c.z), 0);
						visible += 1;
					}
				} else {
					finish_extend_ray(workHit, slot, hitTri, dist, prim);
				}
			}
			gActive = false;
		}
	}
	return visible | (wideOverflow ? 0x80000000u : 0u) | (kGuardPasses && passes > kMaxPasses ? 0x40000000u : 0u);
}

// ======================================================================================
// k_trace_flat: extend(i + 1) and connect(i) in ONE persistent launch (tyr_render only).
//
// What the launch anatomy showed (tools/launch_tail.py on a -DTYR_LAUNCH_ANATOMY build, C3 at 1080p,
// profiles/r02_launch_anatomy_c3.txt): from the moment the first wave finds the queue used up, an extend launch needs
// another 320-390 us until its last wave has finished the rays it holds -- 40 % of the two fat launches of a frame, 52-63 %
// of the four thin ones.  That drain is as long as the launch's longest ray (a chain of dependent node fetches, ~150 quad
// steps at 1-2 us each), whatever the number of rays, and the machine idles through most of it; every traversal launch
// of a frame pays it once: six extends + six connects.  Splitting the frame into concurrent shards does not help (thinner
// launches, the same drains: tools/shard_probe.py).  What helps is fewer drains: connect(i) only depends on shade(i), and
// so does extend(i + 1) -- they can share a launch.  Queue items [0, nExt) are the work queue's rays (closest hit,
// kernel.cu:331-343), items [nExt, nExt + nShadow) the PREVIOUS iteration's shadow rays (any hit, kernel.cu:630-646); a
// lane takes whichever comes next and carries a flag.  The long closest-hit rays go first, the short any-hit rays fill
// the end of the launch.  Every ray is still answered on its own, into its own slot / pixel: bit-exact as before
// (shadow rays traverse in the reference's near-first order here; any-hit answers do not depend on the order).
// ======================================================================================
template <int STACK_LDS, uint32_t kTraceBlock>
// the arguments read where they lie (device_common.hpp kernarg_view): what the refill, the pixel flush and the kernel's end
// read of them is loaded there and not held in scalar registers through the descent (24 scalar spills -> 2)
#define TYR_TRACE_VIEW(name) const FrameParams& name = kernarg_view<FrameParams>();
__global__ void __launch_bounds__(kTraceBlock, (kTraceBlock == kTraceBlockWide ? 6 : TYR_FLAT_WAVES_PER_EU)) k_trace_flat(const FrameParams P) {
	constexpr bool COUNT = false; // (TYR_DBG)
	// the flat kernels' stack (TYR_DECLARE_FLAT_STACK), one [STACK_LDS][256] array per 256-thread part of the block
	__shared__ typename LdsStack<STACK_LDS, true>::entry_t smem_[STACK_LDS * kTraceBlock];
	uint32_t spillRef_[kStackSize - STACK_LDS];
	float spillT_[kStackSize - STACK_LDS];
	LdsStack<STACK_LDS, true> st;
	st.bind(smem_ + (threadIdx.x >> 8) * (STACK_LDS * kBlock) + (threadIdx.x & 255u), spillRef_, spillT_);
	st.reset();
	// LDS: 24,576 B of stack + 7,168 B of staged nodes + 4 = 31,748 B per block, and five blocks per CU are 158,740 of its
	// 163,840 B: ONE more allocation granule (a 256-byte array was enough) and the hardware places four while the occupancy
	// query still answers five -- the fifth of the persistent grid's blocks then run after the others (+30 % per render,
	// profiles/r02_wide_drain_ab.txt).  Nothing more fits here.
	__shared__ float4 stagedNodes[7 * kStagedNodes];
	__shared__ uint32_t blockNext;
	const uint32_t nStaged = P.scene.nStaged;
	for (uint32_t i = threadIdx.x; i < 7 * nStaged; i += kTraceBlock) {
		const uint32_t v = i / nStaged, n = i - v * nStaged;
		stagedNodes[v * kStagedNodes + n] = P.scene.quads[8 * n + v];
	}
	const uint32_t lane = lane_id();
	const unsigned long long below = (1ull << lane) - 1ull;
	[[maybe_unused]] uint32_t tripsFeed = 0; // (anatomy build) descent trips before the queue ran out
	// items = physical slots: the work queue's [0, nExt), then the shadow queue's; the few slots at the segments' ends that
	// hold no record are handed out like the others: the pre-passes (and k_primary) have made them rays that enter nothing
	const uint32_t nExt = P.traceShadow == 2u ? 0u : queue_extent(P.segWork);
	const uint32_t nItems = nExt + (P.traceShadow != 0u ? queue_extent(P.kcPrev->seg) : 0u); // (a render's first launch carries no shadow rays: kcPrev then holds an older render's count)
	const DevScene& sc = P.scene;
	float rox = 0.f, roy = 0.f, roz = 0.f, rdx = 0.f, rdy = 0.f, rdz = 0.f, rix = 0.f, riy = 0.f, riz = 0.f;
	bool regular = true, allRegular = true;
	float dist = 0.0f;          // closest hit so far / the shadow ray's closestDistance
	uint32_t ref = kRefDone, slot = 0; // slot: queue slot of an extend ray, shadow-queue index of a shadow ray
	int prim = 0;
	bool hitTri = false, live = false, overflow = false;
	bool isShadow = false, occluded = false;
	uint32_t visible = 0;
	uint32_t dbg[16] = {};
	uint32_t steps = 0; // TYR_QUAD_STATS: quad steps of this lane's current ray
	[[maybe_unused]] unsigned long long tExhausted = 0ull, tWide = 0ull;
	[[maybe_unused]] uint32_t liveAtExhaustion = 0, liveAtWide = 0, tripsAfter = 0, passesAfter = 0, wideSteps = 0; // (anatomy build)
	const unsigned long long tStart = kAnatomy ? __builtin_amdgcn_s_memrealtime() : 0ull;
	// kernel.cu:640-644, deferred to the wave's next refill (see k_connect_count)
	constexpr uint32_t kNoPending = 0xffffffffu;
	uint32_t pendIdx = kNoPending;
	auto flush_visible = [&]() {
		TYR_TRACE_VIEW(P)
		float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
		int px = 0;
		if (pendIdx != kNoPending) {
			c = P.shadowPrev.color[pendIdx];
			px = __float_as_int(P.shadowPrev.dyz_cd_ix[pendIdx].w);
		}
		accumulate_pixels_wave(P.blit, px, mk3(c.x, c.y, c.z), 0);
		pendIdx = kNoPending;
	};
	const uint32_t perBlock = static_range(nItems, P.staticShare);
	const uint32_t dynBase = perBlock * gridDim.x;
	const uint32_t blockBegin = blockIdx.x * perBlock, blockEnd = blockBegin + perBlock;
	ChunkFeed feed;
	feed.init(nItems - dynBase, P.ticketChunk);
	bool staticDone = (perBlock == 0);
	if (threadIdx.x == 0)
		blockNext = blockBegin;
	__syncthreads();
	if (P.scanPrevInTrace == 1u) {
		// the slot scan of the iteration before (scan_wave.hpp): a wave per 16384-slot block, dealt over the BLOCKS first (block b's wave
		// k takes scan block k * gridDim + b: no block has more than one or two waves at it), behind the barrier -- the block's other
		// waves are already drawing rays from its share, which waits for nobody
		const uint32_t wavesPerBlock = kTraceBlock / 64u;
		scan_blocks_by_wave(P.survFlag, const_cast<unsigned long long*>(P.vPrev.word), const_cast<uint32_t*>(P.vPrev.pre), const_cast<uint32_t*>(P.vPrev.blk), &P.k->scan_blocks_done,
		                    *P.scanLivePrev, (threadIdx.x >> 6) * gridDim.x + blockIdx.x, gridDim.x * wavesPerBlock, P.retireGhosts != 0u);
	}
	bool exhausted = nItems == 0;
	bool wide = false;
	const uint32_t wideLimit = (P.wideDrain != 0u && P.scene.quadMaxStack <= (uint32_t)kWideStackEntries) ? kWideRays : 0u; // (a tree that could need more than the group's 48 entries keeps its rays one to a lane)
	uint32_t passes = 0;

	for (;;) {
		if (kGuardPasses && ++passes > kMaxPasses)
			break;
		// ---- refill free lanes from the two queues ----
		const unsigned long long idleMask = __ballot(!live);
		const uint32_t nIdle = __popcll(idleMask);
		if (!exhausted && nIdle >= P.refillMinIdle) {
			TYR_TRACE_VIEW(P) // the refill's reads of the arguments (queue pointers, tickets, the root box): loaded here, not held through the descent
			const DevScene& sc = P.scene;
			const uint32_t rank = __popcll(idleMask & below);
			uint32_t s = 0;
			bool fed = false;
			uint32_t got = 0;
			if (!staticDone) {
				uint32_t base = 0;
				if (lane == 0)
					base = atomicAdd(&blockNext, nIdle); // LDS
				base = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl(base, 0, 64));
				const uint32_t avail = base < blockEnd ? blockEnd - base : 0u;
				got = avail < nIdle ? avail : nIdle;
				if (!live && rank < got) {
					s = static_slot(base + rank, blockBegin, P.staticInterleave != 0u);
					fed = true;
				}
				staticDone = (base + nIdle >= blockEnd);
			}
			while (staticDone && got < nIdle) {
				if (!feed.refill(P.k->extend_chunks, nItems - dynBase, lane)) {
					exhausted = true;
					if (kAnatomy && tExhausted == 0ull) {
						tExhausted = __builtin_amdgcn_s_memrealtime();
						liveAtExhaustion = (uint32_t)__popcll(__ballot(live)) + got;
					}
					break;
				}
				const uint32_t avail = feed.end - feed.next, room = nIdle - got;
				const uint32_t take = avail < room ? avail : room;
				if (!live && rank >= got && rank < got + take) {
					s = dynBase + feed.next + (rank - got);
					fed = true;
				}
				feed.next += take;
				got += take;
			}
			if (__ballot(pendIdx != kNoPending) != 0ull)
				flush_visible();
			if (fed) {
				TYR_DBG(6)
				float4 a;
				float by, bz, bound;
				bool blocked = false;
				if (s < nExt) { // a ray of the work queue: closest hit
					a = P.work.o_dx[s];
					const float2 b = P.work.dyz[s];
					const float2 h = P.work.hit[s];
					by = b.x, bz = b.y, bound = h.x; // the sphere pre-pass's distance bounds the search
					slot = s;
					isShadow = false;
				} else { // a shadow ray of the previous iteration: any hit within closestDistance
					const uint32_t idx = s - nExt;
					a = P.shadowPrev.o_dx[idx];
					const float4 b = P.shadowPrev.dyz_cd_ix[idx];
					blocked = reinterpret_cast<const float*>(&P.shadowPrev.color[idx])[3] != 0.0f; // the sphere pre-pass's verdict
					by = b.x, bz = b.y, bound = b.z;
					slot = idx;
					isShadow = true;
				}
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
				// an extend ray that misses the root box is finished here (the pre-pass's answer stands, nothing to write); a
				// shadow ray a sphere blocks likewise; a shadow ray that misses the tree is visible: it retires below
				live = (ref != kRefDone) || (isShadow && !blocked);
			}
			if (!exhausted && (uint32_t)__popcll(__ballot(live && ref != kRefDone)) < P.minTraversing)
				if (__ballot(live && ref == kRefDone) == 0ull)
					continue; // mostly short rays (primary rays that end at the root box): top the wave up again first
		}
		{
			// one way out of the loop: the queue is used up and at most wideLimit rays are left (0: none) -- those are
			// finished four lanes to a ray below
			const uint32_t nLiveNow = (uint32_t)__popcll(__ballot(live));
			if (exhausted && nLiveNow <= wideLimit && __ballot(live && st.n > STACK_LDS) == 0ull) {
				wide = nLiveNow != 0u;
				if (kAnatomy) {
					tWide = __builtin_amdgcn_s_memrealtime();
					liveAtWide = nLiveNow;
				}
				break;
			}
			if (kAnatomy && exhausted)
				passesAfter += 1;
			if (nLiveNow == 0u)
				continue;
		}
		allRegular = (__ballot(live && !regular) == 0ull);
		const RayConst r = { mk3(rox, roy, roz), mk3(rdx, rdy, rdz), mk3(rix, riy, riz), rix < 0, riy < 0, riz < 0 }; // bvh.h:120-121
		// ---- descent: one pop attempt + one quad test per lane per trip (the same for both kinds of ray) ----
		for (;;) {
			const uint32_t nTrav = __popcll(lanes_traversing(ref));
			if (nTrav == 0)
				break;
			if (nTrav < P.minTraversing) {
				const bool anyLeaf = lanes_at_leaf(ref) != 0ull;
				const bool canRefill = !exhausted && (uint32_t)__popcll(__ballot(!live || ref == kRefDone)) >= P.refillMinIdle;
				if (anyLeaf || canRefill)
					break;
			}
			if (kAnatomy && exhausted)
				tripsAfter += 1;
			if (kAnatomy && !exhausted)
				tripsFeed += 1;
			if (ref == kRefPop) {
				TYR_DBG(2)
				uint32_t pr;
				float pt;
				if (st.pop(pr, pt)) {
					if (pt < dist) // the pop-time half of Bbox.h:61 (always true for a shadow ray: its bound never shrinks)
						ref = pr;
				} else {
					ref = kRefDone;
				}
			}
			if ((int)ref >= 0) {
				TYR_DBG(0)
				if (kLoopStats)
					steps += 1;
				const QuadHits q = allRegular ? test_quad<true, true, true>(sc.quads, ref, r, dist, stagedNodes, nStaged) : test_quad<false, true, true>(sc.quads, ref, r, dist, stagedNodes, nStaged);
				const lanemask any01 = q.hit[0] | q.hit[1], any012 = any01 | q.hit[2];
				st.push3(q.hit[3] & any012, q.ref[3], q.t[3], q.hit[2] & any01, q.ref[2], q.t[2], q.hit[1] & q.hit[0], q.ref[1], q.t[1]);
				ref = lane_in(q.hit[0]) ? q.ref[0] : lane_in(q.hit[1]) ? q.ref[1] : lane_in(q.hit[2]) ? q.ref[2] : lane_in(q.hit[3]) ? q.ref[3] : kRefPop;
			}
		}
		// ---- leaves: bvh.h:129-140 (closest hit) / bvh.h:229-238 (any hit) ----
		// ---- leaves, packed (round 5; -1.0 ... -1.6 % per C3 render, profiles/r05_packed_leaves_ab.txt).  The loop below runs max(cnt) rounds over the lanes that are at a leaf -- typically 20 / 12 / 7 / 4
		// lanes in rounds 0..3: 17 % of the lanes on average, as many rounds as the kernel has node trips.  Here every (lane, primitive)
		// pair of the phase becomes one ITEM, the items are dealt to the wave's lanes round-major (all first primitives, then all
		// second ones, ...), each lane tests the item it was dealt with its owner's ray, and the owners take the distances back and
		// fold them in array order as before (bvh.h:129-140 / 229-238): one round of tests for the whole phase.  Owner -> item lane is a
		// ds_permute (the owner knows its item's slot: the round's base + its rank among the round's lanes), item lane -> owner
		// registers are ds_bpermutes: the LDS crossbar, no LDS memory.  Phases with a leaf of more than four primitives or more than 63
		// items take the loop. ----
		{
			const bool atLeaf = ref_is_leaf(ref);
			const unsigned long long B0 = __ballot(atLeaf);
			if (B0 != 0ull) {
				const uint32_t off = ref & (kMaxPrimOffset - 1);
				const uint32_t cnt = atLeaf ? ((ref >> 26) & 31u) + 1u : 0u;
				const unsigned long long B1 = __ballot(cnt > 1u), B2 = __ballot(cnt > 2u), B3 = __ballot(cnt > 3u);
				const uint32_t n0 = (uint32_t)__popcll(B0), n1 = (uint32_t)__popcll(B1), n2 = (uint32_t)__popcll(B2), n3 = (uint32_t)__popcll(B3);
				if (__ballot(cnt > 4u) == 0ull && n0 + n1 + n2 + n3 <= 63u) {
					const uint32_t base1 = n0, base2 = n0 + n1, base3 = n0 + n1 + n2;
					const uint32_t s0 = (uint32_t)__popcll(B0 & below), s1 = base1 + (uint32_t)__popcll(B1 & below), s2 = base2 + (uint32_t)__popcll(B2 & below), s3 = base3 + (uint32_t)__popcll(B3 & below);
					// an owner tells the lane of each of its items who it is and which primitive it wants (lanes that own nothing in a
					// round send 0 to lane 63, which holds no item)
					const uint32_t me = lane + 1u;
					uint32_t told = (uint32_t)__builtin_amdgcn_ds_permute((int)((cnt > 0u ? s0 : 63u) << 2), (int)(cnt > 0u ? me : 0u));
					if (n1)
						told |= (uint32_t)__builtin_amdgcn_ds_permute((int)((cnt > 1u ? s1 : 63u) << 2), (int)(cnt > 1u ? (me | (1u << 8)) : 0u));
					if (n2)
						told |= (uint32_t)__builtin_amdgcn_ds_permute((int)((cnt > 2u ? s2 : 63u) << 2), (int)(cnt > 2u ? (me | (2u << 8)) : 0u));
					if (n3)
						told |= (uint32_t)__builtin_amdgcn_ds_permute((int)((cnt > 3u ? s3 : 63u) << 2), (int)(cnt > 3u ? (me | (3u << 8)) : 0u));
					const bool item = lane < n0 + n1 + n2 + n3; // (every slot below the item count was told by exactly one owner)
					const int from = (int)((item ? (told & 0xffu) - 1u : lane) << 2);
					auto pull = [&](float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)__float_as_uint(v))); };
					const uint32_t offO = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)off);
					RayConst ro;
					ro.o = mk3(pull(rox), pull(roy), pull(roz));
					ro.d = mk3(pull(rdx), pull(rdy), pull(rdz));
					// (the test without branches: most lanes hold an item, and three early-outs are three exec-mask sequences; lanes
					// without an item test record 0 and drop the answer)
					if ((COUNT || kLoopStats) && item) {
						TYR_DBG(4)
					}
					float tm = triangle_test_select(triangle_load(sc.tris, item ? offO + (told >> 8) : 0u), ro);
					tm = item ? tm : 0.0f;
					// the owners take their distances back and accept in array order
					const float t0 = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute((int)(s0 << 2), (int)__float_as_uint(tm)));
					const float t1 = n1 ? __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute((int)(s1 << 2), (int)__float_as_uint(tm))) : 0.0f;
					const float t2 = n2 ? __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute((int)(s2 << 2), (int)__float_as_uint(tm))) : 0.0f;
					const float t3 = n3 ? __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute((int)(s3 << 2), (int)__float_as_uint(tm))) : 0.0f;
					const float tk[4] = { t0, t1, t2, t3 };
					bool found = false;
#pragma unroll
					for (uint32_t i = 0; i < 4u; ++i) {
						const float t = tk[i];
						const bool in = i < cnt && !found && t > kEpsilon && ((dist - t) > kEpsilon);
						const bool closer = in && !isShadow && t < dist; // bvh.h:133-137
						found = found || (in && isShadow);                 // bvh.h:232-236
						prim = closer ? (int)(off + i) : prim;
						hitTri = hitTri || closer;
						dist = closer ? t : dist;
					}
					if (atLeaf) {
						occluded = occluded || found;
						ref = found ? kRefDone : kRefPop;
					}
				}
			}
		}
		if (ref_is_leaf(ref)) {
			const uint32_t off = ref & (kMaxPrimOffset - 1);
			const uint32_t cnt = ((ref >> 26) & 31u) + 1u;
			bool found = false;
			TriData tri = triangle_load(sc.tris, off);
			for (uint32_t i = 0; i < cnt && !found; ++i) {
				TYR_DBG(4)
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
		if (live && ref == kRefDone) {
			if (isShadow) {
				if (!occluded) { // kernel.cu:640-644, added to the pixel at the wave's next refill
					pendIdx = slot;
					visible += 1;
				}
			} else {
				finish_extend_ray(P.work.hit, slot, hitTri, dist, prim);
			}
			overflow = overflow || st.overflow;
			live = false;
			if (kLoopStats) {
				// the launch's longest rays: debug[12] = most quad steps of one ray, [9] / [10] / [11] = rays with > 64 / 128 / 256
				atomicMax(&P.k->debug[12], (unsigned long long)steps);
				if (steps > 64)
					atomicAdd(&P.k->debug[9], 1ull);
				if (steps > 128)
					atomicAdd(&P.k->debug[10], 1ull);
				if (steps > 256)
					atomicAdd(&P.k->debug[11], 1ull);
				steps = 0;
			}
		}
	}
	flush_visible();
	TYR_TRACE_VIEW(PE) // what the kernel's end reads (not held through the loop)
	if (wide) {
		WideState w;
		w.rox = rox, w.roy = roy, w.roz = roz, w.rdx = rdx, w.rdy = rdy, w.rdz = rdz, w.rix = rix, w.riy = riy, w.riz = riz, w.dist = dist;
		w.ref = ref, w.slot = slot, w.prim = prim, w.n = st.n;
		w.flags = (regular ? 1u : 0u) | (hitTri ? 2u : 0u) | (isShadow ? 4u : 0u) | (occluded ? 8u : 0u) | (live ? 16u : 0u);
		const uint32_t res = wide_drain<STACK_LDS>(PE.scene.quads, PE.scene.tris, PE.shadowPrev.color, PE.shadowPrev.dyz_cd_ix, PE.work.hit, PE.blit, smem_, w, passes TYR_WIDE_STEPS_ARG);
		visible += res & 0x3fffffffu;
		overflow = overflow || (res & 0x80000000u) != 0u;
		if (res & 0x40000000u)
			passes = kMaxPasses + 1;
	}
	if (overflow)
		atomicOr(&PE.k->device_error, kErrStackOverflow);
	if (kGuardPasses && passes > kMaxPasses)
		atomicOr(&PE.k->device_error, kErrNoProgress);
	wave_add_u64(&PE.k->n_shadow_visible, visible);
	if (kLoopStats) {
		for (int i = 0; i < 9; ++i)
			wave_add_u64(&PE.k->debug[i], dbg[i]);
	}
	if (kAnatomy && lane == 0) {
		const unsigned long long tEnd = __builtin_amdgcn_s_memrealtime();
		atomicMax(&P.k->debug[13], ~tStart);
		atomicMax(&P.k->debug[14], ~(tExhausted ? tExhausted : tEnd));
		atomicMax(&P.k->debug[15], tEnd);
		// per-wave records for host/driver.cpp's TYR_ANATOMY=2 printout, parked in an array nobody uses during this launch (the NEXT queue's
		// hit column): microseconds from this wave's start to "queue used up" and to its exit, and how many of its lanes
		// still held a ray when the queue ran out
		const uint32_t w = blockIdx.x * (kTraceBlock / 64) + (threadIdx.x >> 6);
		if (w < P.N)
			P.next.hit[w] = make_float2((float)((tExhausted ? tExhausted : tEnd) - tStart) * 0.01f, (float)(tEnd - tStart) * 0.01f + (float)liveAtExhaustion * 0.0f);
		if (w < 8192u && 32768u < P.N) { // three more records per wave, further up the same column (host/driver.cpp prints them with TYR_ANATOMY=2)
			P.next.hit[8192u + w] = make_float2(tWide ? (float)(tWide - tStart) * 0.01f : 0.0f, (float)(liveAtExhaustion + 256u * liveAtWide));
			P.next.hit[16384u + w] = make_float2((float)tripsAfter, (float)wideSteps); // trips after the queue ran out one ray to a lane, steps four lanes to a ray
			P.next.hit[24576u + w] = make_float2((float)passesAfter, (float)tripsFeed);
		}
	}
}

#undef TYR_DBG
#undef TYR_TRACE_VIEW


// extend of this iteration + connect of the previous one in one launch.  P.kc must be this iteration's set, P.kcPrev the
// set of the iteration whose shadow rays ride along (P.traceShadow: 0 none, 1 beside the extend rays, 2 they are all of it).
#define TYR_TRACE_STACK 12 // LDS stack entries per lane of k_trace_flat
void launch_trace(const FrameParams& P, uint32_t maxLive, uint32_t nSurvivors, uint32_t maxShadowPrev, const Tuning& t, int numCUs, LaunchCache& lc, hipStream_t stream) {
	launch_trace_prepasses(P, nSurvivors, maxShadowPrev, stream, maxLive); // (the shadow rays' pre-pass reads its counts in kcPrev)
	launch_trace_kernel(P, maxLive + maxShadowPrev, t, numCUs, lc, stream);
}
void launch_trace_kernel(const FrameParams& P, uint32_t items, const Tuning& t, int numCUs, LaunchCache& lc, hipStream_t stream) {
	if (t.wideBlockMinItems >= 0 && items >= (uint32_t)t.wideBlockMinItems)
		hipLaunchKernelGGL((k_trace_flat<TYR_TRACE_STACK, kTraceBlockWide>), dim3(persistent_blocks(k_trace_flat<TYR_TRACE_STACK, kTraceBlockWide>, items, t, numCUs, lc.perCU[kLcTrace][1], kTraceBlockWide)), dim3(kTraceBlockWide), 0, stream, P);
	else
		hipLaunchKernelGGL((k_trace_flat<TYR_TRACE_STACK, (uint32_t)kBlock>), dim3(persistent_blocks(k_trace_flat<TYR_TRACE_STACK, (uint32_t)kBlock>, items, t, numCUs, lc.perCU[kLcTrace][0])), dim3(kBlock), 0, stream, P);
}
void launch_trace_prepasses(const FrameParams& P, uint32_t nSurvivors, uint32_t maxShadowPrev, hipStream_t stream, uint32_t maxLive) {
	FrameParams Pc = P;
	Pc.kc = P.kcPrev;
	Pc.shadow = P.shadowPrev;
	if (P.prevFolded) {
		// the sphere halves were done by the shade launch that made these rays (P.foldSpheres there): only the holes at the
		// segments' ends are left, of the work queue's class 0 and of the shadow queue in one launch -- unless the k_scan_words
		// launch behind that shade launch has done them too (P.prologueDone)
		if (!P.prologueDone)
			launch_pad_holes(Pc, P.traceShadow != 2u, maxShadowPrev != 0, stream, P.traceShadow == 2u);
		return;
	}
	if (P.traceShadow != 2u)
		launch_extend_spheres(P, nSurvivors, stream, maxLive);
	if (maxShadowPrev != 0)
		launch_connect_spheres(Pc, maxShadowPrev, stream);
	if (P.traceShadow != 2u && nSurvivors == 0)
		launch_pad_holes(Pc, true, false, stream); // no sphere pre-pass this iteration: the holes at the ends of the work queue's segments
}

// extend / connect as launches of their own (the stage API, tyr_launch_kernels): the same kernel with one kind of ray;
// the counting build (TYR_FLAG_COUNT_VISITS) traverses pair nodes, the only layout that reproduces the reference's
// visit counts (bvh.h:164-209), one block per kCountRaysPerBlock slots of the queue's capacity.
void launch_extend(const FrameParams& P0, uint32_t maxLive, uint32_t nSurvivors, bool countVisits, const Tuning& t, int numCUs, LaunchCache& lc, hipStream_t stream) {
	if (maxLive == 0)
		return;
	FrameParams P = P0;
	if (countVisits) {
		launch_extend_spheres(P, nSurvivors, stream, maxLive);
		P.raysPerBlock = kCountRaysPerBlock;
		const uint32_t blocks = (P.segCap * kSegs + kCountRaysPerBlock - 1) / kCountRaysPerBlock; // every physical slot a record could lie in
		hipLaunchKernelGGL((k_extend_count<12>), dim3(blocks), dim3(kBlock), 0, stream, P);
		return;
	}
	P.traceShadow = 0u;
	launch_trace(P, maxLive, nSurvivors, 0u, t, numCUs, lc, stream);
}
void launch_connect(const FrameParams& P0, uint32_t maxShadow, bool countVisits, const Tuning& t, int numCUs, LaunchCache& lc, hipStream_t stream) {
	if (maxShadow == 0)
		return;
	FrameParams P = P0;
	if (countVisits) {
		launch_connect_spheres(P, maxShadow, stream);
		P.raysPerBlock = kCountRaysPerBlock;
		const uint32_t blocks = (P.segCap * kSegs + kCountRaysPerBlock - 1) / kCountRaysPerBlock;
		hipLaunchKernelGGL((k_connect_count<12>), dim3(blocks), dim3(kBlock), 0, stream, P);
		return;
	}
	P.kcPrev = P.kc; // the rays of THIS iteration's shadow queue
	P.shadowPrev = P.shadow;
	P.traceShadow = 2u;
	launch_trace(P, 0u, 0u, maxShadow, t, numCUs, lc, stream);
}

} // namespace tyr
