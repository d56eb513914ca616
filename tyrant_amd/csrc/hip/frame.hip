// frame.hip -- the per-frame kernels around the traversal: primary_rays (kernel.cu:247-297),
// set_wavefront_globals (kernel.cu:227-244), the sphere halves of intersect_scene / intersect_scene_simple
// (kernel.cu:125-136, 168-172) as coherent pre-passes, blit_onto_framebuffer (kernel.cu:648-662).
#include "device_common.hpp"
#include "prologue.hpp"

namespace tyr {

// ======================================================================================
// primary_rays, kernel.cu:247-297.  One thread per new queue slot.
// ======================================================================================
__global__ void __launch_bounds__(kBlock) k_primary(const FrameParams P) {
	__shared__ uint32_t baseSh[kClasses], cntSh[8], lastSh;
	const uint32_t index = blockIdx.x * kBlock + threadIdx.x;
	const uint32_t cnt = P.k->primary_ray_cnt; // survivors already in the buffer (kernel.cu:253)
	const unsigned long long room = (unsigned long long)(P.N - cnt);
	const unsigned long long budget = P.k->budget_remaining;
	const uint32_t nNew = (uint32_t)(room < budget ? room : budget);
	const bool mine = index < nNew; // (a launch with nothing to generate is one block that only runs the globals below)
	const uint32_t vslot = index + cnt; // the slot the serial order gives this ray (kernel.cu:254): what seeds its shading
	// kernel.cu:258 seeds by the ticket `index`; with pixel sharding (nranks > 1) the ranks' tickets are interleaved so that
	// rows y = yl * R + r, r = 0..R-1, do not share their jitter and lens samples (nranks == 1: the reference's expression)
	uint32_t seed = (P.frame * 147565741u) * 720898027u * (index * P.nranks + P.rank);

	const uint32_t start = P.k->start_position;
	const int x = (int)((start + index) % P.W);
	const int yl = (int)(((start + index) / P.W) % P.localRows);
	const int y = yl * (int)P.nranks + (int)P.rank; // nranks == 1: kernel.cu:264

	float sx, sy;
	stratified_sample(seed, sx, sy);
	const float jitteredX = (float)x - sx; // kernel.cu:268-269 (jitter is subtracted)
	const float jitteredY = (float)y - sy;
	const float ndcX = (jitteredX / (float)P.W) - 0.5f;
	const float ndcY = (((float)P.H - jitteredY) / (float)P.H) - 0.5f;

	const f3 O = ld3(P.camPos), camFwd = ld3(P.camDir), camRgt = ld3(P.camRight), camUpv = ld3(P.camUp);
	f3 towardFocus = camFwd + ndcX * camRgt + ndcY * camUpv;
	towardFocus = normalize(towardFocus);
	const int kFocalScale = 3; // kernel.cu:286 (`ImGui_slider_hack`: the focal distance is always tripled)
	const f3 focusPoint = O + (P.focalDistance * (float)kFocalScale) * towardFocus;

	const float l0 = rng_float(seed);
	const float l1 = rng_float(seed);
	float dx, dy;
	concentric_sample_disk(l0, l1, dx, dy);
	const float pLx = P.lensRadius * dx, pLy = P.lensRadius * dy;
	const f3 lensPoint = O + camRgt * pLx + camUpv * pLy;
	const f3 direction = normalize(focusPoint - lensPoint);

	// extend's sphere half for this ray, while it is in registers (k_extend_spheres then only has the survivors of the
	// last iteration to do), and the traversal's own first test (root_ref at the refill of k_trace_flat: same function,
	// same bound, same answer): a ray that fails it goes to class 1 and is finished with this record
	const float2 hitRecord = sphere_hit_record(P, lensPoint, direction);
	const bool tree = mine && P.scene.rootRef != kRefDone && root_ref(P.scene, make_ray(lensPoint, direction), hitRecord.x) != kRefDone;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	// A camera ray that hits no sphere and fails the root box is finished before it is queued (P.retireSky, tyr_render's
	// merged path): what shade would do with it is fixed -- kernel.cu:613-617 with the fresh ray's lastSpecular = true and
	// direct = (1,1,1) (kernel.cu:295, variables.h:33): color = sunsky(direction), one finished path, no survivor, no shadow
	// ray, and NO random number drawn -- so it is done here, while the direction is in registers: the pixel gets its
	// radiance (the same operations: 0 + 1 * x is x), the ray's survive byte is 0, and the record is never written or read
	// (42 % of the first wavefront's rays on C3: 112 bytes of queue traffic and a shade lane each).  Every count stays: the
	// ray was generated and traced (n_live and the totals are sums, not queue lengths).
	const bool sky = P.retireSky != 0u && mine && !tree && !(hitRecord.x < kVeryFar);
	if (__ballot(sky) != 0ull) {
		f3 radiance = mk3(0.f, 0.f, 0.f);
		if (sky) {
			if (P.sun.sunAngularDiameterCos == 1.0f) {
				radiance = mk3(1.0f, 0.0f, 0.0f); // sunsky.cu:118-119
			} else {
				const Atmosphere a = atmosphere(P.sun, direction);
				radiance = sunsky_radiance(P.sun, a);
			}
			P.survFlag[vslot] = 0; // what k_shade writes for a ray that does not survive (k_scan_words reads every slot below n_live)
		}
		accumulate_pixels_wave(P.blit, y * (int)P.W + x, radiance, sky ? 1 : 0);
	}
	// the block's rays of either class go to segment blockIdx % 8 of that class, behind what it holds: one atomic each
	const unsigned long long bt = __ballot(tree), bsky = __ballot(mine && !tree && !sky), below = (1ull << lane) - 1ull;
	if (lane == 0) {
		cntSh[wave] = (uint32_t)__popcll(bt);
		cntSh[4 + wave] = (uint32_t)__popcll(bsky);
	}
	__syncthreads();
	uint32_t before[2] = { 0, 0 }, total[2] = { 0, 0 };
#pragma unroll
	for (uint32_t w = 0; w < kBlock / 64; ++w) {
		if (w < wave) {
			before[0] += cntSh[w];
			before[1] += cntSh[4 + w];
		}
		total[0] += cntSh[w];
		total[1] += cntSh[4 + w];
	}
	const uint32_t seg = blockIdx.x & (kSegs - 1u);
	if (threadIdx.x < kClasses) {
		const uint32_t c = threadIdx.x, n = total[c];
		uint32_t base = 0;
		if (n) {
			base = atomicAdd(&P.segWork[c * kClassWords + seg * kSegStride], n);
			if (base + n > P.segCap) {
				atomicOr(&P.k->device_error, kErrQueueOverflow);
				base = 0xffffffffu;
			}
		}
		baseSh[c] = base;
	}
	__syncthreads();
	const uint32_t cls = tree ? 0u : 1u;
	if (mine && !sky && baseSh[cls] != 0xffffffffu) {
		const uint32_t rank = before[cls] + (uint32_t)__popcll((tree ? bt : bsky) & below);
		const uint32_t slot = cls * P.classStride + seg_phys(seg, baseSh[cls] + rank);
		// kernel.cu:295: {origin, direction, {1,1,1}, 0, 0, 0, pixel}; lastSpecular defaults to true (variables.h:33)
		P.work.o_dx[slot] = make_float4(lensPoint.x, lensPoint.y, lensPoint.z, direction.x);
		P.work.dyz[slot] = make_float2(direction.y, direction.z);
		P.work.direct_ix[slot] = make_float4(1.0f, 1.0f, 1.0f, __int_as_float(y * (int)P.W + x));
		P.work.flags[slot] = 0u | (1u << 8);
		P.work.hit[slot] = hitRecord;
		P.work.key[slot] = vslot;
	}
	// set_wavefront_globals (kernel.cu:227-244, launched behind primary_rays at kernel.cu:720): by the block that finishes last
	__syncthreads();
	if (threadIdx.x == 0) {
		// eight counters (one word serves only ~88 returning atomics per microsecond; a 16.6 M-ray top-up is 65 k blocks):
		// the block that completes its word counts the word, the block that completes the eighth word is the last.
		// No fence: the last block reads nothing the others wrote, it only overwrites counts they have finished reading
		// (their own stores depended on those reads); a release here is an L2 write-back per block -- 2.4 ms of a 16.6 M-ray
		// top-up when it was tried.
		const uint32_t w = blockIdx.x & (kTicketWords - 1u);
		const uint32_t mineOfWord = (gridDim.x - w + kTicketWords - 1u) / kTicketWords; // blocks b with b % 8 == w
		uint32_t last = 0;
		if (atomicAdd(&P.k->primary_done[w * 32], 1u) + 1u == mineOfWord) {
			const uint32_t words = gridDim.x < kTicketWords ? gridDim.x : kTicketWords;
			last = atomicAdd(&P.k->primary_blocks_done, 1u) + 1u == words ? 1u : 0u;
		}
		lastSh = last;
	}
	__syncthreads();
	if (lastSh)
		wavefront_globals(P);
}

// extend pre-pass: kernel.cu:125-136 (spheres first; their distance bounds the BVH search)
// (grid-stride over the device's count: the host may have sized the grid from an upper bound, and a capped grid makes a
// loose bound free)
__global__ void __launch_bounds__(kBlock) k_extend_spheres(const FrameParams P) {
	// the counts once per block through LDS: 32 loads of eight hot cache lines per THREAD were a fixed ~25 us of this kernel
	__shared__ uint32_t cntSh[2 * kClasses * kSegs];
	if (threadIdx.x < kClasses * kSegs) {
		const uint32_t c = threadIdx.x / kSegs, w = threadIdx.x % kSegs;
		cntSh[threadIdx.x] = P.k->segSurv[c][w];
		cntSh[kClasses * kSegs + threadIdx.x] = P.segWork[c * kClassWords + w * kSegStride];
	}
	__syncthreads();
	const uint32_t first = blockIdx.x * kBlock + threadIdx.x, stride = gridDim.x * kBlock;
	// the survivors of the last iteration, both classes: the records in front of what a top-up appended (k_primary has done
	// its own rays).  Class 0: the distance bounds the BVH search; class 1: the record is the ray's answer.
	// (validity straight from the LDS table, one read per test: a private array of the eight counts is "promoted" to LDS
	// by the compiler anyway -- 64 bytes per thread -- and then read eight times per test)
	auto extent_of = [&](uint32_t at) {
		uint32_t m = 0;
#pragma unroll
		for (uint32_t w = 0; w < kSegs; ++w)
			m = cntSh[at + w] > m ? cntSh[at + w] : m;
		return ((m + 63u) >> 6) * (kSegs * 64u);
	};
	auto holds = [&](uint32_t at, uint32_t j) { return (((j >> 9) << 6) | (j & 63u)) < cntSh[at + ((j >> 6) & (kSegs - 1u))]; };
	for (uint32_t c = 0; c < kClasses; ++c) {
		const uint32_t surv = c * kSegs, now = (kClasses + c) * kSegs;
		// class 0 is walked up to its extent: the slots at the segments' ends that hold no record become rays that enter
		// nothing (k_trace_flat hands out slots [0, extent) without asking)
		const uint32_t n = c == 0 ? extent_of(now) : extent_of(surv), base = c * P.classStride;
		for (uint32_t j = first; j < n; j += stride) {
			const uint32_t slot = base + j;
			if (!holds(surv, j)) {
				if (c == 0 && !holds(now, j))
					write_dead_ray(P.work, slot);
				continue;
			}
			const float4 a = P.work.o_dx[slot];
			const float2 b = P.work.dyz[slot];
			P.work.hit[slot] = sphere_hit_record(P, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y));
		}
	}
}

// The same for an iteration whose sphere pre-pass does not run (no survivors: a render's first wavefront): the slots at
// the ends of the eight segments that hold no record become rays that enter nothing.  Block w does segment w; what == 0:
// class 0 of the work queue, 1: the shadow queue (unused: its pre-pass always runs).
__global__ void __launch_bounds__(kBlock) k_pad_holes(const FrameParams P, uint32_t workQueue, uint32_t shadowQueue, uint32_t resetTickets) {
	const uint32_t w = blockIdx.x & (kSegs - 1u), what = blockIdx.x / kSegs;
	if (resetTickets && blockIdx.x == 0 && threadIdx.x < kTicketWords)
		P.k->extend_chunks[threadIdx.x * 32] = 0; // k_trace_flat's tickets, when no connect pre-pass opens the launch that ends a render (P.foldSpheres)
	if ((what == 0 && !workQueue) || (what == 1 && !shadowQueue))
		return;
	const uint32_t* cnt = what == 0 ? P.segWork : P.kc->seg;
	const uint32_t mine = cnt[w * kSegStride], ext = queue_extent(cnt) / kSegs; // records per segment up to the extent
	for (uint32_t j = mine + threadIdx.x; j < ext; j += kBlock) {
		const uint32_t slot = seg_phys(w, j);
		if (what == 0)
			write_dead_ray(P.work, slot);
		else
			reinterpret_cast<float*>(&P.shadow.color[slot])[3] = 1.0f;
	}
}

// connect pre-pass: the sphere half of intersect_scene_simple (kernel.cu:168-172).  Any-hit does not
// depend on test order, so spheres go first and an occluded ray never enters the BVH.
// color.w (unused by the reference's 44-byte record) carries the flag.
__global__ void __launch_bounds__(kBlock) k_connect_spheres(const FrameParams P) {
	const uint32_t first = blockIdx.x * kBlock + threadIdx.x, stride = gridDim.x * kBlock;
	if (first < kTicketWords)
		P.k->extend_chunks[first * 32] = 0; // k_trace_flat's tickets, when this pre-pass opens the launch that ends a render (no set_wavefront_globals in front of it)
	__shared__ uint32_t cntSh[kSegs];
	if (threadIdx.x < kSegs)
		cntSh[threadIdx.x] = P.kc->seg[threadIdx.x * kSegStride];
	__syncthreads();
	uint32_t m = 0;
#pragma unroll
	for (uint32_t w = 0; w < kSegs; ++w)
		m = cntSh[w] > m ? cntSh[w] : m;
	const uint32_t n = ((m + 63u) >> 6) * (kSegs * 64u);
	for (uint32_t index = first; index < n; index += stride) {
		if ((((index >> 9) << 6) | (index & 63u)) >= cntSh[(index >> 6) & (kSegs - 1u)]) {
			reinterpret_cast<float*>(&P.shadow.color[index])[3] = 1.0f; // a hole at a segment's end: "occluded" retires it at the traversal's refill
			continue;
		}
		const float4 a = P.shadow.o_dx[index];
		const float4 b = P.shadow.dyz_cd_ix[index];
		const f3 o = mk3(a.x, a.y, a.z), d = mk3(a.w, b.x, b.y);
		const float closest = b.z;
		bool occluded = false;
#pragma unroll
		for (int i = TYR_NUM_SPHERES; i--;) {
			const float t = sphere_intersect(P.spheres[i], o, d);
			occluded = occluded || (t && (t + kEpsilon) < closest);
		}
		reinterpret_cast<float*>(&P.shadow.color[index])[3] = occluded ? 1.0f : 0.0f;
	}
}

// ======================================================================================
// The serial order, recovered as numbers (kernels.hpp "Queues"): shade wrote one byte per ray at its virtual slot --
// survived or not; rank(v) = survivors below v is the survivor's slot in the next iteration by the reference's serial
// ticket order (kernel.cu:607 with the atomics in slot order).  k_scan_words packs 64 bytes into one word and scans the
// words' counts inside 16384-slot blocks; the block that finishes last scans the blocks' totals.  v_lookup() adds the three parts.
// ======================================================================================
constexpr uint32_t kScanBlockSlots = 64u * kBlock;
__global__ void __launch_bounds__(kBlock) k_scan_words(const FrameParams P) {
	__shared__ uint32_t waveSum[kBlock / 64];
	const uint32_t n = *P.scanLive; // &k->n_live
	const uint32_t e = blockIdx.x * kBlock + threadIdx.x, first = e * 64u;
	if (blockIdx.x * kScanBlockSlots >= n)
		return; // (the whole block: the host sized the grid from an upper bound.  With n == 0 that is EVERY block, block 0 and the fold below
		        // included: an iteration without rays opens no successor -- k_shade's shadeOpensNext agrees, host/driver.cpp render_run_ahead
		        // states the invariant: nothing is ever queued behind an empty iteration)
	unsigned long long word = 0ull;
	if (first < n) {
		uint4* p = reinterpret_cast<uint4*>(P.survFlag + first);
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const uint4 q = p[k];
			// read and -- when this iteration's shade finished survivors in place (P.retireGhosts) -- cleared: such a survivor occupies
			// a slot of the NEXT iteration's order that nobody will write, and it must read "did not survive" there.  (Every other
			// slot below the next iteration's ray count is written by whoever makes or shades its ray.)
			if (P.retireGhosts)
				p[k] = make_uint4(0u, 0u, 0u, 0u);
			const uint32_t x[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
			for (int j = 0; j < 4; ++j) { // four bytes, each 0 or 1 -> four bits
				const uint32_t b = (x[j] & 1u) | ((x[j] >> 7) & 2u) | ((x[j] >> 14) & 4u) | ((x[j] >> 21) & 8u);
				word |= (unsigned long long)b << (16 * k + 4 * j);
			}
		}
		if (n - first < 64u)
			word &= (1ull << (n - first)) - 1ull; // bytes past the queue's end were never written
	}
	const uint32_t c = (uint32_t)__popcll(word);
	uint32_t incl = c;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t v = __shfl_up(incl, o, 64);
		if (lane >= (uint32_t)o)
			incl += v;
	}
	if (lane == 63u)
		waveSum[wave] = incl;
	__syncthreads();
	uint32_t before = 0, total = 0;
#pragma unroll
	for (uint32_t w = 0; w < kBlock / 64; ++w) {
		if (w < wave)
			before += waveSum[w];
		total += waveSum[w];
	}
	P.vWordOut[e] = word;
	P.vPreOut[e] = before + incl - c;
	// the blocks' totals become exclusive prefixes in the block that finishes last
	__shared__ uint32_t lastSh, carrySh;
	if (threadIdx.x == 0) {
		__hip_atomic_store(&P.vBlkOut[blockIdx.x], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the total has arrived before this block counts as done (no L2 write-back: see k_shade)
		const uint32_t nBlocksLive = (n + kScanBlockSlots - 1) / kScanBlockSlots;
		lastSh = atomicAdd(&P.k->scan_blocks_done, 1u) + 1u == nBlocksLive ? 1u : 0u;
		carrySh = 0;
	}
	__syncthreads();
	if (!lastSh)
		return;
	const uint32_t nBlocks = (n + kScanBlockSlots - 1) / kScanBlockSlots;
	for (uint32_t base = 0; base < nBlocks; base += kBlock) {
		const uint32_t i = base + threadIdx.x;
		const uint32_t v = i < nBlocks ? __hip_atomic_load(&P.vBlkOut[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
		uint32_t inc2 = v;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t u = __shfl_up(inc2, o, 64);
			if (lane >= (uint32_t)o)
				inc2 += u;
		}
		__syncthreads(); // waveSum is reused
		if (lane == 63u)
			waveSum[wave] = inc2;
		__syncthreads();
		uint32_t before2 = carrySh, total2 = 0;
#pragma unroll
		for (uint32_t w = 0; w < kBlock / 64; ++w) {
			if (w < wave)
				before2 += waveSum[w];
			total2 += waveSum[w];
		}
		if (i < nBlocks)
			P.vBlkOut[i] = before2 + inc2 - v;
		__syncthreads();
		if (threadIdx.x == 0)
			carrySh += total2;
		__syncthreads();
	}
	// the counter is left as the next scan expects it.  (set_wavefront_globals zeroes it too -- but a scan done by the next traversal launch,
	// TYR_TUNE_SCAN_IN_TRACE, runs BEHIND the next iteration's set_wavefront_globals, and the scan after it may be this kernel's)
	if (threadIdx.x == 0)
		P.k->scan_blocks_done = 0;
	// P.foldNextPrologue (tyr_render, an iteration whose successor is already being queued and cannot top the queue up: the budget
	// is spent): what would open that successor -- a one-block k_primary launch for set_wavefront_globals (kernel.cu:227-244) and
	// k_pad_holes in front of its traversal launch, two launches and two gaps between dependent kernels (~25 us of a ~550 us thin
	// iteration) -- is done here, by the block that ends this iteration's last kernel.  Everything it reads is final (shade's
	// launch has ended), everything it resets has been consumed (this iteration's work queue, the other set of shadow counters).
	if (P.foldNextPrologue) {
		pad_work_holes_block(P.next, P.segNext);   // the successor's work queue: class 0's segment ends
		pad_shadow_holes_block(P.shadow, P.kc->seg); // ... and the shadow rays it carries
		__syncthreads();
		wavefront_globals_for(P, P.segWork, P.kcPrev);
	}
}


// ======================================================================================
// blit_onto_framebuffer, kernel.cu:648-662 -> linear RGBA32F
// ======================================================================================
__global__ void __launch_bounds__(kBlock) k_resolve(const float4* __restrict__ blit, float4* __restrict__ out, uint32_t nPixels) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= nPixels)
		return;
	const float4 color = blit[i];
	const float r = color.x / color.w, g = color.y / color.w, b = color.z / color.w;
	constexpr float inv_gamma = 1.0f / 2.2f;
	out[i] = make_float4(dm::powf_det(r / (r + 1.f), inv_gamma), dm::powf_det(g / (g + 1.f), inv_gamma), dm::powf_det(b / (b + 1.f), inv_gamma),
		dm::powf_det(1.f / (1.f + 1.f), inv_gamma));
}

// ======================================================================================
// tyr_vecmath_probe: hip/vecmath.hpp (and the deterministic pow / exp) evaluated ON THE DEVICE over arrays, so that the
// functions every kernel is built from can be pinned to the vendored glm's answers (tests/golden/ref_glm.npz).
// Same op codes as oracle/ref_harness.cpp ref_glm.
// ======================================================================================
__global__ void __launch_bounds__(kBlock) k_vecmath_probe(int op, const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, uint32_t n, float* __restrict__ out) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n)
		return;
	const f3 A = ld3(a + 3 * i), B = ld3(b + 3 * i), Cc = ld3(c + 3 * i);
	f3 r = mk3(0.f, 0.f, 0.f);
	switch (op) {
	case 0: r.x = dot(A, B); break;
	case 1: r = cross(A, B); break;
	case 2: r = normalize(A); break;
	case 3: r.x = length(A); break;
	case 4: r = reflect(A, B); break;
	case 5: r = mk3(gmin(A.x, B.x), gmin(A.y, B.y), gmin(A.z, B.z)); break;
	case 6: r = mk3(gmax(A.x, B.x), gmax(A.y, B.y), gmax(A.z, B.z)); break;
	case 7: r = mk3(gclamp(A.x, B.x, B.y), gclamp(A.y, B.x, B.y), gclamp(A.z, B.x, B.y)); break;
	case 8: r = gmix(A, B, Cc.x); break;
	case 9: r.x = gsmoothstep(Cc.x, Cc.y, A.x); break;
	case 10: // exponent 0.5 is the path's closed form sqrt (sunsky.hpp: pow(inscatter * Fex, vec3(0.5)), sunsky.cu:66)
		r = mk3(B.x == 0.5f ? sqrtf(A.x) : dm::powf_det(A.x, B.x), B.y == 0.5f ? sqrtf(A.y) : dm::powf_det(A.y, B.y), B.z == 0.5f ? sqrtf(A.z) : dm::powf_det(A.z, B.z));
		break;
	case 11: r = A / Cc.x; break;
	case 12: r = A * Cc.x; break;
	case 13: r = Cc.x * A; break;
	case 14: r = mk3(dm::expf_det(A.x), dm::expf_det(A.y), dm::expf_det(A.z)); break;
	case 15: r = A * B; break;
	case 16: r = A / B; break;
	case 17: r = -A; break;
	case 18: r = A + B; break;
	case 19: r = A - B; break;
	default: break;
	}
	out[3 * i + 0] = r.x;
	out[3 * i + 1] = r.y;
	out[3 * i + 2] = r.z;
}

// ======================================================================================
// tyr_sunsky_probe: sun / sky / sunsky and the sun-cone sample exactly as k_shade evaluates them (one shared
// atmosphere() block, then the radiance the ray asked for; shade.hip "The atmosphere ..."), over arrays, so that the
// device arithmetic can be pinned to the reference's own sunsky.cu (tests/golden/ref_sunsky.npz).
// ======================================================================================
__global__ void __launch_bounds__(kBlock) k_sunsky_probe(const SunParams S, int which, const float* __restrict__ dirs, uint32_t n, float* __restrict__ out) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (which == 3) { // one stream, serial by definition: thread 0 draws the n samples in order
		if (i != 0)
			return;
		uint32_t seed = __float_as_uint(dirs[0]);
		for (uint32_t j = 0; j < n; ++j) {
			const f3 r = cone_sample(S, seed);
			out[3 * j + 0] = r.x;
			out[3 * j + 1] = r.y;
			out[3 * j + 2] = r.z;
		}
		out[3 * n] = __uint_as_float(seed);
		return;
	}
	if (i >= n)
		return;
	const f3 viewDir = ld3(dirs + 3 * i);
	f3 r;
	if (which == 2 && S.sunAngularDiameterCos == 1.0f) {
		r = mk3(1.0f, 0.0f, 0.0f); // sunsky.cu:118-119
	} else {
		const Atmosphere a = atmosphere(S, viewDir);
		r = which == 0 ? sun_radiance(S, a) : which == 1 ? sky_radiance(a) : sunsky_radiance(S, a);
	}
	out[3 * i + 0] = r.x;
	out[3 * i + 1] = r.y;
	out[3 * i + 2] = r.z;
}

// ======================================================================================
// extend_debug_BVH, kernel.cu:300-328 via intersect_scene_DEBUG (143-160) and CachedBVH::intersect_debug (bvh.h:164-209):
// the reference's compile-time BVH_DEBUG picture of the traversal cost.  One thread per slot on the pair nodes with the
// reference's counting rule (traversals = nodes visited - 1); a diagnostic, not a hot path.
// ======================================================================================
__global__ void __launch_bounds__(kBlock) k_extend_debug(const FrameParams P) {
	uint32_t refs[kStackSize];
	float ts[kStackSize];
	TravStack<0> st;
	st.bind(nullptr, refs, ts);
	st.reset();
	uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
	{
		const uint32_t c = slot >= P.classStride ? 1u : 0u, j = slot - c * P.classStride;
		const uint32_t* cnt = P.segWork + c * kClassWords;
		if (c >= kClasses || j >= queue_extent(cnt) || !slot_valid(cnt, j))
			return;
	}
	const float4 a = P.work.o_dx[slot];
	const float2 b = P.work.dyz[slot];
	float dist = kVeryFar; // kernel.cu:145; the spheres are commented out there (147-155)
	int prim = 0, traversals = 0;
	if (P.scene.rootRef != kRefDone) {
		VisitCount vc{ 0, 0 };
		const RayConst r = make_ray(mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y));
		const bool hit = bvh_closest<true>(P.scene, r, dist, prim, st, vc);
		traversals = (int)vc.nodes - 1; // bvh.h:172-175: the counter starts at -1 and counts loop iterations
		P.work.hit[slot] = make_float2(dist, __uint_as_float(hit ? (uint32_t)prim : 0u));
		if (st.overflow)
			atomicOr(&P.k->device_error, kErrStackOverflow);
	} else {
		P.work.hit[slot] = make_float2(dist, 0.0f);
	}
	float* px = reinterpret_cast<float*>(&P.blit[__float_as_int(P.work.direct_ix[slot].w)]);
	int green = (int)((0.0002f * (float)traversals) * 255.99f);
	green = green > 255 ? 255 : green;
	px[1] = (float)green;
	px[3] = 1.0f;
	if (traversals >= 70) { // "Color very costly regions distinctly"
		px[0] = (float)green;
		px[1] = 0.0f;
	}
}

// ---- launch wrappers ---------------------------------------------------------------------

constexpr uint32_t kPrepassMaxBlocks = 8192; // 8 waves of 256 threads per SIMD of a 256-CU part: enough to stream at full rate
void launch_pad_holes(const FrameParams& P, bool workQueue, bool shadowQueue, hipStream_t stream, bool resetTickets) {
	hipLaunchKernelGGL(k_pad_holes, dim3(2 * kSegs), dim3(kBlock), 0, stream, P, workQueue ? 1u : 0u, shadowQueue ? 1u : 0u, resetTickets ? 1u : 0u);
}
void launch_primary(const FrameParams& P, uint32_t maxNew, hipStream_t stream) {
	// always launched: its last block is set_wavefront_globals
	hipLaunchKernelGGL(k_primary, dim3(maxNew ? blocks_for(maxNew) : 1u), dim3(kBlock), 0, stream, P);
}
void launch_scan(const FrameParams& P, uint32_t maxLive, hipStream_t stream) {
	if (maxLive == 0)
		return;
	hipLaunchKernelGGL(k_scan_words, dim3((maxLive + kScanBlockSlots - 1) / kScanBlockSlots), dim3(kBlock), 0, stream, P);
}
void launch_vecmath_probe(int op, const float* a, const float* b, const float* c, uint32_t n, float* out, hipStream_t stream) {
	hipLaunchKernelGGL(k_vecmath_probe, dim3(blocks_for(n ? n : 1)), dim3(kBlock), 0, stream, op, a, b, c, n, out);
}
void launch_sunsky_probe(const SunParams& S, int which, const float* dirs, uint32_t n, float* out, hipStream_t stream) {
	hipLaunchKernelGGL(k_sunsky_probe, dim3(blocks_for(which == 3 ? 1 : n)), dim3(kBlock), 0, stream, S, which, dirs, n, out);
}
void launch_extend_debug(const FrameParams& P, uint32_t maxLive, hipStream_t stream) {
	if (maxLive != 0)
		hipLaunchKernelGGL(k_extend_debug, dim3(blocks_for(maxLive)), dim3(kBlock), 0, stream, P);
}
void launch_extend_spheres(const FrameParams& P, uint32_t nSurvivors, hipStream_t stream, uint32_t maxLive) {
	// the grid also covers the hole padding at the ends of class 0's segments, which walks up to the extent of ALL this
	// iteration's rays: a handful of survivors in front of a full top-up must not leave that walk to a single block
	const uint32_t walk = std::max(nSurvivors, maxLive);
	if (nSurvivors != 0)
		hipLaunchKernelGGL(k_extend_spheres, dim3(std::min(blocks_for(walk), kPrepassMaxBlocks)), dim3(kBlock), 0, stream, P);
}
void launch_connect_spheres(const FrameParams& P, uint32_t maxShadow, hipStream_t stream) {
	hipLaunchKernelGGL(k_connect_spheres, dim3(std::min(blocks_for(maxShadow), kPrepassMaxBlocks)), dim3(kBlock), 0, stream, P);
}
void launch_resolve(const float4* blit, float4* out, uint32_t nPixels, hipStream_t stream) {
	hipLaunchKernelGGL(k_resolve, dim3(blocks_for(nPixels)), dim3(kBlock), 0, stream, blit, out, nPixels);
}

} // namespace tyr
