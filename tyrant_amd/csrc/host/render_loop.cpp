// render_loop.cpp -- launch_kernels' loop behind the C ABI (reference: kernel.cu:664-748 and its caller, main.cpp:164-170):
// tyr_launch_kernels (one wavefront iteration, done when it returns) and tyr_render (a primary-ray budget, merged traversal launches,
// one iteration ahead of the counts).  DESIGN.md section 5.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "driver_internal.hpp"

using namespace tyr;
using namespace tyr::drv;

extern "C" {

// ---- the per-frame entry point --------------------------------------------------------------
// One wavefront iteration.  pipelined = false is launch_kernels as the reference has it: primary, extend, shade, connect on
// one stream, done when it returns (kernel.cu:719-733).  Inside tyr_render (pipelined, merged launches) connect(i) rides in
// the traversal launch of iteration i + 1 and the call returns as soon as shade's counts are on the host.
#ifdef TYR_LAUNCH_ANATOMY
// TYR_ANATOMY=2: the per-wave records k_trace_flat's anatomy build leaves in the next queue's hit column
static void print_wave_anatomy(const float2* dHit, unsigned long long feedTicks) {
	std::vector<float2> rec(4 * 8192);
	if (hipMemcpy(rec.data(), dHit, rec.size() * sizeof(float2), hipMemcpyDeviceToHost) != hipSuccess)
		return;
	std::vector<float> drain, normal, wide, perTrip, perStep, liveExh, liveWide, trips, steps, passes;
	for (uint32_t w = 0; w < 8192; ++w) {
		const float tExh = rec[w].x, tEnd = rec[w].y, tWide = rec[8192 + w].x;
		if (!(tExh > 0.0f) || !(tEnd >= tExh) || !(tEnd < 1e5f))
			continue;
		const uint32_t lv = (uint32_t)rec[8192 + w].y;
		const float nTrips = rec[16384 + w].x, nSteps = rec[16384 + w].y;
		drain.push_back(tEnd - tExh);
		normal.push_back((tWide > 0.0f ? tWide : tEnd) - tExh);
		wide.push_back(tWide > 0.0f ? tEnd - tWide : 0.0f);
		if (nTrips > 0.0f)
			perTrip.push_back(((tWide > 0.0f ? tWide : tEnd) - tExh) / nTrips);
		if (nSteps > 0.0f && tWide > 0.0f)
			perStep.push_back((tEnd - tWide) / nSteps);
		liveExh.push_back((float)(lv & 255u));
		liveWide.push_back((float)(lv >> 8));
		trips.push_back(nTrips);
		steps.push_back(nSteps);
		passes.push_back(rec[24576 + w].x);
	}
	auto pct = [](std::vector<float>& v, double p) {
		if (v.empty())
			return 0.0f;
		const size_t k = (size_t)(p * (v.size() - 1));
		std::nth_element(v.begin(), v.begin() + k, v.end());
		return v[k];
	};
	auto line = [&](const char* name, std::vector<float>& v) { std::fprintf(stderr, "[anatomy]    %-44s n %5zu  median %8.2f  90 %% %8.2f  99 %% %8.2f  max %8.2f\n", name, v.size(), pct(v, 0.5), pct(v, 0.9), pct(v, 0.99), pct(v, 1.0)); };
	std::fprintf(stderr, "[anatomy]  per wave, after the queue ran out (feed %.1f us):\n", feedTicks / 100.0);
	line("drain: exit - 'used up' [us]", drain);
	line("  one ray to a lane [us]", normal);
	line("  four lanes to a ray [us]", wide);
	line("rays held when the queue ran out", liveExh);
	line("rays held on going wide", liveWide);
	line("descent trips one ray to a lane", trips);
	line("outer passes (leaf rounds) one ray to a lane", passes);
	line("steps four lanes to a ray", steps);
	line("us per trip, one ray to a lane", perTrip);
	line("us per step, four lanes to a ray", perStep);
	{
		// the feed phase: microseconds per descent trip while the queue lasted
		std::vector<float> feedBusy;
		for (uint32_t w = 0; w < 8192; ++w) {
			const float tExh = rec[w].x, n = rec[24576 + w].y;
			if (!(tExh > 0.0f) || !(n > 0.0f))
				continue;
			feedBusy.push_back(tExh / n);
		}
		line("feed phase: us per trip", feedBusy);
	}
	// the launch ends with these: the five waves that left last
	std::vector<uint32_t> order;
	for (uint32_t w = 0; w < 8192; ++w)
		if (rec[w].x > 0.0f && rec[w].y >= rec[w].x && rec[w].y < 1e5f)
			order.push_back(w);
	std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return rec[a].y > rec[b].y; });
	for (size_t i = 0; i < order.size() && i < 5; ++i) {
		const uint32_t w = order[i];
		const float tExh = rec[w].x, tEnd = rec[w].y, tWide = rec[8192 + w].x;
		const uint32_t lv = (uint32_t)rec[8192 + w].y;
		std::fprintf(stderr, "[anatomy]    last wave %zu: exit at %.1f us; queue used up at %.1f (%u rays held), %.1f us / %.0f trips one ray to a lane, %.1f us / %.0f steps four lanes to a ray (from %u rays)\n", i + 1, tEnd, tExh,
		             lv & 255u, (tWide > 0.0f ? tWide : tEnd) - tExh, rec[16384 + w].x, tWide > 0.0f ? tEnd - tWide : 0.0f, rec[16384 + w].y, lv >> 8);
	}
}
#endif

static int launch_iteration(tyr_ctx* c, bool pipelined) {
	// hK is current: every entry point that enqueues work ends with sync_counters
	int rc = stage_begin(c);
	if (rc)
		return rc;
	const uint32_t nNew = planned_new(c), nLive = c->hK->primary_ray_cnt + nNew;
	FrameParams P = make_params(c);
	if (c->cfg.flags & TYR_FLAG_DEBUG_BVH) {
		// kernel.cu:720-722 under BVH_DEBUG: primary_rays, set_wavefront_globals, extend_debug_BVH -- no shade, no connect
		// (nothing survives: the next call regenerates the whole queue from the cursor)
		enqueue_primary(c, P, nNew);
		enqueue_extend(c, P, nLive, nLive - nNew);
		HIPCHK(hipGetLastError());
		rc = sync_counters(c);
		collect_timings(c);
		stage_end(c);
		return rc ? rc : check_device_error(c);
	}
	const bool merge = pipelined && merged_render(c);
	if (merge && c->tuning.foldSpheres) {
		P.foldSpheres = 1u; // this iteration's shade does the sphere halves for the rays it emits
		P.resolveShadows = c->tuning.resolveShadows ? 1u : 0u; // ... and answers the shadow rays that cannot reach a triangle
		P.retireGhosts = (c->tuning.retireSky && c->unboundedRender) ? 1u : 0u; // ... and finishes the survivors that will hit nothing (a render cut short would see their pixels an iteration early)
	}
	if (merge && c->tuning.retireSky)
		P.retireSky = 1u;   // ... and k_primary finishes the camera rays that hit nothing
	enqueue_primary(c, P, nNew);
	if (merge) { // every traversal launch of a merged render is k_trace_flat; the first one has no shadow rays to carry yet
		const uint32_t carried = c->shadowPending ? c->shadowPendingMax : 0u;
		c->shadowPending = false;
		enqueue_trace(c, P, nLive, nLive - nNew, carried);
		enqueue_shade(c, P, nLive);
	} else {
		if ((rc = flush_pending_shadow(c))) // (a render whose merge setting changed between iterations: never, but cheap)
			return rc;
		enqueue_extend(c, P, nLive, nLive - nNew);
		enqueue_shade(c, P, nLive);
	}
	if (merge) {
		// everything the host needs to launch iteration i + 1 (survivors, budget, the shadow-ray count) is final here
		HIPCHK(hipMemcpyAsync(c->hK, c->dK, sizeof(DevCounters), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipEventRecord(c->evSnapshot, c->stream));
		HIPCHK(hipGetLastError());
		HIPCHK(hipEventSynchronize(c->evSnapshot));
		c->shadowPending = c->hK->shadow_ray_cnt != 0;
		c->shadowPendingMax = c->hK->shadow_ray_cnt;
#if defined(TYR_QUAD_STATS) || defined(TYR_LAUNCH_ANATOMY)
		if (std::getenv("TYR_ANATOMY")) {
			// launch anatomy of this iteration's traversal launch (s_memrealtime, 100 MHz): first wave's start, first wave to
			// find the queue used up, last wave's exit
			const unsigned long long t0 = ~c->hK->debug[13], tx = ~c->hK->debug[14], t1 = c->hK->debug[15];
			std::fprintf(stderr, "[anatomy] iteration %u: %u rays: feed %.1f us, drain %.1f us", c->iter, nLive, (tx - t0) / 100.0, (t1 - tx) / 100.0);
#ifdef TYR_QUAD_STATS
			std::fprintf(stderr, "; longest ray %llu quad steps, rays with > 64 / 128 / 256 steps: %llu / %llu / %llu (running totals)", c->hK->debug[12], c->hK->debug[9], c->hK->debug[10], c->hK->debug[11]);
#endif
			std::fprintf(stderr, "\n");
#ifdef TYR_LAUNCH_ANATOMY
			if (std::getenv("TYR_ANATOMY")[0] == '2' && P.N > 32768u)
				print_wave_anatomy(P.next.hit, tx - t0);
#endif
		}
#endif
	} else {
		enqueue_connect(c, P, nLive); // at most one shadow ray per live ray
		HIPCHK(hipGetLastError());
		rc = sync_counters(c); // kernel.cu:733 cudaDeviceSynchronize
	}
	collect_timings(c);
	stage_end(c);
	return rc ? rc : check_device_error(c);
}

int tyr_launch_kernels(tyr_ctx* c) {
	if (!c)
		return TYR_ERR_INVALID;
	if (!c->haveScene)
		return TYR_ERR_NO_SCENE;
	int rc = use_device(c);
	if (rc)
		return rc;
	return launch_iteration(c, false);
}

// ---- tyr_render, one iteration ahead of the counts (TYR_TUNE_RUN_AHEAD) -------------------------
// Between shade(i) and the first kernel of iteration i + 1 the stream used to run dry for ~20 us: the counters travel
// to the host, the host wakes up, sizes the grids and launches.  Nothing in iteration i + 1 needs the host for that:
// k_primary (and set_wavefront_globals in its last block) compute the top-up from the device's counters (kernel.cu:253, 227-244 do the same), the
// persistent kernels read their item counts there, k_shade its tile count.  So the host queues iteration i + 1 right
// behind iteration i, sizing every grid from upper bounds it can already compute -- survivors(i) <= live(i), shadow rays
// (i) <= live(i), and live(i), the budget and the top-up of i + 1's predecessors follow exactly from the last counts that
// DID arrive -- and waits for iteration i's counts afterwards, with iteration i + 1 already running or queued.
// It learns one iteration late that the render has ended (no survivors, no budget): that last iteration has no rays of
// its own and traces the final shadow rays -- the connect launch a merged render needs at its end anyway -- and the
// host takes back its frame counter, queue swap and iteration parity, so that the ctx is where the reference's loop
// would have left it (kernel.cu:735-745, main.cpp:169).
struct IterationPlan {
	uint32_t nNew, nLive, nSurvivors, carried; // exact values or upper bounds; carried: shadow rays of the iteration before (0: none to trace)
};
// foldNext: this iteration's k_scan_words also opens the next one (no top-up can follow and the next one IS going to be queued);
// prologueDone: the previous iteration's did that for this one -- no k_primary launch, no k_pad_holes
static int enqueue_merged_iteration(tyr_ctx* c, const IterationPlan& p, bool begun, bool foldNext = false, bool prologueDone = false) {
	int rc = begun ? TYR_OK : stage_begin(c);
	if (rc)
		return rc;
	const int set = static_cast<int>(c->iter & 1u);
	FrameParams P = make_params(c);
	if (c->tuning.foldSpheres) {
		P.foldSpheres = 1u;
		P.resolveShadows = c->tuning.resolveShadows ? 1u : 0u;
		P.retireGhosts = (c->tuning.retireSky && c->unboundedRender) ? 1u : 0u;
	}
	if (c->tuning.retireSky)
		P.retireSky = 1u;
	const bool aside = foldNext && c->tuning.scanInTrace != 0;
	P.foldNextPrologue = (foldNext && !aside) ? 1u : 0u;
	P.shadeOpensNext = aside ? 1u : 0u; // k_shade's last block opens the next iteration, whose traversal launch does this iteration's slot scan on its way in (TYR_TUNE_SCAN_IN_TRACE)
	if (aside) {
		P.scanSet = static_cast<uint32_t>(set);
		P.scanLive = &c->dK->scan_live[set];
	}
	P.prologueDone = prologueDone ? 1u : 0u;
	// the counts the loop waits for: written by k_shade's last block into pinned host memory (nothing in the stream between this shade
	// launch and the next traversal launch; a ctx that times its stages still has their event pairs there)
	const bool kernelSnap = c->tuning.kernelSnapshot != 0;
	c->snapSeqOf[set] = 0;
	if (kernelSnap) {
		if (++c->snapSeq == 0u)
			++c->snapSeq;
		c->snapSeqOf[set] = c->snapSeq;
		P.hostSnap = c->hostSnapDev[set];
		P.snapSeq = c->snapSeq;
	}
	if (!prologueDone)
		enqueue_primary(c, P, p.nNew);
	enqueue_trace(c, P, p.nLive, p.nSurvivors, p.carried);
	enqueue_shade(c, P, p.nLive);
	if (!kernelSnap) {
		HIPCHK(hipMemcpyAsync(c->hSnap[set], c->dK, sizeof(DevCounters), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(hipEventRecord(c->evSnap[set], c->stream));
	}
	HIPCHK(hipGetLastError());
	stage_end(c);
	return TYR_OK;
}
static bool run_ahead_eligible(const tyr_ctx* c) {
#if defined(TYR_QUAD_STATS) || defined(TYR_LAUNCH_ANATOMY)
	return false; // the instrumented builds print per-iteration records from the host mirror (launch_iteration)
#else
	const bool wanted = c->tuning.runAhead != 0; // (2 meant "queues of at most 6 Mi slots" while a render's last iteration was followed by an empty one: render_run_ahead's lastBirth)
	return wanted && merged_render(c) && c->blit != nullptr;
#endif
}
static int render_run_ahead(tyr_ctx* c, uint32_t max_iterations, uint32_t& it) {
	it = 0;
	if (max_iterations == 0)
		return TYR_OK;
	int rc = flush_pending_shadow(c);
	if (rc)
		return rc;
	if ((rc = stage_begin(c))) // may reset the accumulation and the survivor count (kernel.cu:712-718): before the plan is made
		return rc;
	const uint64_t N = c->cfg.queue_size;
	// exact state in front of iteration 0 (hK is current: every entry point ends with sync_counters)
	uint64_t s = c->hK->primary_ray_cnt, budget = c->hK->budget_remaining;
	uint32_t nNew = static_cast<uint32_t>(std::min<uint64_t>(N - s, budget));
	uint32_t live = static_cast<uint32_t>(s) + nNew; // live(enq - 1), exact
	budget -= nNew;                                   // budget left behind iteration enq - 1, exact
	const uint32_t iter0 = c->iter; // iteration j of this render is the ctx's iteration iter0 + j: its events and its counters use set (iter0 + j) & 1
	// Will iteration j + 1 be queued without a look at iteration j's counts, and can it do without a top-up?  Then iteration j's
	// last kernel opens it (FrameParams::foldNextPrologue): set_wavefront_globals and the hole padding cost a ~5 us launch and a
	// gap between dependent kernels each, every iteration.  Both answers follow from what the host knows when it queues j: the
	// budget left behind j (exact once it is zero) and the last iteration that gave birth to rays.
	const bool mayFold = c->tuning.foldPrologue != 0 && c->tuning.foldSpheres != 0;
	// INVARIANT the kernels rely on: an iteration that turns out to have no rays (n_live == 0: the one queued ahead of its predecessor's
	// counts for nothing) is never followed by another -- the loop below returns when it sees "budget == 0 && s == 0" -- so the kernels that
	// would open its successor (k_scan_words' and k_shade's last blocks) skip that when n_live is 0, and the counters of the last real
	// iteration stay what tyr_shadow_export reads.
	auto queued_ahead_behind = [&](uint32_t j, uint64_t budgetBehindJ, uint32_t lastBirthAtJ) { return j + 1 < max_iterations && (budgetBehindJ != 0 || j < lastBirthAtJ + static_cast<uint32_t>(kMaxBounces)); };
	bool folded = mayFold && budget == 0 && queued_ahead_behind(0, budget, 0); // (of the iteration queued last: its k_scan_words has opened the next one)
	if ((rc = enqueue_merged_iteration(c, IterationPlan{ nNew, live, static_cast<uint32_t>(s), 0u }, true, folded, false)))
		return rc;
	uint32_t enq = 1;
	// The last iteration (of this render) that gave birth to rays: a primary ray survives at most kMaxBounces times
	// (kernel.cu:600-607), so shade of iteration lastBirth + kMaxBounces leaves no survivor -- once the budget is spent the
	// render's end is known in advance and no iteration has to be queued ahead for nothing.  (Survivors the ctx held when the
	// render began count as born in iteration 0: their bounce counts are not known here.)
	uint32_t lastBirth = 0;
	for (;;) {
		// iterations 0 .. enq - 1 are queued; the counts of 0 .. enq - 2 have arrived
		bool ahead = false;
		bool foldedPrev = folded; // whether the iteration whose counts are awaited below (enq - 1) opened its successor
		uint32_t frameBefore = c->frame;
		const uint32_t shadowSetBefore = c->shadowSet; // (enqueue_shade of an iteration queued ahead moves it: an iteration that turns out empty must give it back, or tyr_shadow_export would read the empty iteration's counters)
		const bool foldedBefore = c->lastShadeFolded;
		const bool canHaveSurvivors = budget != 0 || enq - 1 < lastBirth + static_cast<uint32_t>(kMaxBounces); // of iteration enq - 1
		if (enq < max_iterations && canHaveSurvivors) {
			const uint32_t liveMax = static_cast<uint32_t>(std::min<uint64_t>(N, static_cast<uint64_t>(live) + budget));
			const uint32_t newMax = static_cast<uint32_t>(std::min<uint64_t>(N, budget));
			const bool opened = folded; // iteration enq - 1's k_scan_words has done this one's set_wavefront_globals and hole padding
			foldedPrev = folded;
			folded = mayFold && budget == 0 && queued_ahead_behind(enq, 0, lastBirth); // (budget == 0: iteration enq tops nothing up, gives birth to nothing)
			if ((rc = enqueue_merged_iteration(c, IterationPlan{ newMax, liveMax, live, live }, false, folded, opened))) {
				(void)hipStreamSynchronize(c->stream); // (the failed iteration may be partly queued; nothing of it is the render's)
				c->scanCarried = false;
				c->shadowSet = shadowSetBefore;
				c->lastShadeFolded = foldedBefore;
				return rc;
			}
			ahead = true;
		}
		// The render's end is known (the budget is spent, iteration enq - 1 cannot leave a survivor): the launch that traces its last
		// shadow rays goes out now, sized from an upper bound (at most one shadow ray per ray; the kernel takes its counts from the
		// device), instead of after the ~25 us it takes the counts to reach the host and the launch to reach the GPU.
		bool flushedEarly = false;
		if (!ahead && mayFold && budget == 0 && !canHaveSurvivors) {
			c->shadowPending = true;
			c->shadowPendingMax = live;
			if ((rc = flush_pending_shadow(c)))
				return rc;
			flushedEarly = true;
		}
		const int set = static_cast<int>((iter0 + enq - 1) & 1u);
		// a failure from here on leaves an iteration queued that the render will never own: drain the stream and take the
		// host's bookkeeping of it back, so that the ctx is where its last completed iteration left it
		auto abandon = [&](int code) {
			// (also when nothing was queued ahead: kernels of iteration enq - 1 may still be running and would go on writing the snapshot
			// record and the blit buffer behind an error return)
			(void)hipStreamSynchronize(c->stream);
			c->scanCarried = false;
			if (ahead) {
				c->frame = frameBefore;
				c->cur ^= 1;
				c->iter--;
				c->shadowPending = false;
				c->shadowSet = shadowSetBefore;
				c->lastShadeFolded = foldedBefore;
			}
			return code;
		};
		if (c->snapSeqOf[set] != 0u) {
			// the kernel-written snapshot: poll its stamp (the stream is looked at now and then: a fault must not hang the host)
			volatile tyr::HostSnap* const hs = c->hostSnap[set];
			const uint32_t want = c->snapSeqOf[set];
			// An iteration is tens to hundreds of microseconds: spin.  The stream is looked at every 16 K spins -- idle (or failed)
			// without the stamp is the only verdict; a slow iteration (a serialising profiler, a very large scene) is waited for as
			// hipStreamSynchronize would, and once the wait is past a few milliseconds the core is given back between looks.
			for (uint32_t spins = 0; __atomic_load_n(&hs->seq, __ATOMIC_ACQUIRE) != want; ++spins) {
				if ((spins & 0x3fffu) == 0x3fffu) {
					const hipError_t q = hipStreamQuery(c->stream);
					if (q != hipErrorNotReady && __atomic_load_n(&hs->seq, __ATOMIC_ACQUIRE) != want) // idle (or failed) without the stamp
						return abandon(q == hipSuccess ? TYR_ERR_DEVICE : static_cast<int>(q));
					if (spins >= (1u << 20))
						std::this_thread::sleep_for(std::chrono::microseconds(50));
				}
#if defined(__x86_64__)
				__builtin_ia32_pause();
#endif
			}
			c->hK->primary_ray_cnt = hs->survivors;
			c->hK->shadow_ray_cnt = hs->shadows;
			c->hK->device_error = hs->device_error;
			c->hK->n_live = live;
		} else {
			const hipError_t e = hipEventSynchronize(c->evSnap[set]);
			if (e != hipSuccess)
				return abandon(static_cast<int>(e));
			std::memcpy(c->hK, c->hSnap[set], sizeof(DevCounters));
		}
		if (c->snapSeqOf[set] == 0u && foldedPrev) {
			// iteration enq - 1's k_scan_words ran the next iteration's set_wavefront_globals in front of this snapshot: the two counts
			// the host steers by were kept aside (DevCounters::reserved0 / reserved1), n_live already reads the next iteration's
			c->hK->primary_ray_cnt = c->hK->reserved0;
			c->hK->shadow_ray_cnt = c->hK->reserved1;
			c->hK->n_live = live;
		}
		collect_timings_of(c, set);
		s = c->hK->primary_ray_cnt; // survivors of iteration enq - 1
		const uint32_t shadows = c->hK->shadow_ray_cnt;
		if ((rc = check_device_error(c)))
			return abandon(rc);
		it = enq; // (counted once it is known to have completed without a device error)
		if (budget == 0 && s == 0) { // kernel loop of the reference's caller: nothing left to trace or to start
			if (ahead) {
				// iteration enq was queued for nothing but the shadow rays of iteration enq - 1: take the host state back
				c->frame = frameBefore;
				c->cur ^= 1;
				c->iter--;
				c->shadowPending = false;
				c->shadowSet = shadowSetBefore;
				c->lastShadeFolded = foldedBefore;
				c->scanCarried = false; // (the empty iteration's shade launch left no scan behind: its last block opens nothing when n_live is 0)
				c->runAheadUndo = true;
				c->undoLive = live;
				c->undoShadows = shadows;
			} else {
				c->shadowPending = !flushedEarly && shadows != 0;
				c->shadowPendingMax = shadows;
			}
			return TYR_OK;
		}
		if (!ahead && enq >= max_iterations) { // max_iterations reached
			c->shadowPending = !flushedEarly && shadows != 0;
			c->shadowPendingMax = shadows;
			return TYR_OK;
		}
		// iteration enq is real; what it does, exactly, now that its predecessor's survivors are known
		nNew = static_cast<uint32_t>(std::min<uint64_t>(N - s, budget));
		if (nNew != 0)
			lastBirth = enq;
		live = static_cast<uint32_t>(s) + nNew;
		budget -= nNew;
		if (!ahead) {
			// (it was not queued ahead because no survivor was expected, and there are some: cannot happen while a ray survives
			// at most kMaxBounces times -- queued now, from the exact counts, rather than trusted)
			folded = false;
			if ((rc = enqueue_merged_iteration(c, IterationPlan{ nNew, live, static_cast<uint32_t>(s), flushedEarly ? 0u : shadows }, false, false, false)))
				return rc;
		}
		++enq;
	}
}

int tyr_render(tyr_ctx* c, uint32_t spp, uint32_t max_iterations, uint32_t* iterations_out) {
	if (!c)
		return TYR_ERR_INVALID;
	int rc = tyr_set_budget(c, static_cast<uint64_t>(spp) * c->localPixels);
	if (rc)
		return rc;
	if (!c->haveScene)
		return TYR_ERR_NO_SCENE;
	uint32_t it = 0;
	c->unboundedRender = max_iterations == 0xFFFFFFFFu;
	if (run_ahead_eligible(c)) {
		if ((rc = use_device(c)))
			return rc;
		rc = render_run_ahead(c, max_iterations, it);
	} else {
		while (it < max_iterations) {
			if ((rc = launch_iteration(c, true)))
				break;
			++it;
			if (c->hK->budget_remaining == 0 && c->hK->primary_ray_cnt == 0)
				break;
		}
	}
	{
		// the last shadow rays; counters refreshed (connect's included), nothing in flight when this returns
		int rcj = rc ? TYR_OK : flush_pending_shadow(c);
		c->shadowPending = false;
		if (!rcj)
			rcj = sync_counters(c);
		if (!rcj)
			collect_timings(c); // (an iteration queued ahead may still have had its event pairs out)
		if (c->runAheadUndo) {
			// the empty iteration's set_wavefront_globals zeroed the live and shadow counts of the last real iteration
			c->runAheadUndo = false;
			if (!rcj) {
				c->hK->n_live = c->undoLive;
				c->hK->shadow_ray_cnt = c->undoShadows;
				rcj = push_counters(c);
			}
		}
		if (!rc)
			rc = rcj ? rcj : check_device_error(c);
	}
	if (iterations_out)
		*iterations_out = it;
	return rc;
}

} // extern "C"
