// kernels.hpp -- launch interface between the host driver (host/driver.cpp) and the
// gfx950 kernels (hip/kernels.hip): device-side data layout and by-value kernel arguments.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../../include/tyr_c.h"
#include "sunsky.hpp"
#include "traverse.hpp"

namespace tyr {

// ---- ray queues in HBM: structure of arrays, 16/8/4-byte lanes --------------------------
// The reference's RayQueue is a 60-byte AoS record (variables.h:24-34): a wave touching one
// field strides 60 B per lane.  Here each kernel reads exactly the arrays it needs with
// consecutive lanes on consecutive 16/8/4-byte elements (1 KiB / 512 B / 256 B per wave
// instruction):
//   extend  reads  o_dx + dyz (24 B)                     writes hit (8 B)
//   shade   reads  o_dx + dyz + hit + direct_ix + flags (52 B), writes 44 B per survivor
struct RayQ {
	float4* o_dx;        // origin.xyz, direction.x
	float2* dyz;         // direction.y, direction.z
	float4* direct_ix;   // direct.rgb (path throughput), pixel index (int bits)     variables.h:27,31
	uint32_t* flags;     // bounces | lastSpecular << 8                               variables.h:30,33
	float2* hit;         // distance, identifier (int bits; bit 31 = sphere)         variables.h:28,29,32
	uint32_t* key;       // the ray's VIRTUAL slot (see "Queues" below): bit 31 = kKeyIndirect
};
constexpr uint32_t kHitSphere = 0x80000000u;
// ---- Queues: physically unordered, virtually in the serial ticket order ---------------------------------------------
// The reference appends survivors with atomicAdd(&primary_ray_cnt, 1) (kernel.cu:607) and seeds a ray's random numbers
// with its SLOT (kernel.cu:363), so "fixed seed" is only defined against one serial order; rounds 1-2 reproduced that
// order physically (a stable device-wide compaction with a decoupled look-back inside k_shade).  That made every
// tile's output position depend on every earlier tile -- shade could not start before the last straggler of the
// traversal launch had finished.  Since round 3 only the NUMBER is kept: every ray carries its virtual slot (the slot
// the serial order gives it), the RNG is seeded from it, and where the record physically lies is free:
//   * a queue is eight segments; segment w owns the 64-slot chunks w, w + 8, w + 16, ... of the arrays, filled in
//     order (record j of segment w lies at slot ((j / 64) * 8 + w) * 64 + j % 64).  A producer block appends its tile's
//     records to ONE segment with ONE atomic on that segment's counter (eight counters, 128 bytes apart: a single
//     word serves only ~88 returning atomics per microsecond): k_primary block b to segment b % 8, shade tile t to
//     segment (t / 2) % 8 -- tiles 2k and 2k + 1 hold records [64k, 64k + 64) of all eight segments, so the tiles that
//     feed one segment hold an eighth of every segment and no choice of surviving rays can hand a segment more than
//     N/8 + 1024 records; with the top-up's eighth of the new primaries on top: segCap = 15 N/64 + 4096 (host/driver.cpp);
//   * consumers walk physical slots [0, extent) and skip the few holes at the segments' ends (slot_valid);
//   * shade(i) writes one byte per ray at its virtual slot v: survived or not.  A scan of those bytes (k_scan_*) gives
//     rank(v) = survivors with a lower virtual slot = the survivor's slot in iteration i + 1 by the serial order.  The
//     survivor's record carries v | kKeyIndirect; shade(i + 1) looks rank(v) up in the scan's tables (v_lookup).
//     Fresh primary rays carry their slot directly (survivors + ticket, kernel.cu:254).
//   * a RAY queue is two such sets of segments, one per CLASS: class 0 holds the rays that may enter the tree (they pass
//     the root box), class 1 the rays that cannot (sky, ground sphere, the other spheres: 61 % of the extend rays of a C3
//     render).  Whoever makes a ray knows which it is -- k_primary and k_shade have origin and direction in registers --
//     and the traversal kernel is only ever handed class 0; the sphere pre-pass and shade walk both.  Class c's records lie
//     at slots [c * classStride, ...) of the same arrays.
constexpr uint32_t kKeyIndirect = 0x80000000u;   // the low bits are the slot of the PREVIOUS iteration: look the rank up
constexpr uint32_t kKeyMask = 0x3fffffffu;
constexpr uint32_t kSegs = 8;                    // = kTicketWords (k_trace_flat's ticket word w draws the chunks of segment w)
constexpr uint32_t kSegStride = 32;              // uint32 words between two segment counters (128 bytes)
constexpr uint32_t kClasses = 2;                 // ray queues: 0 = may enter the tree, 1 = cannot
constexpr uint32_t kClassWords = kSegs * kSegStride; // uint32 words between the counters of class 0 and class 1
struct VTable {                                  // scan of one iteration's survive bytes, 64 virtual slots per entry
	const unsigned long long* word;              // bit b of word[e]: slot 64 e + b survived
	const uint32_t* pre;                         // survivors in front of word e inside its 16384-slot block
	const uint32_t* blk;                         // survivors in front of block (e / 256)
};

// ShadowQueue (variables.h:36-42), 44 B -> 32 B read by every connect ray + 16 B only when visible
struct ShadowQ {
	float4* o_dx;        // origin.xyz, direction.x
	float4* dyz_cd_ix;   // direction.y, direction.z, closestDistance, buffer_index (int bits)
	float4* color;       // color.rgb, occluded-by-a-sphere flag of the connect pre-pass
	uint32_t* key;       // virtual slot of the ray that emitted it (tyr_shadow_export sorts by it; the kernels never read it)
};

// kernel.cu:211-224 device counters + the extensions (budget, totals, visit counters)
constexpr uint32_t kTicketWords = 8; // = XCDs: block b draws from word b % 8 first (round-robin block placement)
// INVARIANT of the "block that finishes last does X" patterns (k_primary -> set_wavefront_globals, k_shade's totals, k_scan_words'
// block totals): they count finished blocks with relaxed agent-scope atomics behind an `s_waitcnt vmcnt(0)`, NOT behind a
// release fence (a fence per block is an L2 write-back per block: 2.4 ms per 16.6 M-ray top-up when it was tried).  That is
// only sound while everything the finishing block READS of the others' output was written with agent-scope atomics (the
// counters, vBlkOut) or is only OVERWRITTEN by it (counts the others have finished reading).  A plain store added to
// something a finaliser reads would race silently across the XCDs' L2s: use __hip_atomic_store / atomicAdd there.
// what k_shade's last block tells the host directly (FrameParams::hostSnap): the counts of the iteration, then -- behind a system-scope
// fence -- the stamp the host is waiting for
struct HostSnap {
	uint32_t survivors, shadows, device_error, n_live;
	uint32_t pad[11];
	uint32_t seq;
};
struct DevCounters {
	uint32_t primary_ray_cnt;
	uint32_t start_position;
	uint32_t shadow_ray_cnt;
	uint32_t n_live;
	uint32_t reserved0;      // set_wavefront_globals: the survivor count it found (primary_ray_cnt before its reset) ...
	uint32_t device_error;
	uint32_t reserved_t;
	uint32_t reserved1;      // ... and the shadow-ray count: what the host reads of an iteration whose k_scan_words opened the next one (FrameParams::foldNextPrologue)
	unsigned long long budget_remaining;
	unsigned long long total_extend_rays;
	unsigned long long total_shadow_rays;
	unsigned long long total_primary_rays;
	unsigned long long nodes_extend, tris_extend;
	unsigned long long nodes_connect, tris_connect;
	unsigned long long n_survive, n_shadow_visible;
	unsigned long long rays_in_tree_extend, rays_in_tree_connect; // counting build: rays that passed the root box
	// counting build only: where the extend kernel's lanes spend their wave-iterations.
	// [0]/[1] node-test loop: wave iterations / lane iterations; [2]/[3] pop loop; [4]/[5] triangle loop;
	// [6]/[7] refills / lanes refilled; [8..15] lane-state census of the descent trips (TYR_QUAD_STATS builds)
	unsigned long long debug[16];
	// chunk tickets of the persistent traversal kernel, one word per 128 bytes so that the eight words
	// are eight L2 lines (a single word serves only ~88 returning atomics per microsecond)
	uint32_t extend_chunks[kTicketWords * 32];
	uint32_t reserved2[kTicketWords * 32]; // (connect's chunk tickets live in ConnectCounters)
	uint32_t shade_tiles[kTicketWords * 32]; // k_shade: word w hands out tiles w, w + 8, w + 16, ...
	uint32_t seg[2][kClasses][kSegs * kSegStride]; // records in segment w of class c of ray queue q: seg[q][c][w * kSegStride]
	uint32_t shade_blocks_done;               // k_shade: blocks that have finished (the last one folds the segment counters into the totals)
	uint32_t scan_blocks_done;                // k_scan_words: likewise (the last one scans the blocks' totals)
	uint32_t primary_blocks_done;             // k_primary: likewise (the last one is set_wavefront_globals)
	uint32_t scan_live[2];                    // n_live of iteration i at [i & 1], kept by k_shade's last block for a slot scan that runs BEHIND the next iteration's prologue (FrameParams::shadeOpensNext, scanPrevInTrace)
	uint32_t reserved3[27];
	uint32_t primary_done[kTicketWords * 32]; // k_primary: finished blocks b with b % 8 == w, one word per 128 bytes
	uint32_t segSurv[kClasses][kSegs];        // seg[next] as shade left it: the records in front of the primary rays a top-up appends (the sphere pre-pass's share)
	uint32_t reserved4[16];
};
// What connect reads and draws from, apart from the shadow queue.  Two of them, used by alternate iterations:
// inside tyr_render connect(i) runs on a second stream next to primary / extend of iteration i + 1, whose
// set_wavefront_globals resets ITS set and leaves the one connect(i) is working on alone.
struct ConnectCounters {
	uint32_t shadow_cnt;                // = shadow_ray_cnt of the iteration (kernel.cu:416-417), written by shade's last tile
	uint32_t reserved_t;
	uint32_t pad[30];
	uint32_t chunks[kTicketWords * 32]; // (unused since the traversal launches are one kernel: extend_chunks serves them)
	uint32_t seg[kSegs * kSegStride];   // records in segment w of this iteration's shadow queue
};
constexpr uint32_t kErrStackOverflow = 1u;
// (bit 2 was the look-back time-out of rounds 1-2's stable compaction: no longer raised, not reused)
constexpr uint32_t kErrNoProgress = 4u; // -DTYR_GUARD_PASSES builds: a wave of a flat traversal kernel ran out of passes (kMaxPasses)
constexpr uint32_t kErrQueueOverflow = 8u; // a queue segment ran out of room (the records beyond it were dropped); cannot happen with segCap as host/driver.cpp sizes it, kept as a check

struct FrameParams {
	uint32_t W, H, N;
	uint32_t rank, nranks;
	uint32_t localRows;      // H / nranks
	uint32_t localPixels;    // W * localRows
	uint32_t flags;
	uint32_t frame;          // kernel.cu:667
	float camPos[3], camDir[3], camRight[3], camUp[3]; // kernel.cu:699-700, 719
	float focalDistance, lensRadius;
	tyr_sphere spheres[TYR_NUM_SPHERES]; // kernel.cu:123 __constant__ spheres
	SunParams sun;
	DevScene scene;
	RayQ work, next;
	ShadowQ shadow;          // what this iteration's shade writes and its connect reads
	ShadowQ shadowPrev;      // the previous iteration's (k_trace_flat traces its rays beside this iteration's extend rays, while an early shade launch already fills `shadow`)
	float4* blit;            // main.cpp:129-130
	DevCounters* k;
	ConnectCounters* kc;     // this iteration's set
	ConnectCounters* kcPrev; // the previous iteration's set (k_trace_flat: its shadow rays are traced beside this iteration's extend)
	uint32_t* segWork;            // segment counters of `work` / `next`, class 0 (class 1: + kClassWords); the shadow queue's are kc->seg / kcPrev->seg
	uint32_t* segNext;
	uint32_t segCap;              // records one segment has room for
	uint32_t classStride;         // first slot of class 1 in the ray queues' arrays (= kSegs * segCap)
	uint8_t* survFlag;            // [N + 64] shade writes 1 / 0 at the ray's virtual slot; k_scan_words clears what it reads
	VTable vPrev;                 // the scan of the previous iteration's survive bytes (what kKeyIndirect keys are looked up in)
	unsigned long long* vWordOut; // ... and where the scan of this iteration's goes
	uint32_t* vPreOut;
	uint32_t* vBlkOut;
	uint32_t shadeBlocks;         // k_shade: blocks of all of this iteration's shade launches together (the last one to finish finalises)
	uint32_t refillMinIdle;       // persistent traversal: refill a wave once this many lanes are free
	uint32_t minTraversing;       // flat traversal: leave the descent loop below this many descending lanes
	uint32_t ticketChunk;         // queue slots a wave takes per draw from a device-wide ticket
	uint32_t raysPerBlock;        // the counting build's kernels: queue slots owned by one 256-thread block
	uint32_t staticShare;         // sixteenths of the queue handed out as fixed per-block ranges
	uint32_t traceShadow;         // k_trace_flat: the previous iteration's shadow rays ride in this launch (0: a render's first launch; 2: they are ALL of it -- the launch that ends a render)
	uint32_t staticInterleave;    // ... as 64-slot chunks b, b + G, ... (1) or as one contiguous range per block (0)
	uint32_t wideDrain;           // k_trace_flat: finish a wave's last <= 16 rays four lanes to a ray
	uint32_t foldSpheres;         // k_shade: also do the sphere half of extend / connect for the rays it emits (kernel.cu:125-136, 168-172), as k_primary does for its own: no sphere pre-pass follows
	uint32_t retireGhosts;        // k_shade (with foldSpheres, renders that run to their end): a survivor that will hit nothing is finished in place (it still counts as a survivor and keeps its slot in the next iteration's order)
	uint32_t resolveShadows;      // k_shade (with foldSpheres): a shadow ray that a sphere occludes, or that cannot enter the tree, is answered in place and never queued
	uint32_t retireSky;           // k_primary: finish the camera rays that hit nothing (no sphere, not the root box) on the spot instead of queueing them for shade (tyr_render's merged path; the stage API keeps the reference's full queue)
	uint32_t foldNextPrologue;    // k_scan_words: its last block also opens the NEXT iteration (set_wavefront_globals + the hole padding in front of its traversal launch): tyr_render one iteration ahead of the counts, once the budget is spent (no top-up can follow)
	uint32_t prologueDone;        // the traversal launchers: the previous iteration's k_scan_words did that (no k_primary launch, no k_pad_holes)
	uint32_t shadeOpensNext;      // k_shade: its last block opens the NEXT iteration (what foldNextPrologue has k_scan_words do) and keeps this iteration's n_live in scan_live[]: the scan is then left to the next traversal launch (scanPrevInTrace; TYR_TUNE_SCAN_IN_TRACE)
	uint32_t scanSet;             // ... scan_live[scanSet & 1]: the iteration's parity
	HostSnap* hostSnap;           // k_shade: its last block writes what the host's render loop steers by here (pinned host memory) -- no copy and no event between this launch and the next (TYR_TUNE_KERNEL_SNAPSHOT)
	uint32_t snapSeq;             // ... stamped with this number, written last
	uint32_t scanPrevInTrace;     // k_trace_flat: on its way in, its waves do the slot scan of the iteration BEFORE (n = *scanLivePrev, tables = vPrev): that iteration's shade opened this one (shadeOpensNext) and no k_scan_words was launched
	const uint32_t* scanLivePrev;
	uint32_t prevFolded;          // the traversal launchers: the shade launch that made this iteration's survivors and shadow rays did so (only the holes at the segments' ends are left to mark)
	const uint32_t* scanLive;     // k_scan_words: where this iteration's ray count is (&k->n_live)
	// TYR_FLAG_LIGHT_LIST (extension): emissive triangles, as indices into scene.tris in array order
	const uint32_t* lights;
	uint32_t nLights;
	float triEmission[3];
	// TYR_FLAG_TRIANGLE_COLORS (extension): 256 x { colour rgb, emission rgb } as two float4 per entry
	const float4* palette;
};

// launch shape of the traversal kernel (tyr_set_tuning; DESIGN.md section 4.4 has the measurements behind the defaults)
struct Tuning {
	int minTraversing = 32;
	int ticketChunk = 64;
	int staticShare = 12;     // sixteenths of the queue dealt out as fixed (interleaved) per-block chunks before the ticketed rest
	int staticInterleave = 1;
	int wideDrain = 1;        // the last rays of a wave four lanes to a ray
	int stagedNodes = 64;
	int refillMinIdle = 16;
	int wavesPerSimd = 0;     // persistent grid size; 0 = what the occupancy query admits
	int runAhead = 2;         // tyr_render: queue iteration i + 1 before iteration i's counts are on the host: 0 never, 1 / 2 always
	int mergeTrace = 1;       // tyr_render: connect(i) rides in the launch of extend(i + 1)
	int profileMask = 31;     // TYR_FLAG_PROFILE: which stages (bit TYR_K_*) get a hipEvent pair
	int resolveShadows = 1;   // merged path of tyr_render (needs foldSpheres): shade answers the shadow rays that cannot reach a triangle itself
	int retireSky = 1;        // merged path of tyr_render: k_primary finishes the camera rays that hit nothing itself (they never reach a queue)
	int wideBlockMinItems = 3 << 20; // k_trace_flat: launches of at least this many rays run as 768-thread blocks, six waves per SIMD (< 0: never)
	int foldPrologue = 1;     // tyr_render one iteration ahead of the counts: once the budget is spent, an iteration's last kernel opens the next one (set_wavefront_globals, hole padding): two launches and two gaps fewer per iteration
	int layoutOnDevice = 1;   // tyr_scene_upload: the reference's arrays go to the device as they are and hip/bvh_layout_dev.hip makes the records there (the same bytes); 0 = host/bvh_layout.cpp makes them and they are copied
	int scanInTrace = 1;      // tyr_render one iteration ahead, the next iteration known to come without a top-up: no k_scan_words launch -- k_shade's last block opens that iteration and its traversal launch's waves do the slot scan on their way in (hip/scan_wave.hpp)
	int kernelSnapshot = 1;   // tyr_render one iteration ahead: the counts the host waits for are written to pinned host memory by k_shade's last block instead of copied behind it and signalled by an event (two packets in the stream between this iteration's shade and the next traversal launch)
	int foldSpheres = 1;      // merged path of tyr_render: shade does the sphere pre-passes' work for the rays it emits ; 0: k_extend_spheres / k_connect_spheres re-read them
};
constexpr uint32_t kCountRaysPerBlock = 1024; // the counting build's kernels: queue slots owned by one 256-thread block

constexpr int kBlock = 256; // 4 wave64 per workgroup

// Per-context cache of the occupancy queries that size the persistent grids (a slow host call: asked once per
// kernel, not once per launch).  Lives in tyr_ctx -- one ctx per device, no process-wide statics.
enum { kLcShade = 0, kLcTrace, kLcKinds };
struct LaunchCache {
	int perCU[kLcKinds][6] = {};
};

// launches (all on `stream`); grids are sized by the host from upper bounds, kernels bound-check
// against the device counters
void launch_primary(const FrameParams& P, uint32_t maxNew, hipStream_t stream);
void launch_pad_holes(const FrameParams& P, bool workQueue, bool shadowQueue, hipStream_t stream, bool resetTickets = false); // the slots at the segments' ends that hold no record become rays that enter nothing
void launch_scan(const FrameParams& P, uint32_t maxLive, hipStream_t stream); // the survive bytes of this iteration -> vWordOut / vPreOut / vBlkOut
// nSurvivors: upper bound of the slots the sphere pre-pass still has to do (primary rays get theirs in k_primary)
void launch_extend(const FrameParams& P, uint32_t maxLive, uint32_t nSurvivors, bool countVisits, const Tuning& t, int numCUs, LaunchCache& lc, hipStream_t stream);
void launch_shade(const FrameParams& P, uint32_t maxLive, int numCUs, LaunchCache& lc, hipStream_t stream);
void launch_trace_prepasses(const FrameParams& P, uint32_t nSurvivors, uint32_t maxShadowPrev, hipStream_t stream, uint32_t maxLive = 0);
void launch_trace_kernel(const FrameParams& P, uint32_t items, const Tuning& t, int numCUs, LaunchCache& lc, hipStream_t stream); // k_trace_flat alone (launch_trace = pre-passes + this)
void launch_connect(const FrameParams& P, uint32_t maxShadow, bool countVisits, const Tuning& t, int numCUs, LaunchCache& lc, hipStream_t stream);
void launch_trace(const FrameParams& P, uint32_t maxLive, uint32_t nSurvivors, uint32_t maxShadowPrev, const Tuning& t, int numCUs, LaunchCache& lc, hipStream_t stream);
// the sphere pre-passes of extend / connect (frame.hip), launched by the traversal launchers
void launch_extend_spheres(const FrameParams& P, uint32_t nSurvivors, hipStream_t stream, uint32_t maxLive = 0);
void launch_connect_spheres(const FrameParams& P, uint32_t maxShadow, hipStream_t stream);
void launch_resolve(const float4* blit, float4* out, uint32_t nPixels, hipStream_t stream);
void launch_extend_debug(const FrameParams& P, uint32_t maxLive, hipStream_t stream); // TYR_FLAG_DEBUG_BVH
void launch_vecmath_probe(int op, const float* a, const float* b, const float* c, uint32_t n, float* out, hipStream_t stream);
void launch_sunsky_probe(const SunParams& S, int which, const float* dirs, uint32_t n, float* out, hipStream_t stream);

} // namespace tyr
