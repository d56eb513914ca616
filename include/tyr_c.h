/*
 * include/tyr_c.h -- C ABI of libtyrant_hip.so, the MI355X-native drop-in for the
 * wavefront render loop of stijnherfst/Tyrant.
 *
 * Every entry point cites the reference interface it replaces (file:line under
 * PathTracer/).  Plain pointers and sizes only; record layouts are byte-for-byte
 * the reference's structs so a cgo/FFI/C++ caller can pass its own arrays.
 * All functions return 0 on success or a negative tyr_status / positive
 * hipError_t; nothing in the library calls exit() (the reference's cuda() macro
 * does, assert_cuda.cpp:3-14 -- a caller wanting that wraps calls in TYR_CHECK).
 *
 * Threading: one tyr_ctx per device, one HIP stream per ctx, calls on one ctx
 * are not thread-safe (the reference is single-threaded with function statics,
 * kernel.cu:665-667).
 */
#ifndef TYR_C_H
#define TYR_C_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TYR_ABI_VERSION 5 /* 2: tyr_counters grows (rays_in_tree_*, debug[16]), tyr_dist_*, per-triangle colours; 3: retired tuning keys removed, tyr_sunsky_probe / tyr_sun_setup; 4: tyr_set_frame, tyr_layout_probe, tyr_bvh_build_device, tyr_scene_build_upload, tyr_scene_hash, tyr_scene_info grows (upload_*_s, layout_on_device); 5: the streamed tail's tuning keys (16, 17, 18) retired with its kernels */

/* ---- record layouts (identical to the reference structs) ------------------ */

/* loader.h:13-19  struct Triangle  (40 B) */
typedef struct tyr_triangle {
	float vert[3];
	float e1[3];
	float e2[3];
	uint8_t materialType;
	uint8_t pad_[3];
} tyr_triangle;

/* Bbox.h:3-5  struct BBox  (24 B): bounds[0] = bottom, bounds[1] = top */
typedef struct tyr_bbox {
	float bounds[2][3];
} tyr_bbox;

/* bvh.h:55-68  BVH::BVHNode  (32 B), depth-first, left child = index + 1 */
typedef struct tyr_bvh_node {
	tyr_bbox bbox;
	int32_t offset; /* union { primitiveOffset; secondChildOffset; } */
	uint16_t primitiveCount;
	uint8_t splitAxis;
	uint8_t pad;
} tyr_bvh_node;

/* variables.h:24-34  struct RayQueue  (60 B) -- import/export format only; device queues are SoA */
typedef struct tyr_ray_queue {
	float origin[3];
	float direction[3];
	float direct[3];
	float distance;
	int32_t identifier;
	int32_t bounces;
	int32_t index;
	int32_t geometry_type; /* variables.h:20-22: Sphere = 0, Triangle = 1 */
	uint8_t lastSpecular;
	uint8_t pad_[3];
} tyr_ray_queue;

/* variables.h:36-42  struct ShadowQueue  (44 B) */
typedef struct tyr_shadow_queue {
	float origin[3];
	float direction[3];
	float color[3];
	int32_t buffer_index;
	float closestDistance;
} tyr_shadow_queue;

/* kernel.cu:67-81  enum Refl_t, struct Sphere  (44 B) */
enum { TYR_DIFF = 0, TYR_SPEC = 1, TYR_REFR = 2, TYR_PHONG = 3, TYR_LIGHT = 4 };
#define TYR_NUM_SPHERES 7 /* kernel.cu:14 */
typedef struct tyr_sphere {
	float radius;
	float position[3];
	float color[3];
	float emmission[3];
	int32_t refl;
} tyr_sphere;

/* camera.h:3-9  struct Camera: the fields launch_kernels reads (kernel.cu:699-702, 719) */
typedef struct tyr_camera {
	float position[3];
	float direction[3];
	float up[3];
	float focalDistance;
	float lensRadius;
} tyr_camera;

/* ---- status codes ---------------------------------------------------------- */
enum {
	TYR_OK = 0,
	TYR_ERR_INVALID = -1,    /* bad argument (null, zero size, non-finite geometry, ...) */
	TYR_ERR_NO_DEVICE = -2,  /* no HIP device: the product path has no CPU fallback */
	TYR_ERR_NO_SCENE = -3,
	TYR_ERR_NO_BUFFER = -4,  /* no blit_buffer bound */
	TYR_ERR_OOM = -5,
	TYR_ERR_DEVICE = -6,     /* a kernel reported an internal error: tyr_counters.device_error says which */
	TYR_ERR_UNSUPPORTED = -7,
	TYR_ERR_IO = -8          /* a file could not be opened, or a write came up short */
};
const char* tyr_status_string(int status);
int tyr_abi_version(void);

/* ---- context --------------------------------------------------------------- */

/* Replaces the compile-time constants render_width/render_height/ray_queue_buffer_size
 * (variables.h:9-10, 44), argv[1] device pick (main.cpp:91) and sm_cores (variables.h:18). */
typedef struct tyr_config {
	uint32_t width;        /* variables.h:9 */
	uint32_t height;       /* variables.h:10 */
	uint32_t queue_size;   /* variables.h:44 ray_queue_buffer_size */
	int32_t device;        /* main.cpp:91 */
	uint32_t rank;         /* pixel sharding: this ctx owns image rows y with y % nranks == rank */
	uint32_t nranks;       /* 1 = reference behaviour */
	uint32_t flags;        /* TYR_FLAG_* */
	void* stream;          /* hipStream_t to launch on; NULL = the ctx creates its own */
} tyr_config;

#define TYR_FLAG_TRIANGLE_MATERIALS 1u /* shade switch driven by Triangle::materialType (loader.h:16) instead of hard-wired DIFF (kernel.cu:380-383) */
#define TYR_FLAG_PROFILE 2u            /* bracket every kernel with hipEvents on the ctx stream (tyr_get_timings) */
#define TYR_FLAG_COUNT_VISITS 4u       /* counting build of extend/connect: nodes visited / triangles tested (bvh.h:164-209) */
#define TYR_FLAG_LIGHT_LIST 8u         /* with TRIANGLE_MATERIALS: triangles of materialType LIGHT emit and are sampled by next-event
                                        * estimation next to spheres[6] -- the light array kernel.cu:420 / 560 leave as a TODO.  Off = the
                                        * reference's behaviour and random sequence. */

#define TYR_FLAG_TRIANGLE_COLORS 16u   /* with TRIANGLE_MATERIALS: colour and emission per triangle -- the reference's commented-out `tempTriangle.color =
                                        * mesh.color` (Scene.cpp:44).  The 40-byte Triangle record keeps its layout: the first of its three padding bytes
                                        * (offset 37) indexes a 256-entry palette (tyr_set_triangle_palette).  The colour acts like a sphere's
                                        * (kernel.cu:375-377: throughput x colour unless REFR / LIGHT; REFR absorbs with it, kernel.cu:511-513); with
                                        * LIGHT_LIST an emissive triangle emits its own palette emission.  Off = white triangles, one emission. */

#define TYR_FLAG_DEBUG_BVH 32u         /* the reference's compile-time BVH_DEBUG (kernel.cu:721-722): the extend stage is extend_debug_BVH (kernel.cu:300-328,
                                        * intersect_debug bvh.h:164-209) -- spheres ignored, and the blit_buffer receives, per ray, green = steps x 0.0002 x 255.99
                                        * (red instead from 70 steps on), a = 1 -- and tyr_launch_kernels / tyr_render run primary + that stage only */

typedef struct tyr_ctx tyr_ctx;

int tyr_create(tyr_ctx** out, const tyr_config* cfg);
int tyr_destroy(tyr_ctx* ctx);

/* Scene::Load's upload half, Scene.cpp:55-67: copies the flat node array and the
 * (builder-reordered) triangle array to the device.  The device layout behind the
 * ABI is private; the traversal visit order of bvh.h:118-161 is preserved. */
int tyr_scene_upload(tyr_ctx* ctx, const tyr_bvh_node* nodes, int32_t nNodes, const tyr_triangle* prims, int32_t nPrims);

/* kernel.cu:674-681: the __constant__ sphere table (7 entries, light = slot 6). NULL = the reference's table. */
int tyr_set_spheres(tyr_ctx* ctx, const tyr_sphere* spheres);
/* TYR_FLAG_LIGHT_LIST: the emission of every LIGHT triangle, float[3]; default (3,3,3), the reference light's (kernel.cu:680) */
int tyr_set_triangle_emission(tyr_ctx* ctx, const float* rgb);
/* TYR_FLAG_TRIANGLE_COLORS: 256 colours (float[256][3]) and, optionally, 256 emissions (NULL keeps (3,3,3)); entry i serves the
 * triangles whose byte 37 is i.  Defaults: white, (3,3,3). */
int tyr_set_triangle_palette(tyr_ctx* ctx, const float* color_rgb256, const float* emission_rgb256);
/* the global `camera` (camera.h:24) read at kernel.cu:699-702, 719 */
int tyr_set_camera(tyr_ctx* ctx, const tyr_camera* cam);
/* sun_position / sun_position_changed (variables.h:16-17, kernel.cu:704-710) */
int tyr_set_sun_position(tyr_ctx* ctx, float x, float y);
/* blit_buffer: caller-allocated device float4[width*height] (main.cpp:129-130), argument 2 of launch_kernels.
 * NULL = the ctx allocates and owns one. */
int tyr_set_blit_buffer(tyr_ctx* ctx, void* device_float4);
void* tyr_get_blit_buffer(tyr_ctx* ctx);

/* ---- the per-frame entry point --------------------------------------------- */

/* cudaError launch_kernels(array, blit_buffer, gpuScene, queue, queue2, shadowQueue)  interop.h:25, kernel.cu:664-748
 * plus the caller's std::swap of the two ray buffers (main.cpp:169): exactly one wavefront iteration
 * (top-up -> extend -> shade -> connect), frame counter advanced.  Returns after the stream is idle
 * (kernel.cu:733 cudaDeviceSynchronize).  The GL surface write (kernel.cu:731) is tyr_resolve. */
int tyr_launch_kernels(tyr_ctx* ctx);

/* Primary-ray budget (extension; the reference streams forever): after `n` more primaries
 * the queue is no longer topped up.  UINT64_MAX = reference behaviour. */
int tyr_set_budget(tyr_ctx* ctx, uint64_t primary_rays);
/* The frame counter every seed is built from (`static unsigned frame = 1`, kernel.cu:667; read at kernel.cu:258 and 363,
 * advanced once per launch_kernels, kernel.cu:736-739).  The reference can only count on; a host that wants the SAME
 * render again -- a fixed-seed comparison, a benchmark whose steps are one job -- restarts it here (with tyr_reset_accum
 * and a budget that is a multiple of the pixel count the scan-line cursor is back where it was, too).  0 is not a frame
 * the reference ever has (it skips it on wrap): TYR_ERR_INVALID. */
int tyr_set_frame(tyr_ctx* ctx, uint32_t frame);

typedef struct tyr_counters {
	uint32_t primary_ray_cnt; /* kernel.cu:211 survivors written by the last shade */
	uint32_t start_position;  /* kernel.cu:214 */
	uint32_t shadow_ray_cnt;  /* kernel.cu:224 */
	uint32_t n_live;          /* rays in the work queue after top-up (== queue_size in the reference) */
	uint32_t frame;           /* kernel.cu:667 */
	uint32_t device_error;    /* non-zero: TYR_ERR_DEVICE detail bits: 1 traversal stack overflow, 4 (builds with -DTYR_GUARD_PASSES only) a traversal wave gave up after 2^24 passes without finishing, 8 a queue segment ran out of room (the records beyond it were dropped; the segments are sized so that no set of rays can do it -- a check, not an expected outcome) */
	uint64_t budget_remaining;
	uint64_t total_extend_rays; /* sum of n_live over iterations */
	uint64_t total_shadow_rays; /* sum of shadow_ray_cnt over iterations */
	uint64_t total_primary_rays;
	uint64_t nodes_extend, tris_extend;   /* TYR_FLAG_COUNT_VISITS only */
	uint64_t nodes_connect, tris_connect; /* TYR_FLAG_COUNT_VISITS only */
	uint64_t n_survive, n_shadow_visible;
	/* TYR_FLAG_COUNT_VISITS only: rays whose test of the root box passed, i.e. rays that enter the tree (the others cost the
	 * reference one node visit and end there): bench.py's in-tree ray rate */
	uint64_t rays_in_tree_extend, rays_in_tree_connect;
	/* TYR_FLAG_COUNT_VISITS (and the instrumented diagnostics builds) only: SIMD occupancy of the extend kernel's loops, as
	 * (wave iterations, lane iterations) pairs: [0,1] node tests, [2,3] stack pops, [4,5] triangle tests, [6,7] refills /
	 * lanes refilled; lane / (64 * wave) is the fraction of the 64 lanes doing work in that loop.  [8..15]: a census of lane
	 * states per descent trip (tools/loop_occupancy.py). */
	uint64_t debug[16];
} tyr_counters;
int tyr_get_counters(tyr_ctx* ctx, tyr_counters* out);

/* The render loop of main.cpp:139-170 without the window: iterate launch_kernels until
 * spp * (width*height/nranks) primaries were generated and every path finished.
 * iterations_out may be NULL. */
int tyr_render(tyr_ctx* ctx, uint32_t spp, uint32_t max_iterations, uint32_t* iterations_out);
/* max_iterations: UINT32_MAX (exactly) = run to the end.  Only such a render lets shade finish a SURVIVOR in place
 * (TYR_TUNE_RETIRE_SKY's second half: a bounce ray that can hit nothing gets the next iteration's sky term now) -- any other
 * limit, however large, could cut the render between the two iterations and show that pixel one iteration early, so it
 * turns that half off (results are the same numbers either way).  A render that FAILS (non-zero return) leaves the
 * accumulation buffer undefined: contributions of an iteration that was never counted may already be in it; call
 * tyr_reset_accum before rendering again. */

/* blit_onto_framebuffer, kernel.cu:648-662: rgb/a -> c/(c+1) -> gamma 1/2.2, written to a linear
 * RGBA32F device buffer (float4[width*height]) instead of a GL surface. */
int tyr_resolve(tyr_ctx* ctx, void* device_rgba_out);
/* cudaMemset(blit_buffer) + primary_ray_cnt = 0, kernel.cu:712-718 */
int tyr_reset_accum(tyr_ctx* ctx);
/* copy blit_buffer to host float4[width*height] */
int tyr_read_accum(tyr_ctx* ctx, float* host_float4);

/* ---- stage-level entry points (one kernel of kernel.cu each; parity tests) -- */
int tyr_stage_begin(tyr_ctx* ctx);   /* host prologue of launch_kernels, kernel.cu:671-718 */
int tyr_stage_primary(tyr_ctx* ctx); /* primary_rays + set_wavefront_globals, kernel.cu:227-297 */
int tyr_stage_extend(tyr_ctx* ctx);  /* extend, kernel.cu:331-343 */
int tyr_stage_shade(tyr_ctx* ctx);   /* shade, kernel.cu:347-627 */
int tyr_stage_connect(tyr_ctx* ctx); /* connect, kernel.cu:630-646 */
int tyr_stage_end(tyr_ctx* ctx);     /* frame++ (kernel.cu:735-745) and the caller's swap (main.cpp:169) */
int tyr_sync(tyr_ctx* ctx);

/* AoS import/export of the SoA device queues in the reference's record formats.
 * which: 0 = work queue (input of the next extend), 1 = next queue (survivors of the last shade).
 * What a queue HOLDS after tyr_render with its default tuning: only the rays that still have to be traced.  TYR_TUNE_RETIRE_SKY
 * and TYR_TUNE_RESOLVE_SHADOWS finish rays whose fate is known where they are made (camera rays and survivors that can hit
 * nothing, shadow rays that cannot reach a triangle): such rays are counted (primary_ray_cnt, n_live, shadow_ray_cnt, the
 * totals) and never written to a queue, so an export of `count` = one of those counters returns the queued records in the
 * serial order followed by zero-filled records for the ones finished in place.  The stage API (tyr_stage_*,
 * tyr_launch_kernels) and renders with those two knobs at 0 queue every ray, as the reference does. */
int tyr_queue_export(tyr_ctx* ctx, int which, tyr_ray_queue* host, uint32_t count);
/* Test hook: every record of the queue (`which` as tyr_queue_export) is looked up in the DEVICE's rank tables -- the scan of the
 * survive bytes that k_shade resolves virtual slots with -- and the slot found is compared with the record's place in the
 * order tyr_queue_export presents (its host-side sort by key).  *checked_out = records, *mismatches_out = records whose
 * table rank differs from that place.  Call it between tyr_stage_shade and tyr_stage_end (which = 1) or after
 * tyr_stage_primary (which = 0). */
int tyr_queue_rank_check(tyr_ctx* ctx, int which, uint32_t* checked_out, uint32_t* mismatches_out);
int tyr_queue_import(tyr_ctx* ctx, const tyr_ray_queue* host, uint32_t n_survivors);
int tyr_shadow_export(tyr_ctx* ctx, tyr_shadow_queue* host, uint32_t count);
/* overwrite the shadow queue with `n` records (the input of the next tyr_stage_connect; kernel-level parity tests) */
int tyr_shadow_import(tyr_ctx* ctx, const tyr_shadow_queue* host, uint32_t n);

/* What tyr_scene_upload made of the tree: sizes of the private device layout against its encoding limits (a quad-node
 * reference has 25 index bits, a leaf reference 26 offset bits; bvh.h:124's 64-entry stack is checked at run time and
 * reported through tyr_counters.device_error). */
typedef struct tyr_scene_info {
	uint32_t n_prims, n_pair_nodes, n_quad_nodes, n_staged_nodes, n_lights; /* n_pair_nodes: 0 unless the ctx has TYR_FLAG_COUNT_VISITS or TYR_FLAG_DEBUG_BVH (only those traverse pair nodes; others neither lay them out nor upload them) */
	uint32_t max_quad_nodes;   /* 1 << 25 */
	uint32_t max_prim_offset;  /* 1 << 26 */
	uint32_t quad_max_stack;   /* the most stack entries any traversal of this tree can need (the drain's four-lanes-to-a-ray form holds 48 and is used only below that) */
	uint64_t device_bytes;     /* quad nodes + pair nodes + 48-byte triangles resident in HBM */
	double upload_layout_s;    /* the last tyr_scene_upload / tyr_scene_build_upload: seconds in the layout passes (on the device, or on the threads of tyr_set_build_threads) ... */
	double upload_copy_s;      /* ... and in device allocation + the copies to HBM */
	uint32_t layout_on_device; /* 1: that layout ran on the device (TYR_TUNE_LAYOUT_ON_DEVICE; hip/bvh_layout_dev.hip), 0: on the host */
	uint32_t reserved_;
} tyr_scene_info;
int tyr_get_scene_info(tyr_ctx* ctx, tyr_scene_info* out);

/* Test / measurement hook, no device needed: the host half of tyr_scene_upload -- the reference's flat node array re-laid out
 * as 128-byte quad nodes (and, when want_pairs != 0, the counting build's 64-byte pair nodes) plus 48-byte triangles, on the
 * threads of tyr_set_build_threads -- without the copy to HBM.  Reports sizes, FNV-1a 64-bit hashes of the three arrays (the
 * layout is byte-identical whatever the thread count: tests/test_host_and_abi.py) and the seconds it took. */
typedef struct tyr_layout_stats {
	uint32_t n_pair_nodes, n_quad_nodes, n_staged_nodes, quad_max_stack, root_ref, quad_root_ref;
	uint64_t hash_pairs, hash_quads, hash_tris;
	double seconds;
} tyr_layout_stats;
int tyr_layout_probe(const tyr_bvh_node* nodes, int32_t nNodes, const tyr_triangle* prims, int32_t nPrims, int32_t want_pairs, tyr_layout_stats* out);
/* The same figures for the scene a ctx HOLDS: the arrays are read back from HBM and hashed like tyr_layout_probe's (pair nodes:
 * only a ctx that has them).  Equal hashes = the device holds the bytes the host pass makes, wherever the layout ran. */
int tyr_scene_hash(tyr_ctx* ctx, tyr_layout_stats* out);

/* The vector functions every kernel is built from (hip/vecmath.hpp: glm's dot, cross, normalize, length, reflect, min,
 * max, clamp, mix, smoothstep and the vec3 operators in glm's evaluation order -- Dependencies/glm-0.9.9.3/glm/detail/
 * func_geometric.inl:14-116, func_common.inl:16-29, 103-111, 257-265, 566 -- plus the deterministic pow / exp), run ON
 * THE DEVICE over n float3 triples of host arrays a, b, c into host array out (float3 each): the hook that pins the
 * device arithmetic to the vendored glm's own answers (tests/golden/ref_glm.npz).  op codes: oracle/ref_harness.cpp. */
int tyr_vecmath_probe(int32_t device, int32_t op, const float* a, const float* b, const float* c, uint32_t n, float* out);

/* The atmosphere as the shade kernel evaluates it (hip/sunsky.hpp, hip/device_common.hpp cone_sample), run ON THE DEVICE
 * over host arrays, for the sun position (sun_x, sun_y) (variables.cpp:3; the constants of kernel.cu:683-709 come from
 * the library's own host setup): the hook that pins the device arithmetic to the answers of the reference's own
 * sunsky.cu compiled as C++ (tests/golden/ref_sunsky.npz, oracle/ref_host_harness.cpp).
 *   which 0 sun(dir) sunsky.cu:32-74 | 1 sky(dir) 76-114 | 2 sunsky(dir) 116-161: dirs and out are n float3
 *   which 3 getConeSample(sunDirection, 1 - sunAngularDiameterCos, seed) sunsky.cu:170-185 as kernel.cu:410 calls it:
 *           n successive samples along ONE xorshift stream; dirs[0] carries the seed's bits in, out[3n] the state after */
int tyr_sunsky_probe(int32_t device, float sun_x, float sun_y, int32_t which, const float* dirs, uint32_t n, float* out);
/* the host half of the above (no GPU needed): the 25 floats of the per-sun-change constants -- sunDirection[3],
 * sunAngularDiameterCos, sunE, rayleighAtX[3], mieAtX[3], totalLightAtX[3], mixFactor, coneDir[3], coneO1[3], coneO2[3],
 * coneExtent (kernel.cu:683-684, 704-709; sunsky.cu:15-26, 66-67, 172-175) */
int tyr_sun_setup(float sun_x, float sun_y, float* out25);

/* ---- measurement ------------------------------------------------------------ */
enum { TYR_K_PRIMARY = 0, TYR_K_EXTEND = 1, TYR_K_SHADE = 2, TYR_K_CONNECT = 3, TYR_K_RESOLVE = 4, TYR_K_COUNT = 5 };
typedef struct tyr_timings {
	double ms[TYR_K_COUNT];        /* summed hipEvent time per kernel since the last reset */
	uint64_t launches[TYR_K_COUNT];
} tyr_timings;
int tyr_get_timings(tyr_ctx* ctx, tyr_timings* out, int reset);

/* Launch-shape knobs of the traversal kernel (the reference's equivalents are the literals `sm_cores * 8, 128` at
 * kernel.cu:719-726).  They never change results; DESIGN.md section 4.4 has the measurements behind the defaults.
 * (ABI 3 retired TRAVERSAL_VARIANT, STACK_LDS_DEPTH, RAYS_PER_BLOCK, MIN_LEAVES and OVERLAP_CONNECT with the kernels
 * they selected; the remaining keys keep their numbers.) */
enum {
	TYR_TUNE_REFILL_MIN_IDLE = 1,    /* refill a wave when at least this many of its 64 lanes are free (1..64, default 16) */
	TYR_TUNE_WAVES_PER_SIMD = 2,     /* resident 256-thread blocks per CU of the persistent grid; 0 (default) = the occupancy query's answer */
	TYR_TUNE_MIN_TRAVERSING = 4,     /* leave the descent loop below this many descending lanes when leaves / refills are pending (1..64, default 32) */
	TYR_TUNE_TICKET_CHUNK = 5,       /* queue slots a wave reserves per draw from a ticket (64..65536, default 64) */
	TYR_TUNE_STATIC_SHARE = 8,       /* sixteenths of the queue dealt to the blocks as fixed chunks before the ticketed rest (0..15, default 12) */
	TYR_TUNE_STAGED_NODES = 9,       /* top-of-tree quad nodes each block keeps in LDS (0..64, default 64) */
	TYR_TUNE_PROFILE_MASK = 11,      /* with TYR_FLAG_PROFILE: bit TYR_K_* set = that stage gets a hipEvent pair (default 31 = all five; a pair costs ~10 us of idle GPU) */
	TYR_TUNE_MERGE_TRACE = 12,       /* tyr_render: 1 (default) = connect(i) shares the launch of extend(i + 1); 0 = launch_kernels' order, iteration by iteration */
	TYR_TUNE_STATIC_INTERLEAVE = 13, /* the fixed per-block part as interleaved 64-slot chunks (1, default) or one contiguous range per block (0) */
	TYR_TUNE_RUN_AHEAD = 14,         /* tyr_render: queue iteration i + 1 before iteration i's counts reach the host: 0 never, 1 or 2 (default) always -- nothing is queued behind the iteration that is known to end the render (budget spent, kMaxBounces iterations since the last top-up) */
	TYR_TUNE_WIDE_DRAIN = 15,        /* 1 (default) = a wave's last <= 16 rays are finished four lanes to a ray */
	/* 16, 17, 18: STREAM_TAIL, STREAM_SHADE_PER_CU, STREAM_TRACE_PER_CU -- retired in ABI 5 with the streamed tail (round 4: one traversal kernel across a render's last iterations,
	   bit-exact, 30 % slower; branch experiment/stream-tail, docs/HISTORY.md section 4.8); tyr_set_tuning answers TYR_ERR_INVALID for them */
	TYR_TUNE_RETIRE_SKY = 20,        /* merged path of tyr_render: 1 (default) = a camera ray that hits no sphere and misses the tree's root box is finished by k_primary itself -- its pixel gets sunsky(direction) (kernel.cu:613-617 for a fresh ray; no random number is involved) and it never enters a queue; 0 = shade does it an iteration later */
	TYR_TUNE_RESOLVE_SHADOWS = 21,   /* merged path of tyr_render, with TYR_TUNE_FOLD_SPHERES: 1 (default) = a shadow ray that a sphere occludes or that fails the tree's root box for its bound is answered by shade itself (visible: its colour joins the pixel's contribution; kernel.cu:630-646 reduced to what is known) and never queued; it still counts as emitted / visible */
	TYR_TUNE_WIDE_BLOCK_MIN_ITEMS = 22, /* k_trace_flat: a launch of at least this many rays (extend + carried shadow rays) runs as 768-thread blocks -- two per CU, six waves per SIMD, one copy of the staged nodes per three 256-thread parts -- instead of 256-thread blocks at five waves per SIMD; default 3 Mi (the sixth wave feeds a fat launch faster and lengthens the drain of a thin one); -1: never */
	TYR_TUNE_FOLD_PROLOGUE = 23,     /* tyr_render, one iteration ahead of the counts (TYR_TUNE_RUN_AHEAD) with TYR_TUNE_FOLD_SPHERES: 1 (default) = once the primary budget is spent, the kernel that ends an iteration (the slot scan) also opens the next one -- set_wavefront_globals (kernel.cu:227-244) and the padding of the queue segments' ends -- instead of a one-block launch each in front of the traversal kernel; 0 = those launches */
	TYR_TUNE_LAYOUT_ON_DEVICE = 24,  /* tyr_scene_upload: 1 (default) = the reference's node and triangle arrays are copied to the device as they are (32 + 40 bytes per node / triangle) and the layout pass runs there (hip/bvh_layout_dev.hip: the same bytes as the host pass; trees it leaves to the host -- pair nodes wanted, leaves of more than 31 primitives, a tree that is one leaf, malformed input -- take the host pass); 0 = always the host pass + a copy of the finished records (128 + 48 bytes) */
	TYR_TUNE_SCAN_IN_TRACE = 25,     /* tyr_render one iteration ahead of the counts (TYR_TUNE_RUN_AHEAD), with TYR_TUNE_FOLD_PROLOGUE: 1 (default) = once the primary budget is spent, an iteration's slot scan -- whose tables only the NEXT shade launch reads -- is not a launch of its own: the shade launch's last block opens the next iteration (set_wavefront_globals, kernel.cu:227-244, and the padding of the queue segments' ends) and that iteration's traversal launch does the scan on its way in, a wave per 16384 slots; 0 = a k_scan_words launch in front of the traversal launch, opening the iteration itself */
	TYR_TUNE_KERNEL_SNAPSHOT = 26,   /* tyr_render one iteration ahead of the counts (TYR_TUNE_RUN_AHEAD): 1 (default) = the counts the host's loop waits for (survivors, shadow rays, the error word) are written to pinned host memory by the shade launch's last block and the host polls their stamp -- no copy and no event in the stream between an iteration's shade launch and the next traversal launch; 0 = a copy of the counters behind the shade launch + an event */
	TYR_TUNE_FOLD_SPHERES = 19       /* merged path of tyr_render: 1 (default) = shade does the sphere pre-passes' work (kernel.cu:127-136, 168-172) for the rays it emits, while they are in registers; 0 = the pre-pass kernels re-read them */
};
int tyr_set_tuning(tyr_ctx* ctx, int key, int value);

/* ---- multi-GPU: the frame's rows are dealt y % nranks == rank (tyr_config.rank / nranks) --------------------
 * The reference is single-GPU (main.cpp:94 computes `multi_gpu` and never uses it); BASELINE.json's north_star makes
 * the 8 GPUs of one node the target: one ctx (and one host thread or process) per device, every rank renders its rows
 * with the unmodified loop into a zero-initialised full-frame blit_buffer, and ONE exchange over RCCL (xGMI) ends the
 * render -- no collective inside the wavefront loop.  librccl is opened at run time (dlopen) by tyr_dist_create. */
#define TYR_DIST_ID_BYTES 128 /* sizeof(ncclUniqueId) */
typedef struct tyr_dist tyr_dist;
enum {
	TYR_DIST_GATHER = 0, /* every rank ships only the rows it owns (1/nranks of the frame, ncclSend -> ncclRecv on `root`, point to
	                      * point = one xGMI link per peer); packed into a double-buffered staging area first, so the blit_buffer is
	                      * free for the next tyr_reset_accum at once and the exchange leaves the critical path */
	TYR_DIST_REDUCE = 1  /* ncclReduce(sum) of the full zero-padded accumulation buffers onto `root`: the form north_star names
	                      * (every rank ships the whole frame; staged through one copy, so the blit_buffer is free at once too) */
};
/* ncclGetUniqueId: one rank calls it and hands the 128 bytes to the others (MPI, a socket, a file, torch.distributed ...) */
int tyr_dist_unique_id(void* id_out128);
/* ncclCommInitRank(nranks, id, rank) on ctx's device; (rank, nranks) must be the ctx's tyr_config pair.  Collective: every
 * rank calls it.  nranks == 1 is allowed (a one-rank communicator: the self-test of the RCCL path on a single GPU). */
int tyr_dist_create(tyr_dist** out, tyr_ctx* ctx, const void* id128, int32_t rank, int32_t nranks);
int tyr_dist_destroy(tyr_dist* d);
/* After tyr_render: combine the ranks' parts of the frame on `root` (collective, asynchronous on the communicator's own
 * stream).  frame_out: device float4[width*height], read on `root` only (may be NULL elsewhere); afterwards it holds
 * the complete accumulation buffer -- rgb sums and path counts of every pixel, i.e. what a one-GPU blit_buffer holds.
 * tyr_dist_wait blocks the host until the last combine has finished. */
int tyr_dist_combine(tyr_dist* d, int32_t mode, int32_t root, void* frame_out_device);
int tyr_dist_wait(tyr_dist* d);
/* what RCCL itself says about the communicator: ncclCommCount (-1 when this librccl has no such entry point) and this rank */
int tyr_dist_info(tyr_dist* d, int32_t* comm_ranks_out, int32_t* rank_out);
/* Row ownership, pure host arithmetic (no device, no communicator): rank r of nranks owns rows r, r + nranks, ...;
 * *n_rows = height / nranks; returns TYR_ERR_INVALID when height % nranks != 0 or rank >= nranks.
 * tyr_dist_row_owner: the rank that owns row y, and that row's index in the owner's packed slab. */
int tyr_dist_owned_rows(uint32_t height, uint32_t rank, uint32_t nranks, uint32_t* first_row, uint32_t* n_rows);
int tyr_dist_row_owner(uint32_t y, uint32_t nranks, uint32_t* rank_out, uint32_t* local_row_out);
/* The two copies tyr_dist_combine puts around the exchange, for a host that moves the slabs itself (all GPUs in one
 * process with hipMemcpyPeer, its own communicator, ...): pack the rows `rank` owns out of a full-frame buffer into a
 * contiguous slab (float4[width * height / nranks]); scatter nranks slabs, stored one after the other, into the rows of
 * a full frame.  Device pointers; asynchronous on `stream` (a hipStream_t; NULL = the default stream). */
int tyr_dist_pack_rows(const void* frame_device, void* slab_device, uint32_t width, uint32_t height, uint32_t rank, uint32_t nranks, void* stream);
int tyr_dist_scatter_rows(const void* slabs_device, void* frame_device, uint32_t width, uint32_t height, uint32_t nranks, void* stream);

/* ---- host side of the hot path ---------------------------------------------- */

/* class BVH, bvh.h:49-108 / bvh.cpp:3-225: binned-SAH build emitting the flat depth-first
 * node array; reorders `prims` in place (bvh.cpp:24).  bboxes: one per primitive (Scene.cpp:29-33).
 * nodes_out must hold 2*n-1 nodes.  algo: 1 = EqualCounts, 2 = SAH (bvh.h:45-47).
 * SAH (what Scene.cpp:53 uses) is byte-pinned: nodes and reordered primitives equal the output of the reference's own
 * bvh.cpp on every fixture scene incl. the 1.1 M-node C3 and 12.6 M-node C5 trees (tests/test_ref_pins.py).
 * EqualCounts is NOT byte-pinned and cannot be: bvh.cpp:113-120 partitions with std::nth_element, whose permutation is
 * implementation-defined (MSVC's STL in the original, libstdc++ in the reference as compiled here); this library's
 * deterministic selection yields a valid median split with its own, thread-count-independent, bytes.
 * Returns the node count (>= 0) or a negative status. */
int tyr_bvh_build(tyr_triangle* prims, int32_t n, const tyr_bbox* bboxes, tyr_bvh_node* nodes_out, int32_t algo);
/* The same build -- SAH, the same bytes: every node and the primitive order of the reference's bvh.cpp -- on the DEVICE
 * (SURVEY.md 8f-1's other alternative; hip/bvh_build_dev.hip): the top of the tree level by level with every primitive in
 * flight (bounds and the 14 buckets by parallel min / max, the order-dependent SAH arithmetic by one thread per range in the
 * reference's order, std::partition's permutation from two prefix sums), subtrees of at most 32 primitives by one thread each
 * running the reference's recursion.  prims / bboxes / nodes_out are HOST arrays as for tyr_bvh_build (prims reordered in
 * place); seconds_out2 (may be NULL): [0] the device's work, [1] the copies in and out.  Returns the node count or a negative
 * status (TYR_ERR_UNSUPPORTED: a degenerate range overflowed a task thread's stack -- use tyr_bvh_build). */
int tyr_bvh_build_device(int32_t device, tyr_triangle* prims, int32_t n, const tyr_bbox* bboxes, tyr_bvh_node* nodes_out, double* seconds_out2);
/* Scene::Load's two halves in ONE call (Scene.cpp:53 the build, :55-67 the upload): the tree is built on the ctx's device
 * (tyr_bvh_build_device's kernels) and laid out there (TYR_TUNE_LAYOUT_ON_DEVICE's kernels) -- the nodes never leave HBM unless
 * asked for.  prims is reordered in place (bvh.cpp:24); nodes_out (may be NULL; else 2n - 1 entries) receives the reference's
 * node array, *n_nodes_out (may be NULL) its length; seconds_out3 (may be NULL): [0] the build, [1] the layout, [2] the copies
 * (triangles and boxes in, triangles -- and nodes -- out).  Whatever the device halves leave to the host (see both) is done
 * there: the scene the ctx ends up with is tyr_bvh_build + tyr_scene_upload's, byte for byte (tyr_scene_hash). */
int tyr_scene_build_upload(tyr_ctx* ctx, tyr_triangle* prims, int32_t n, const tyr_bbox* bboxes, tyr_bvh_node* nodes_out, int32_t* n_nodes_out, double* seconds_out3);
/* Threads tyr_bvh_build may use: the top of the tree fans out into tasks, the output is byte-identical
 * to the serial build.  0 = automatic (env TYR_BUILD_THREADS, else min(16, cores)); 1 = the reference's serial behaviour. */
int tyr_set_build_threads(int32_t threads);
/* Scene.cpp:22-33: per-face bounding boxes */
int tyr_triangle_bboxes(const tyr_triangle* prims, int32_t n, tyr_bbox* out);
/* The import half of Scene::Load (Scene.cpp:3-47, static_mesh.cpp:3-32) for PLY files (ASCII or binary
 * little-endian): first mesh, polygons triangulated as fans (aiProcess_Triangulate), one
 * Triangle{vert, e1, e2} per face; the y/z exchange of static_mesh.cpp:17 and its undo at Scene.cpp:10 cancel.
 * Returns the triangle count (>= 0) or a negative status; *prims_out is malloc'ed, release it with tyr_free. */
int tyr_load_ply(const char* path, tyr_triangle** prims_out);
void tyr_free(void* p);
/* Export of the frame tyr_resolve produced (host copy, float4 per pixel), in place of the GL blit (interop.cpp:50):
 * 8-bit binary PPM of the tonemapped colours, or a float PFM. */
int tyr_write_ppm(const char* path, const float* rgba, uint32_t width, uint32_t height);
int tyr_write_pfm(const char* path, const float* rgba, uint32_t width, uint32_t height);
/* the same 8-bit frame as tyr_write_ppm in a PNG container (stored deflate blocks, no zlib needed) */
int tyr_write_png(const char* path, const float* rgba, uint32_t width, uint32_t height);
/* Camera::update, camera.cpp:46-52 */
int tyr_camera_update(double horizontal_angle, double vertical_angle, float direction_out[3]);

/* Camera::handle_input (camera.cpp:3-44) as a pure function: the reference reads a GLFW window (glfwGetKey,
 * glfwGetCursorPos, glfwGetWindowSize) and re-centres the cursor (glfwSetCursorPos); a headless node has no window, so the
 * state it would have read comes in as a record and the camera record is updated in place -- same arithmetic, same order:
 * W/S along `direction`, A/D along normalize(cross(direction, up)), SPACE / LEFT_CONTROL along z, LEFT_SHIFT = 40 x the
 * speed, all times float(delta); LEFT_ALT skips the mouse look; otherwise the angles move by 0.012 per pixel of cursor
 * offset from the window centre and the vertical one is clamped to +-(pi/2 - 0.001).  Camera::update (tyr_camera_update)
 * turns the angles into `direction` afterwards, as main.cpp:164-166 calls them. */
typedef struct tyr_input_state {
	uint8_t key_w, key_s, key_a, key_d, key_space, key_left_control, key_left_shift, key_left_alt; /* glfwGetKey(...) != 0 */
	double cursor_x, cursor_y;  /* glfwGetCursorPos */
	int32_t window_w, window_h; /* glfwGetWindowSize */
} tyr_input_state;
typedef struct tyr_camera_pose {
	float position[3], direction[3], up[3]; /* camera.h:4-6 */
	double horizontal_angle, vertical_angle; /* camera.h:17-18 */
} tyr_camera_pose;
int tyr_camera_handle_input(tyr_camera_pose* camera, const tyr_input_state* input, double delta);
/* the reference's hard-wired sphere table, kernel.cu:674-680 */
int tyr_default_spheres(tyr_sphere* out7);

#ifdef __cplusplus
}
#endif
#endif
