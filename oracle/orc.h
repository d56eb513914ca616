/*
 * oracle/orc.h -- public C interface of the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE ONLY.  This directory holds a serial CPU restatement of
 * the wavefront path-tracing hot path of stijnherfst/Tyrant.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product (tyrant_amd/, include/) never includes, links or calls anything here.
 *
 * Pinning: see oracle/README.md.  The traversal/intersection functions are
 * checked against the reference's own headers compiled in oracle/_ref
 * (bvh.h, Bbox.h, loader.h); RNG / sun-sky / struct layouts against the known
 * answers recorded in SURVEY.md section 8c (tests/golden/).
 *
 * All citations are file:line under /root/reference/PathTracer/.
 */
#ifndef ORC_H
#define ORC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- data contracts (byte layouts of the reference structs) ------------- */

/* loader.h:13-19  struct Triangle {vert,e1,e2,materialType}  40 B */
typedef struct {
	float vert[3];
	float e1[3];
	float e2[3];
	uint8_t materialType;
	uint8_t pad_[3];
} orc_triangle;

/* Bbox.h:3-5  struct BBox { glm::vec3 bounds[2]; }  24 B */
typedef struct {
	float bounds[2][3];
} orc_bbox;

/* bvh.h:55-68  BVH::BVHNode  32 B */
typedef struct {
	orc_bbox bbox;
	int32_t offset; /* union primitiveOffset / secondChildOffset */
	uint16_t primitiveCount;
	uint8_t splitAxis;
	uint8_t pad;
} orc_node;

/* variables.h:24-34  struct RayQueue  60 B */
typedef struct {
	float origin[3];
	float direction[3];
	float direct[3];
	float distance;
	int32_t identifier;
	int32_t bounces;
	int32_t index;
	int32_t geometry_type; /* variables.h:20-22  Sphere = 0, Triangle = 1 */
	uint8_t lastSpecular;
	uint8_t pad_[3];
} orc_ray;

/* variables.h:36-42  struct ShadowQueue  44 B */
typedef struct {
	float origin[3];
	float direction[3];
	float color[3];
	int32_t buffer_index;
	float closestDistance;
} orc_shadow;

/* kernel.cu:67-81  enum Refl_t + struct Sphere  44 B */
enum { ORC_DIFF = 0, ORC_SPEC = 1, ORC_REFR = 2, ORC_PHONG = 3, ORC_LIGHT = 4 };
typedef struct {
	float radius;
	float position[3];
	float color[3];
	float emmission[3];
	int32_t refl;
} orc_sphere;

/* camera.h:3-9 (the fields the render path reads) */
typedef struct {
	float position[3];
	float direction[3];
	float up[3];
	float focalDistance;
	float lensRadius;
} orc_camera;

/* sunsky.cu:4-8 device globals + every per-frame constant derived from them */
typedef struct {
	float sunDirection[3];
	float sunAngularDiameterCos;
	float sunE;          /* SunIntensity(dot(sunDirection, up))  sunsky.cu:24-26 */
	float rayleighAtX[3];
	float mieAtX[3];     /* totalMie(...) * mieCoefficient        sunsky.cu:15-19,44 */
	float totalLightAtX[3];
	float mixFactor;     /* clamp(pow(1 - dot(up,sunDirection),5),0,1)  sunsky.cu:66 */
	float coneDir[3];    /* getConeSample basis                    sunsky.cu:173-175 */
	float coneO1[3];
	float coneO2[3];
	float coneExtent;    /* 1 - sunAngularDiameterCos              kernel.cu:410 */
} orc_sunparams;

#define ORC_NUM_SPHERES 7      /* kernel.cu:14 */
#define ORC_VERY_FAR 1e20f     /* kernel.cu:15 */
#define ORC_MAX_BOUNCES 5      /* kernel.cu:16 */
#define ORC_EPSILON 0.001f     /* variables.h:14 */

/* flags */
#define ORC_FLAG_TRIANGLE_MATERIALS 1u /* extension: shade switch driven by Triangle::materialType */
#define ORC_FLAG_TRIANGLE_COLORS 16u     /* extension (with TRIANGLE_MATERIALS): colour / emission per triangle from a 256-entry palette indexed by Triangle::pad_[0]; same value as TYR_FLAG_TRIANGLE_COLORS */
#define ORC_FLAG_DEBUG_BVH 32u           /* the reference's BVH_DEBUG build (kernel.cu:721-722): launch_kernels = primary + extend_debug_BVH, the traversal-cost picture; same value as TYR_FLAG_DEBUG_BVH */
#define ORC_FLAG_LIGHT_LIST 8u          /* extension (with the former): LIGHT triangles emit and are sampled by NEE; same value as TYR_FLAG_LIGHT_LIST */

/* ---- a1-a3: RNG and sampling helpers (kernel.cu:23-65, 181-208) ---------- */
uint32_t orc_random_int(uint32_t* seed);
float orc_random_float(uint32_t* seed);
float orc_random_float2(uint32_t* seed);
int orc_random_int_between_0_and_max(uint32_t* seed, int max);
void orc_random_2d_stratified_sample(uint32_t* seed, float out[2]);
void orc_concentric_sample_disk(const float u[2], float out[2]);
void orc_orthonormal_basis_naive(const float w[3], float u[3], float v[3]);

/* ---- deterministic transcendental layer (the numeric spec) ---- */
float orc_dm_sinf(float x);
float orc_dm_cosf(float x);
float orc_dm_expf(float x);
float orc_dm_powf(float x, float y);

/* ---- f4, headless: Camera::handle_input / Camera::update (camera.cpp:3-52) as pure functions ---------------------- */
typedef struct {
	uint8_t key_w, key_s, key_a, key_d, key_space, key_left_control, key_left_shift, key_left_alt;
	double cursor_x, cursor_y;
	int32_t window_w, window_h;
} orc_input_state;
typedef struct {
	float position[3], direction[3], up[3];
	double horizontal_angle, vertical_angle;
} orc_camera_pose;
void orc_camera_handle_input(orc_camera_pose* cam, const orc_input_state* in, double delta); /* camera.cpp:3-44 */
void orc_camera_update(orc_camera_pose* cam);                                                 /* camera.cpp:46-52 */

/* ---- a13/a14: sun & sky (sunsky.cu) --------------------------------------- */
void orc_sun_setup(const float sun_position[2], orc_sunparams* out); /* kernel.cu:683-709 */
void orc_sun(const orc_sunparams* S, const float viewDir[3], float out[3]);
void orc_sky(const orc_sunparams* S, const float viewDir[3], float out[3]);
void orc_sunsky(const orc_sunparams* S, const float viewDir[3], float out[3]);
void orc_cone_sample(const orc_sunparams* S, uint32_t* seed, float out[3]);

/* ---- a17/a18: host BVH builder (bvh.cpp:3-225) --------------------------- */
/* prims is reordered in place (bvh.cpp:24).  nodes_out must hold 2n-1 nodes.
 * algo: 1 = EqualCounts, 2 = SAH (bvh.h:45-47).  Returns nNodes, <0 on error. */
int orc_bvh_build(orc_triangle* prims, int n, const orc_bbox* bboxes, orc_node* nodes_out, int algo);
void orc_triangle_bbox(const orc_triangle* t, orc_bbox* out); /* Scene.cpp:29-33 */

/* ---- a8-a11: traversal (bvh.h:118-256, Bbox.h:38-62, loader.h:21-46) ------ */
float orc_triangle_intersect(const orc_triangle* t, const float origin[3], const float direction[3]);
int orc_bbox_intersect(const orc_bbox* b, const float origin[3], const float invDir[3], const int dirIsNeg[3], float lowest);
/* closest hit; updates ray->distance / ray->identifier; returns hit flag.
 * counters (may be NULL, else 3 entries): [0] += nodes visited, [1] += triangle tests, [2] += 1 when the root box passed. */
int orc_bvh_intersect(const orc_node* nodes, const orc_triangle* prims, orc_ray* ray, uint64_t* counters);
void orc_bvh_intersect_batch(const orc_node* nodes, const orc_triangle* prims, orc_ray* rays, int n, int* hit_out);
int orc_bvh_intersect_simple(const orc_node* nodes, const orc_triangle* prims, const orc_shadow* ray, float closestAllowed, uint64_t* counters);
float orc_sphere_intersect(const orc_sphere* s, const float origin[3], const float direction[3]); /* kernel.cu:83-93 */

void orc_bbox_host_ops(const float* vertices, int n, orc_bbox* bbox_out, float* out2); /* Bbox.h:8-36 */
/* the v3 helpers of orc_internal.h (glm's evaluation order) over arrays of float3: op codes of oracle/ref_harness.cpp ref_glm */
int orc_glm(int op, const float* a, const float* b, const float* c, int n, float* out);

/* ---- a5-a7, a12, a15, a16: the wavefront loop ----------------------------- */
typedef struct orc_ctx orc_ctx;

typedef struct {
	uint32_t primary_ray_cnt; /* kernel.cu:211 */
	uint32_t start_position;  /* kernel.cu:214 */
	uint32_t shadow_ray_cnt;  /* kernel.cu:224 */
	uint32_t n_live;          /* rays in the work queue after top-up (== N in the reference) */
	uint32_t frame;           /* kernel.cu:667 */
	uint32_t pad_;
	uint64_t budget_remaining;  /* primary rays still to generate (UINT64_MAX = reference behaviour) */
	uint64_t total_extend_rays; /* sum of n_live over iterations */
	uint64_t total_shadow_rays; /* sum of shadow_ray_cnt over iterations */
	uint64_t total_primary_rays;
	uint64_t nodes_extend, tris_extend;   /* reference-order visit counts (bvh.h:164-209 counting rule) */
	uint64_t nodes_connect, tris_connect;
	uint64_t n_survive, n_shadow_visible;
	uint64_t rays_in_tree_extend, rays_in_tree_connect; /* rays whose test of the root box passed (the first bvh.h:127 / 222 test) */
} orc_counters;

orc_ctx* orc_create(uint32_t width, uint32_t height, uint32_t queue_size, uint32_t rank, uint32_t nranks, uint32_t flags);
void orc_destroy(orc_ctx* c);
/* nodes/prims are copied (Scene.cpp:55-67) */
int orc_scene_upload(orc_ctx* c, const orc_node* nodes, int nNodes, const orc_triangle* prims, int nPrims);
void orc_set_spheres(orc_ctx* c, const orc_sphere spheres[ORC_NUM_SPHERES]);
void orc_set_triangle_palette(orc_ctx* c, const float* color_rgb256, const float* emission_rgb256); /* ORC_FLAG_TRIANGLE_COLORS; emission may be NULL */
void orc_set_triangle_emission(orc_ctx* c, const float rgb[3]); /* ORC_FLAG_LIGHT_LIST; default (3,3,3) like kernel.cu:680 */
void orc_default_spheres(orc_sphere out[ORC_NUM_SPHERES]); /* kernel.cu:674-680 */
void orc_set_camera(orc_ctx* c, const orc_camera* cam);
void orc_set_sun_position(orc_ctx* c, float x, float y);
void orc_set_budget(orc_ctx* c, uint64_t primary_rays);
/* one wavefront iteration == one launch_kernels call + the caller's swap (kernel.cu:664-748, main.cpp:168-169) */
int orc_launch_kernels(orc_ctx* c);
/* iterate until spp * local_pixels primaries were generated and the queue drained; returns iterations */
int orc_render(orc_ctx* c, uint32_t spp, int max_iterations);
void orc_reset_accum(orc_ctx* c);
void orc_get_counters(const orc_ctx* c, orc_counters* out);
const float* orc_blit_buffer(const orc_ctx* c);           /* float4[W*H]  main.cpp:129-130 */
void orc_resolve(const orc_ctx* c, float* out_rgba);       /* kernel.cu:648-662 */
const orc_ray* orc_ray_queue(const orc_ctx* c, int which); /* 0 = work (input of next iteration), 1 = other */
const orc_shadow* orc_shadow_queue(const orc_ctx* c);
const orc_sunparams* orc_sun_params(const orc_ctx* c);

/* stage-level entry points (each is one kernel of kernel.cu run serially) */
void orc_stage_begin(orc_ctx* c);   /* host prologue of launch_kernels: constants, camera basis, reset (kernel.cu:671-718) */
void orc_stage_primary(orc_ctx* c); /* kernel.cu:247-297 + set_wavefront_globals 227-244 */
void orc_stage_extend(orc_ctx* c);  /* kernel.cu:331-343 */
void orc_stage_extend_debug(orc_ctx* c); /* kernel.cu:300-328 (ORC_FLAG_DEBUG_BVH) */
void orc_stage_shade(orc_ctx* c);   /* kernel.cu:347-627 */
void orc_stage_connect(orc_ctx* c); /* kernel.cu:630-646 */
void orc_stage_end(orc_ctx* c);     /* frame++ and the caller's std::swap (kernel.cu:736-739, main.cpp:169) */
/* overwrite the work queue (for kernel-level parity tests) */
void orc_import_work_queue(orc_ctx* c, const orc_ray* rays, uint32_t n_survivors);

#ifdef __cplusplus
}
#endif
#endif
