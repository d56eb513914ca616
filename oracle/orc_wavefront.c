/*
 * oracle/orc_wavefront.c -- serial CPU restatement of the wavefront loop of
 * kernel.cu: primary_rays / set_wavefront_globals / extend / shade / connect /
 * blit_onto_framebuffer and the host driver launch_kernels.
 * TEST INFRASTRUCTURE (see orc.h).  Citations: /root/reference/PathTracer/.
 *
 * The reference's atomicAdd tickets (kernel.cu:251, 333, 350, 416, 444, 555,
 * 586, 607, 632) are restated as SERIAL TICKET ORDER: slot i is processed
 * before slot i+1 and survivors / shadow rays are appended in that order
 * (SURVEY.md section 7 "Determinism vs the reference's atomics").
 *
 * Extensions over the reference (all default to reference behaviour):
 *   - primary-ray budget (orc_set_budget / orc_render): stop topping up after
 *     spp * pixels primaries and drain; then n_live < N in the last iterations;
 *   - pixel sharding (rank, nranks): rank r owns image rows y with y % nranks == r;
 *   - ORC_FLAG_TRIANGLE_MATERIALS: triangles use Triangle::materialType
 *   - ORC_FLAG_LIGHT_LIST (with the former): triangles of materialType LIGHT emit (one emission colour for all of
 *     them, default the reference light's (3,3,3), kernel.cu:680) and are sampled by next-event estimation next to
 *     spheres[6] -- the reference's own TODO "Use light array" (kernel.cu:420, 560).  Defined in orc.h.
 *     instead of the hard-wired DIFF (kernel.cu:380-383).
 */
#include <stdlib.h>

#include "orc_internal.h"

struct orc_ctx {
	uint32_t W, H, N, rank, nranks, flags;
	uint32_t local_pixels;
	orc_node* nodes;
	orc_triangle* prims;
	int nNodes, nPrims;
	/* ORC_FLAG_LIGHT_LIST (extension): triangles whose materialType is LIGHT, in array order */
	uint32_t* lights;
	int nLights;
	float tri_emission[3];
	/* ORC_FLAG_TRIANGLE_COLORS (extension): colour and emission of a triangle = palette entry Triangle::pad_[0] */
	float palette_color[256][3], palette_emission[256][3];
	orc_sphere spheres[ORC_NUM_SPHERES];
	orc_camera camera;
	float sun_position[2];
	int sun_position_changed;
	orc_sunparams sun;
	/* launch_kernels statics, kernel.cu:665-667, 688-691 */
	int first_time;
	uint32_t frame;
	float last_pos[3], last_dir[3];
	float last_focaldistance, last_lensradius;
	/* per-frame camera basis, kernel.cu:699-700 */
	v3 camera_right, camera_up;
	/* queues (main.cpp:119-130) */
	orc_ray* ray_buffer;
	orc_ray* ray_buffer_next;
	orc_shadow* shadow_queue;
	float* blit_buffer; /* float4[W*H] */
	orc_counters k;
};

void orc_default_spheres(orc_sphere s[ORC_NUM_SPHERES]) {
	/* kernel.cu:674-680 */
	const orc_sphere t[ORC_NUM_SPHERES] = {
		{ 16.5f, { 0, 40, 16.5f }, { 1, 1, 1 }, { 0, 0, 0 }, ORC_DIFF },
		{ 16.5f, { 40, 0, 16.5f }, { 0.5f, 0.5f, 0.06f }, { 0, 0, 0 }, ORC_REFR },
		{ 16.5f, { -40, -50, 36.5f }, { 0.6f, 0.5f, 0.4f }, { 0, 0, 0 }, ORC_PHONG },
		{ 16.5f, { -40, -50, 16.5f }, { 0.6f, 0.5f, 0.4f }, { 0, 0, 0 }, ORC_SPEC },
		{ 1e4f, { 0, 0, -1e4f - 20 }, { 1, 1, 1 }, { 0, 0, 0 }, ORC_DIFF },
		{ 20, { 0, -80, 20 }, { 1.0f, 0.0f, 0.0f }, { 0, 0, 0 }, ORC_DIFF },
		{ 9, { 0, -80, 120.0f }, { 0.0f, 1.0f, 0.0f }, { 3, 3, 3 }, ORC_LIGHT },
	};
	memcpy(s, t, sizeof(t));
}

orc_ctx* orc_create(uint32_t W, uint32_t H, uint32_t N, uint32_t rank, uint32_t nranks, uint32_t flags) {
	if (W == 0 || H == 0 || N == 0 || nranks == 0 || rank >= nranks || (H % nranks) != 0)
		return NULL;
	if ((flags & ORC_FLAG_TRIANGLE_COLORS) && !(flags & ORC_FLAG_TRIANGLE_MATERIALS))
		return NULL;
	if ((flags & ORC_FLAG_LIGHT_LIST) && !(flags & ORC_FLAG_TRIANGLE_MATERIALS))
		return NULL;
	orc_ctx* c = (orc_ctx*)calloc(1, sizeof(orc_ctx));
	if (!c)
		return NULL;
	c->W = W;
	c->H = H;
	c->N = N;
	c->rank = rank;
	c->nranks = nranks;
	c->flags = flags;
	c->local_pixels = W * (H / nranks);
	orc_default_spheres(c->spheres);
	/* camera.h:4-9 defaults */
	const orc_camera cam = { { 1, 30, 90 }, { 1, 0, 0 }, { 0, 0, 1 }, 1.0f, 0.0f };
	c->camera = cam;
	c->sun_position[0] = 0.05f; /* variables.cpp:3 */
	c->sun_position[1] = 0.3f;
	c->sun_position_changed = 1; /* variables.cpp:4 */
	c->first_time = 1;
	c->tri_emission[0] = c->tri_emission[1] = c->tri_emission[2] = 3.0f; /* kernel.cu:680 */
	for (int i = 0; i < 256; ++i)
		for (int k = 0; k < 3; ++k) {
			c->palette_color[i][k] = 1.0f;    /* kernel.cu:383: triangles are white */
			c->palette_emission[i][k] = 3.0f; /* kernel.cu:680 */
		}
	c->frame = 1;
	c->last_focaldistance = 1.0f;
	c->last_lensradius = 0.02f;
	c->ray_buffer = (orc_ray*)calloc(N, sizeof(orc_ray));
	c->ray_buffer_next = (orc_ray*)calloc(N, sizeof(orc_ray));
	c->shadow_queue = (orc_shadow*)calloc(N, sizeof(orc_shadow));
	c->blit_buffer = (float*)calloc((size_t)W * H * 4, sizeof(float));
	c->k.budget_remaining = UINT64_MAX;
	c->k.frame = 1;
	if (!c->ray_buffer || !c->ray_buffer_next || !c->shadow_queue || !c->blit_buffer) {
		orc_destroy(c);
		return NULL;
	}
	return c;
}

void orc_destroy(orc_ctx* c) {
	if (!c)
		return;
	free(c->nodes);
	free(c->prims);
	free(c->lights);
	free(c->ray_buffer);
	free(c->ray_buffer_next);
	free(c->shadow_queue);
	free(c->blit_buffer);
	free(c);
}

int orc_scene_upload(orc_ctx* c, const orc_node* nodes, int nNodes, const orc_triangle* prims, int nPrims) {
	free(c->nodes);
	free(c->prims);
	free(c->lights);
	c->nodes = NULL;
	c->prims = NULL;
	c->lights = NULL;
	c->nNodes = 0;
	c->nPrims = 0;
	c->nLights = 0;
	if (nPrims <= 0 || nNodes <= 0)
		return 0; /* Scene.cpp:49-52: empty scene, no BVH */
	c->nodes = (orc_node*)malloc(sizeof(orc_node) * (size_t)nNodes);
	c->prims = (orc_triangle*)malloc(sizeof(orc_triangle) * (size_t)nPrims);
	if (!c->nodes || !c->prims)
		return -1;
	memcpy(c->nodes, nodes, sizeof(orc_node) * (size_t)nNodes);
	memcpy(c->prims, prims, sizeof(orc_triangle) * (size_t)nPrims);
	c->nNodes = nNodes;
	c->nPrims = nPrims;
	free(c->lights);
	c->lights = NULL;
	c->nLights = 0;
	if (c->flags & ORC_FLAG_LIGHT_LIST) {
		c->lights = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)nPrims);
		if (!c->lights)
			return -1;
		for (int i = 0; i < nPrims; ++i)
			if (c->prims[i].materialType == ORC_LIGHT)
				c->lights[c->nLights++] = (uint32_t)i;
	}
	return 0;
}

void orc_set_triangle_emission(orc_ctx* c, const float rgb[3]) { memcpy(c->tri_emission, rgb, 12); }
void orc_set_triangle_palette(orc_ctx* c, const float* color_rgb256, const float* emission_rgb256) {
	memcpy(c->palette_color, color_rgb256, sizeof c->palette_color);
	if (emission_rgb256)
		memcpy(c->palette_emission, emission_rgb256, sizeof c->palette_emission);
}
/* the emission of emissive triangle t: its palette entry with ORC_FLAG_TRIANGLE_COLORS, else the one colour of all */
static v3 triangle_emission(const orc_ctx* c, const orc_triangle* t) {
	return (c->flags & ORC_FLAG_TRIANGLE_COLORS) ? v3load(c->palette_emission[t->pad_[0]]) : v3load(c->tri_emission);
}

void orc_set_spheres(orc_ctx* c, const orc_sphere s[ORC_NUM_SPHERES]) { memcpy(c->spheres, s, sizeof(c->spheres)); }
void orc_set_camera(orc_ctx* c, const orc_camera* cam) { c->camera = *cam; }
void orc_set_sun_position(orc_ctx* c, float x, float y) {
	c->sun_position[0] = x;
	c->sun_position[1] = y;
	c->sun_position_changed = 1;
}
void orc_set_budget(orc_ctx* c, uint64_t primary_rays) { c->k.budget_remaining = primary_rays; }
void orc_get_counters(const orc_ctx* c, orc_counters* out) { *out = c->k; }
const float* orc_blit_buffer(const orc_ctx* c) { return c->blit_buffer; }
const orc_ray* orc_ray_queue(const orc_ctx* c, int which) { return which == 0 ? c->ray_buffer : c->ray_buffer_next; }
const orc_shadow* orc_shadow_queue(const orc_ctx* c) { return c->shadow_queue; }
const orc_sunparams* orc_sun_params(const orc_ctx* c) { return &c->sun; }

void orc_reset_accum(orc_ctx* c) {
	/* kernel.cu:712-718 */
	memset(c->blit_buffer, 0, sizeof(float) * 4 * (size_t)c->W * c->H);
	c->k.primary_ray_cnt = 0;
}

void orc_import_work_queue(orc_ctx* c, const orc_ray* rays, uint32_t n_survivors) {
	if (n_survivors > c->N)
		n_survivors = c->N;
	memcpy(c->ray_buffer, rays, sizeof(orc_ray) * n_survivors);
	c->k.primary_ray_cnt = n_survivors;
}

/* ---- sampling helpers ---------------------------------------------------- */

/* kernel.cu:44-65 */
void orc_random_2d_stratified_sample(uint32_t* seed, float out[2]) {
	const int width2D = 4, height2D = 4;
	const float pixelWidth = 1.0f / width2D, pixelHeight = 1.0f / height2D;
	const int chosenStratum = rng_int_0_max(seed, width2D * height2D);
	const int stratumX = chosenStratum % width2D;
	const int stratumY = (chosenStratum / width2D) % height2D;
	const float stratumXStart = pixelWidth * stratumX;
	const float stratumYStart = pixelHeight * stratumY;
	out[0] = stratumXStart + (rng_float(seed) * pixelWidth);
	out[1] = stratumYStart + (rng_float(seed) * pixelHeight);
}

/* kernel.cu:190-208 */
void orc_concentric_sample_disk(const float u[2], float out[2]) {
	float ox = 2.f * u[0] - 1.0f, oy = 2.f * u[1] - 1.0f;
	if (ox == 0 && oy == 0) {
		out[0] = 0;
		out[1] = 0;
		return;
	}
	float theta, r;
	if (fabsf(ox) > fabsf(oy)) {
		r = ox;
		theta = ORC_PI / 4 * (oy / ox);
	} else {
		r = oy;
		theta = ORC_PI / 2 - ORC_PI / 4 * (ox / oy);
	}
	out[0] = r * dm_cosf(theta);
	out[1] = r * dm_sinf(theta);
}

/* kernel.cu:181-189 */
void orc_orthonormal_basis_naive(const float w_[3], float u_[3], float v_[3]) {
	v3 w = v3load(w_), u;
	if (fabs(w.x) > .9) {
		u = v3make(0.0f, 1.0f, 0.0f);
	} else {
		u = v3make(1.0f, 0.0f, 0.0f);
	}
	u = v3normalize(v3cross(u, w));
	v3store(u_, u);
	v3store(v_, v3cross(w, u));
}

/* ---- host prologue of launch_kernels (kernel.cu:671-718) ------------------ */
void orc_stage_begin(orc_ctx* c) {
	if (c->first_time) {
		c->first_time = 0; /* sphere table and sun_angular are set in orc_create / orc_sun_setup */
	}
	v3 dir = v3load(c->camera.direction), up = v3load(c->camera.up);
	/* kernel.cu:699-700 */
	c->camera_right = v3scale(v3scale(v3normalize(v3cross(dir, up)), 1.5f), (float)c->W / (float)c->H);
	c->camera_up = v3scale(v3normalize(v3cross(c->camera_right, dir)), 1.5f);

	/* kernel.cu:702 */
	int reset_buffer = memcmp(c->last_pos, c->camera.position, 12) != 0 || memcmp(c->last_dir, c->camera.direction, 12) != 0 ||
		c->last_focaldistance != c->camera.focalDistance || c->camera.lensRadius != c->last_lensradius;
	/* vec3 != compares values; memcmp differs only for -0/NaN components, which the ABI rejects */
	if (c->sun_position_changed) { /* kernel.cu:704-710 */
		c->sun_position_changed = 0;
		reset_buffer = 1;
		orc_sun_setup(c->sun_position, &c->sun);
	}
	if (reset_buffer) {
		orc_reset_accum(c);
	}
}

/* ---- primary_rays (kernel.cu:247-297) + set_wavefront_globals (227-244) --- */
void orc_stage_primary(orc_ctx* c) {
	const uint32_t cnt = c->k.primary_ray_cnt;
	uint64_t room = c->N - cnt;
	uint32_t n_new = (uint32_t)(room < c->k.budget_remaining ? room : c->k.budget_remaining);
	const v3 O = v3load(c->camera.position), camera_direction = v3load(c->camera.direction);
	const float focalDistance = c->camera.focalDistance, lens_radius = c->camera.lensRadius;
	const uint32_t frame = c->frame;
	const uint32_t W = c->W, Hl = c->H / c->nranks;

	for (uint32_t index = 0; index < n_new; ++index) {
		const uint32_t ray_index_buffer = index + cnt;
		/* kernel.cu:258 seeds by the ticket `index`.  Sharded (nranks > 1, an extension) every rank would draw the SAME
		 * jitter and lens samples for its index-th ray, i.e. for the pixels (x, yl * R + r), r = 0..R-1: correlated noise in
		 * groups of R rows.  The ranks' tickets are interleaved instead; nranks == 1 is the reference's expression. */
		uint32_t seed = (frame * 147565741u) * 720898027u * (index * c->nranks + c->rank);

		const int x = (int)((c->k.start_position + index) % W);
		const int yl = (int)(((c->k.start_position + index) / W) % Hl);
		const int y = yl * (int)c->nranks + (int)c->rank; /* nranks == 1: y = yl, kernel.cu:264 */

		float sample2D[2];
		orc_random_2d_stratified_sample(&seed, sample2D);
		const float rand_point_pixelX = (float)x - sample2D[0];
		const float rand_point_pixelY = (float)y - sample2D[1];

		const float normalized_i = (rand_point_pixelX / (float)c->W) - 0.5f;
		const float normalized_j = (((float)c->H - rand_point_pixelY) / (float)c->H) - 0.5f;

		v3 directionToFocalPlane = v3add(v3add(camera_direction, v3rscale(normalized_i, c->camera_right)), v3rscale(normalized_j, c->camera_up));
		directionToFocalPlane = v3normalize(directionToFocalPlane);

		const int ImGui_slider_hack = 3;
		v3 convergencePoint = v3add(O, v3rscale(focalDistance * (float)ImGui_slider_hack, directionToFocalPlane));

		float lens_sample[2];
		lens_sample[0] = rng_float(&seed);
		lens_sample[1] = rng_float(&seed);
		float disk[2];
		orc_concentric_sample_disk(lens_sample, disk);
		float pLx = lens_radius * disk[0], pLy = lens_radius * disk[1];
		v3 newOrigin = v3add(v3add(O, v3scale(c->camera_right, pLx)), v3scale(c->camera_up, pLy));
		v3 direction = v3normalize(v3sub(convergencePoint, newOrigin));

		/* kernel.cu:295  { newOrigin, direction, {1,1,1}, 0, 0, 0, y*W+x } + member defaults variables.h:32-33 */
		orc_ray* r = &c->ray_buffer[ray_index_buffer];
		v3store(r->origin, newOrigin);
		v3store(r->direction, direction);
		r->direct[0] = r->direct[1] = r->direct[2] = 1.0f;
		r->distance = 0;
		r->identifier = 0;
		r->bounces = 0;
		r->index = y * (int)c->W + x;
		r->geometry_type = 1;
		r->lastSpecular = 1;
		r->pad_[0] = r->pad_[1] = r->pad_[2] = 0;
	}

	/* set_wavefront_globals, kernel.cu:227-244 (progress = rays generated) */
	c->k.start_position = (uint32_t)(((uint64_t)c->k.start_position + n_new) % c->local_pixels);
	c->k.n_live = cnt + n_new;
	c->k.shadow_ray_cnt = 0;
	c->k.primary_ray_cnt = 0;
	if (c->k.budget_remaining != UINT64_MAX)
		c->k.budget_remaining -= n_new;
	c->k.total_primary_rays += n_new;
	c->k.total_extend_rays += c->k.n_live;
}

/* ---- extend (kernel.cu:331-343) via intersect_scene (125-142) ------------- */
void orc_stage_extend(orc_ctx* c) {
	uint64_t cnt[3] = { 0, 0, 0 };
	for (uint32_t index = 0; index < c->k.n_live; ++index) {
		orc_ray* ray = &c->ray_buffer[index];
		float d;
		ray->distance = ORC_VERY_FAR;
		for (int i = ORC_NUM_SPHERES; i--;) {
			if ((d = orc_sphere_intersect(&c->spheres[i], ray->origin, ray->direction)) && d < ray->distance) {
				ray->distance = d;
				ray->identifier = i;
				ray->geometry_type = 0;
			}
		}
		if (c->nPrims > 0 && orc_bvh_intersect(c->nodes, c->prims, ray, cnt)) {
			ray->geometry_type = 1;
		}
	}
	c->k.nodes_extend += cnt[0];
	c->k.tris_extend += cnt[1];
	c->k.rays_in_tree_extend += cnt[2];
}

/* ---- extend_debug_BVH (kernel.cu:300-328) via intersect_scene_DEBUG (143-160): the reference's compile-time BVH_DEBUG
 * build -- spheres ignored, every ray starts at VERY_FAR, the pixel is coloured by the number of traversal steps
 * (intersect_debug, bvh.h:164-209: loop iterations - 1); plain stores, the last ray of a pixel wins ------------------ */
void orc_stage_extend_debug(orc_ctx* c) {
	for (uint32_t index = 0; index < c->k.n_live; ++index) {
		orc_ray* ray = &c->ray_buffer[index];
		ray->distance = ORC_VERY_FAR;
		int traversals = 0;
		if (c->nPrims > 0) {
			uint64_t cnt[3] = { 0, 0, 0 };
			if (orc_bvh_intersect(c->nodes, c->prims, ray, cnt))
				ray->geometry_type = 1;
			traversals = (int)cnt[0] - 1;
		}
		float* px = &c->blit_buffer[4 * (size_t)ray->index];
		int green = (int)((0.0002f * (float)traversals) * 255.99f);
		green = green > 255 ? 255 : green;
		px[1] = (float)green;
		px[3] = 1.0f;
		if (traversals >= 70) { /* "Color very costly regions distinctly" */
			px[0] = (float)green;
			px[1] = 0.0f;
		}
	}
}

/* NEE toward spheres[6] (kernel.cu:419-448 and 559-591); returns 1 if a shadow ray was produced */
static int sample_sphere_light(const orc_ctx* c, uint32_t* seed, v3 origin, v3 normal, v3* lightDir, float* cosSurfaceToLight, float* cosLightToSurface, v3* lightVector) {
	const orc_sphere* lightsource = &c->spheres[6];
	float cosPhi = 2.0f * rng_float(seed) - 1.0f;
	float sinPhi = sqrtf(1.0f - cosPhi * cosPhi);
	float theta = 2.0f * ORC_PI * rng_float(seed);

	float x = lightsource->position[0] + lightsource->radius * sinPhi * dm_sinf(theta);
	float y = lightsource->position[1] + lightsource->radius * cosPhi;
	float z = lightsource->position[2] + lightsource->radius * sinPhi * dm_cosf(theta);

	v3 p = v3make(x, y, z);
	*lightVector = v3sub(p, origin);
	v3 nL = v3normalize(v3sub(p, v3load(lightsource->position)));
	*lightDir = v3normalize(*lightVector);
	*cosSurfaceToLight = v3dot(normal, *lightDir);
	*cosLightToSurface = v3dot(nL, v3neg(*lightDir));
	return *cosSurfaceToLight > 0 && *cosLightToSurface > 0;
}

/* The emitter a next-event sample goes to (ORC_FLAG_LIGHT_LIST; without it, or without emissive triangles, this is
 * spheres[6] and draws nothing extra, so the reference's random sequence is untouched):
 *   k = RandomIntBetween0AndMax(seed, nLights) (kernel.cu:39-41, range [0, nLights]); k == nLights -> spheres[6];
 *   otherwise triangle lights[k], a point p = vert + e1*b1 + e2*b2 with b1 = sqrt(u1)*(1-u2), b2 = sqrt(u1)*u2 (uniform
 *   over the triangle), emitting from its front side (normal e1 x e2, loader.h:28).
 * Returns 1 if a shadow ray is due; *weight = emission * area-measure factors that replace "emission, 4 pi r^2" of the
 * sphere formulas (kernel.cu:436-442), including the 1/(pick probability) = nLights + 1. */
static int sample_light(const orc_ctx* c, uint32_t* seed, v3 origin, v3 normal, v3* lightDir, float* cosSurfaceToLight, float* cosLightToSurface, v3* lightVector, v3* emission, float* area) {
	const orc_sphere* ls = &c->spheres[6];
	if (!(c->flags & ORC_FLAG_LIGHT_LIST) || c->nLights == 0) {
		*emission = v3load(ls->emmission);
		*area = 4 * ORC_PI * ls->radius * ls->radius;
		return sample_sphere_light(c, seed, origin, normal, lightDir, cosSurfaceToLight, cosLightToSurface, lightVector);
	}
	const int k = rng_int_0_max(seed, c->nLights);
	const float pick = (float)(c->nLights + 1);
	if (k >= c->nLights) {
		*emission = v3scale(v3load(ls->emmission), pick);
		*area = 4 * ORC_PI * ls->radius * ls->radius;
		return sample_sphere_light(c, seed, origin, normal, lightDir, cosSurfaceToLight, cosLightToSurface, lightVector);
	}
	const orc_triangle* t = &c->prims[c->lights[k]];
	const float u1 = rng_float(seed);
	const float u2 = rng_float(seed);
	const float su = sqrtf(u1);
	const float b1 = su * (1.0f - u2);
	const float b2 = su * u2;
	const v3 e1 = v3load(t->e1), e2 = v3load(t->e2);
	const v3 p = v3add(v3add(v3load(t->vert), v3scale(e1, b1)), v3scale(e2, b2));
	const v3 cr = v3cross(e1, e2);
	*lightVector = v3sub(p, origin);
	const v3 nL = v3normalize(cr);
	*lightDir = v3normalize(*lightVector);
	*cosSurfaceToLight = v3dot(normal, *lightDir);
	*cosLightToSurface = v3dot(nL, v3neg(*lightDir));
	*emission = v3scale(triangle_emission(c, t), pick);
	*area = 0.5f * v3length(cr);
	return *cosSurfaceToLight > 0 && *cosLightToSurface > 0;
}

static void push_shadow(orc_ctx* c, v3 origin, v3 dir, v3 color, int index, float closest) {
	orc_shadow* s = &c->shadow_queue[c->k.shadow_ray_cnt++];
	v3store(s->origin, origin);
	v3store(s->direction, dir);
	v3store(s->color, color);
	s->buffer_index = index;
	s->closestDistance = closest;
}

/* ---- shade (kernel.cu:347-627) ------------------------------------------- */
void orc_stage_shade(orc_ctx* c) {
	const uint32_t frame = c->frame;
	const float phongexponent = 40.0f;
	for (uint32_t index = 0; index < c->k.n_live; ++index) {
		int new_frame = 0;
		orc_ray ray = c->ray_buffer[index]; /* local copy; the reference mutates in place, nothing reads it back */
		v3 color = v3make(0.f, 0.f, 0.f);
		v3 object_color = v3make(0.f, 0.f, 0.f);
		uint32_t seed = (frame * (uint32_t)ray.index * 147565741u) * 720898027u * index;
		int reflection_type = ORC_DIFF;

		v3 origin = v3load(ray.origin), direction = v3load(ray.direction), direct = v3load(ray.direct);

		if (ray.distance < ORC_VERY_FAR) {
			origin = v3add(origin, v3scale(direction, ray.distance));

			v3 normal;
			if (ray.geometry_type == 0) {
				const orc_sphere* object = &c->spheres[ray.identifier];
				normal = v3divs(v3sub(origin, v3load(object->position)), object->radius);
				reflection_type = object->refl;
				if (reflection_type != ORC_REFR && reflection_type != ORC_LIGHT) {
					direct = v3mul(direct, v3load(object->color));
				}
				object_color = v3load(object->color);
			} else {
				const orc_triangle* triangle = &c->prims[ray.identifier];
				normal = v3normalize(v3cross(v3load(triangle->e1), v3load(triangle->e2)));
				reflection_type = ORC_DIFF;
				object_color = v3make(1.f, 1.f, 1.f);
				if (c->flags & ORC_FLAG_TRIANGLE_MATERIALS) {
					/* extension: LIGHT on a triangle needs ORC_FLAG_LIGHT_LIST */
					const int highest = (c->flags & ORC_FLAG_LIGHT_LIST) ? ORC_LIGHT : ORC_PHONG;
					reflection_type = triangle->materialType <= highest ? triangle->materialType : ORC_DIFF;
				}
				if (c->flags & ORC_FLAG_TRIANGLE_COLORS) {
					/* the reference's commented-out `tempTriangle.color = mesh.color` (Scene.cpp:44), treated like a sphere's
					 * colour (kernel.cu:375-377) */
					object_color = v3load(c->palette_color[triangle->pad_[0]]);
					if (reflection_type != ORC_REFR && reflection_type != ORC_LIGHT)
						direct = v3mul(direct, object_color);
				}
			}

			int outside = v3dot(normal, direction) < 0;
			normal = outside ? normal : v3scale(normal, -1.f);
			origin = v3add(origin, v3scale(normal, ORC_EPSILON));

			if (reflection_type == ORC_LIGHT) {
				if (ray.lastSpecular) {
					color = v3mul(direct, ray.geometry_type == 0 ? v3load(c->spheres[ray.identifier].emmission) : triangle_emission(c, &c->prims[ray.identifier]));
				} else {
					color = v3make(0.f, 0.f, 0.f);
					direct = v3make(0.f, 0.f, 0.f);
				}
			}
			ray.lastSpecular = 0;
			switch (reflection_type) {
			case ORC_LIGHT:
				break;
			case ORC_DIFF: {
				float sd[3];
				orc_cone_sample(&c->sun, &seed, sd);
				v3 sunSampleDir = v3load(sd);
				float sunLight = v3dot(normal, sunSampleDir);
				if (rng_float(&seed) < 0.5f) {
					if (sunLight > 0.f) {
						float sv[3];
						orc_sun(&c->sun, sd, sv);
						v3 col = v3mul(v3rscale(2.0f, direct), v3scale(v3scale(v3load(sv), sunLight), 1E-5f));
						push_shadow(c, origin, sunSampleDir, col, ray.index, 1e20f);
					}
				} else {
					v3 lightDir, lightVector, emission;
					float cosS, cosL, area;
					if (sample_light(c, &seed, origin, normal, &lightDir, &cosS, &cosL, &lightVector, &emission, &area)) {
						float closestAllowed = v3length(lightVector);
						float solidAngle = (cosL * area) / v3dot(lightVector, lightVector);
						v3 shadowColor = v3scale(v3scale(v3scale(v3mul(v3scale(emission, 2.0f), direct), solidAngle), ORC_INV_PI), cosS);
						push_shadow(c, origin, lightDir, shadowColor, ray.index, closestAllowed);
					}
				}
				if (ray.bounces < ORC_MAX_BOUNCES) {
					float r1 = 2.f * ORC_PI * rng_float(&seed);
					float r2 = rng_float(&seed);
					float r2s = sqrtf(r2);
					float n_[3], u_[3], v_[3];
					v3store(n_, normal);
					orc_orthonormal_basis_naive(n_, u_, v_);
					v3 u = v3load(u_), v = v3load(v_);
					v3 d = v3add(v3add(v3scale(v3scale(u, dm_cosf(r1)), r2s), v3scale(v3scale(v, dm_sinf(r1)), r2s)), v3scale(normal, sqrtf(1 - r2)));
					direction = v3normalize(d);
				}
				break;
			}
			case ORC_SPEC: {
				ray.lastSpecular = 1;
				direction = v3reflect(direction, normal);
				break;
			}
			case ORC_REFR: {
				const float n1 = outside ? 1.2f : 1.0f;
				const float n2 = outside ? 1.0f : 1.2f;
				float fresnel = 0;
				float r0 = (n1 - n2) / (n1 + n2);
				r0 *= r0;
				const float cosI = -v3dot(normal, direction);
				const float n = n2 / n1;
				const float sinT2 = n * n * (1.0f - cosI * cosI);
				if (sinT2 > 1.0f) {
					fresnel = 1.0f;
				} else {
					const float x = 1.0f - cosI;
					fresnel = r0 + (1.0f - r0) * x * x * x * x * x;
				}
				if (rng_float(&seed) < fresnel) {
					ray.lastSpecular = 1;
					direction = v3reflect(direction, normal);
				} else {
					origin = v3sub(origin, v3scale(v3scale(normal, 2.f), ORC_EPSILON));
					const float cosT = sqrtf(1.0f - sinT2);
					direction = v3add(v3rscale(n, direction), v3rscale(n * cosI - cosT, normal));
				}
				if (!outside) {
					v3 a = v3scale(v3neg(object_color), ray.distance);
					direct = v3mul(direct, v3make(dm_expf(a.x), dm_expf(a.y), dm_expf(a.z)));
				}
				break;
			}
			case ORC_PHONG: {
				v3 w, u, v, d;
				do {
					float phi = 2 * ORC_PI * rng_float(&seed);
					float r2 = rng_float(&seed);
					float cosTheta = dm_powf(1.0f - r2, 1.0f / (phongexponent + 1.0f));
					float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
					w = v3sub(direction, v3scale(v3scale(normal, 2.0f), v3dot(normal, direction)));
					w = v3normalize(w);
					float w_[3], u_[3], v_[3];
					v3store(w_, w);
					orc_orthonormal_basis_naive(w_, u_, v_);
					u = v3load(u_);
					v = v3load(v_);
					d = v3add(v3add(v3scale(v3scale(u, dm_cosf(phi)), sinTheta), v3scale(v3scale(v, dm_sinf(phi)), sinTheta)), v3scale(w, cosTheta));
					d = v3normalize(d);
				} while (v3dot(d, normal) <= ORC_EPSILON);

				float sd[3];
				orc_cone_sample(&c->sun, &seed, sd);
				v3 sunSampleDir = v3load(sd);
				float sunLight = v3dot(normal, sunSampleDir);
				if (rng_float(&seed) < 0.5f) {
					if (sunLight > 0.f) {
						float phongCos = v3dot(sunSampleDir, w);
						if (phongCos > ORC_EPSILON) {
							sunLight *= dm_powf(phongCos, phongexponent);
							float sv[3];
							orc_sun(&c->sun, sd, sv);
							v3 col = v3mul(v3scale(v3rscale(2.0f, direct), (phongexponent + 2) * 0.5f * ORC_INV_PI), v3scale(v3scale(v3load(sv), sunLight), 1E-5f));
							push_shadow(c, origin, sunSampleDir, col, ray.index, 1e20f);
						}
					}
				} else {
					v3 lightDir, lightVector, emission;
					float cosS, cosL, area;
					if (sample_light(c, &seed, origin, normal, &lightDir, &cosS, &cosL, &lightVector, &emission, &area)) {
						float phongCos = v3dot(lightDir, w);
						if (phongCos > ORC_EPSILON) {
							phongCos = dm_powf(phongCos, phongexponent);
							float closestAllowed = v3length(lightVector);
							float solidAngle = (cosL * area) / v3dot(lightVector, lightVector);
							v3 sc = v3mul(v3scale(emission, 2.0f), direct);
							sc = v3scale(sc, solidAngle);
							sc = v3scale(sc, (phongexponent + 2));
							sc = v3scale(sc, 0.5f);
							sc = v3scale(sc, ORC_INV_PI);
							sc = v3scale(sc, phongCos);
							sc = v3scale(sc, cosS);
							push_shadow(c, origin, lightDir, sc, ray.index, closestAllowed);
						}
					}
				}
				origin = v3add(origin, v3scale(w, ORC_EPSILON));
				direction = d;
				break;
			}
			}

			/* Russian roulette, kernel.cu:599-611 */
			float p = glm_minf(1.0f, glm_maxf(direct.z, glm_maxf(direct.x, direct.y)));
			if (ray.bounces < ORC_MAX_BOUNCES && p > (0 + ORC_EPSILON) && rng_float(&seed) <= p) {
				ray.bounces++;
				direct = v3scale(direct, 1.0f / p);
				v3store(ray.origin, origin);
				v3store(ray.direction, direction);
				v3store(ray.direct, direct);
				c->ray_buffer_next[c->k.primary_ray_cnt++] = ray;
				c->k.n_survive++;
			} else {
				new_frame++;
			}
		} else {
			float dv[3], sv[3];
			v3store(dv, direction);
			if (!ray.lastSpecular)
				orc_sky(&c->sun, dv, sv);
			else
				orc_sunsky(&c->sun, dv, sv);
			color = v3add(color, v3mul(direct, v3load(sv)));
			new_frame++;
		}

		float* px = &c->blit_buffer[4 * (size_t)ray.index];
		px[0] += color.x;
		px[1] += color.y;
		px[2] += color.z;
		px[3] += (float)new_frame;
	}
	c->k.total_shadow_rays += c->k.shadow_ray_cnt;
}

/* ---- connect (kernel.cu:630-646) via intersect_scene_simple (162-174) ----- */
void orc_stage_connect(orc_ctx* c) {
	uint64_t cnt[3] = { 0, 0, 0 };
	for (uint32_t index = 0; index < c->k.shadow_ray_cnt; ++index) {
		const orc_shadow* ray = &c->shadow_queue[index];
		int occluded = 0;
		if (c->nPrims > 0 && orc_bvh_intersect_simple(c->nodes, c->prims, ray, ray->closestDistance, cnt)) {
			occluded = 1;
		} else {
			float d;
			for (int i = ORC_NUM_SPHERES; i--;) {
				if ((d = orc_sphere_intersect(&c->spheres[i], ray->origin, ray->direction)) && (d + ORC_EPSILON) < ray->closestDistance) {
					occluded = 1;
					break;
				}
			}
		}
		if (!occluded) {
			float* px = &c->blit_buffer[4 * (size_t)ray->buffer_index];
			px[0] += ray->color[0];
			px[1] += ray->color[1];
			px[2] += ray->color[2];
			c->k.n_shadow_visible++;
		}
	}
	c->k.nodes_connect += cnt[0];
	c->k.tris_connect += cnt[1];
	c->k.rays_in_tree_connect += cnt[2];
}

void orc_stage_end(orc_ctx* c) {
	/* kernel.cu:735-745 */
	if (c->frame == UINT32_MAX)
		c->frame = 0;
	c->frame++;
	c->k.frame = c->frame;
	memcpy(c->last_pos, c->camera.position, 12);
	memcpy(c->last_dir, c->camera.direction, 12);
	c->last_focaldistance = c->camera.focalDistance;
	c->last_lensradius = c->camera.lensRadius;
	/* main.cpp:169 std::swap(ray_buffer_work, ray_buffer_next) */
	orc_ray* t = c->ray_buffer;
	c->ray_buffer = c->ray_buffer_next;
	c->ray_buffer_next = t;
}

int orc_launch_kernels(orc_ctx* c) {
	orc_stage_begin(c);
	orc_stage_primary(c);
	if (c->flags & ORC_FLAG_DEBUG_BVH) { /* kernel.cu:721-722: #if BVH_DEBUG */
		orc_stage_extend_debug(c);
		orc_stage_end(c);
		return 0;
	}
	orc_stage_extend(c);
	orc_stage_shade(c);
	orc_stage_connect(c);
	orc_stage_end(c);
	return 0;
}

int orc_render(orc_ctx* c, uint32_t spp, int max_iterations) {
	orc_set_budget(c, (uint64_t)spp * c->local_pixels);
	int it = 0;
	while (it < max_iterations) {
		orc_launch_kernels(c);
		++it;
		if (c->k.budget_remaining == 0 && c->k.primary_ray_cnt == 0)
			break;
	}
	return it;
}

/* blit_onto_framebuffer, kernel.cu:648-662 (linear RGBA32F instead of a GL surface) */
void orc_resolve(const orc_ctx* c, float* out) {
	const size_t n = (size_t)c->W * c->H;
	for (size_t i = 0; i < n; ++i) {
		const float* color = &c->blit_buffer[4 * i];
		float cl[3] = { color[0] / color[3], color[1] / color[3], color[2] / color[3] };
		for (int k = 0; k < 3; ++k) {
			out[4 * i + k] = dm_powf(cl[k] / (cl[k] + 1.f), 1.0f / 2.2f);
		}
		/* cl.a = 1 -> pow(1/(1+1), 1/2.2) */
		out[4 * i + 3] = dm_powf(1.f / (1.f + 1.f), 1.0f / 2.2f);
	}
}


/* ---- Camera::handle_input / update (camera.cpp:3-52): the GLFW reads of the original arrive as a record ------------ */
void orc_camera_handle_input(orc_camera_pose* cam, const orc_input_state* in, double delta) {
	float speed = 1;
	if (in->key_left_shift) /* camera.cpp:5-7 */
		speed = 40;
	v3 position = v3load(cam->position);
	const v3 direction = v3load(cam->direction), up = v3load(cam->up);
	if (in->key_w) /* camera.cpp:9-13: position += direction * speed * float(delta) */
		position = v3add(position, v3scale(v3scale(direction, speed), (float)delta));
	else if (in->key_s)
		position = v3sub(position, v3scale(v3scale(direction, speed), (float)delta));
	const v3 displacement = v3scale(v3scale(v3normalize(v3cross(direction, up)), speed), (float)delta); /* camera.cpp:15 */
	if (in->key_a)
		position = v3sub(position, displacement);
	else if (in->key_d)
		position = v3add(position, displacement);
	if (in->key_space) /* camera.cpp:22-26 */
		position.z += 1 * speed * (float)delta;
	else if (in->key_left_control)
		position.z -= 1 * speed * (float)delta;
	v3store(cam->position, position);
	if (in->key_left_alt) /* camera.cpp:27-29 */
		return;
	const double diffx = in->cursor_x - in->window_w * 0.5; /* camera.cpp:36-37 */
	const double diffy = in->cursor_y - in->window_h * 0.5;
	cam->horizontal_angle += diffx * 0.012;
	cam->vertical_angle -= diffy * 0.012;
	const double lo = -ORC_PI / 2 + 0.001, hi = ORC_PI / 2 - 0.001; /* camera.cpp:41: pi is a float, the sums are doubles */
	double va = cam->vertical_angle < hi ? cam->vertical_angle : hi;
	cam->vertical_angle = lo > va ? lo : va;
}
void orc_camera_update(orc_camera_pose* cam) {
	v3 d = v3make((float)(cos(cam->vertical_angle) * sin(cam->horizontal_angle)), (float)(cos(cam->vertical_angle) * cos(cam->horizontal_angle)), (float)sin(cam->vertical_angle));
	d = v3normalize(d);
	v3store(cam->direction, d);
}
