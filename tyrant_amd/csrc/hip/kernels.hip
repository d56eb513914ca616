// kernels.hip -- the wavefront path-tracing kernels for gfx950 (MI355X, wave64).
//
// One file = one code object: primary_rays, set_wavefront_globals, extend, shade,
// connect, blit (reference: kernel.cu:227-662).  Differences of STRUCTURE, not of
// arithmetic (DESIGN.md has the full account):
//   - no persistent-thread `while(true){atomicAdd(&raynr_x,1)}` loops (kernel.cu:250-255,
//     332-337, 349-354, 631-636): slot = global thread id, so the four hot atomic
//     addresses disappear and every queue access is coalesced SoA;
//   - survivors and shadow rays are appended by a STABLE device-wide compaction
//     (wave ballot + mbcnt, LDS across the four waves, decoupled look-back across
//     workgroups) instead of `atomicAdd(&primary_ray_cnt,1)` (kernel.cu:607): slot order
//     equals the serial ticket order, which makes a fixed-seed render reproducible
//     (the RNG seed depends on the slot, kernel.cu:363) and keeps pixels sorted inside
//     each generation of rays;
//   - traversal uses 64-byte child-pair nodes (hip/traverse.hpp);
//   - zero contributions are not added to the pixel (kernel.cu:621 TODO), which is
//     value-identical.
#include <hip/hip_runtime.h>

#include "detmath.hpp"
#include "kernels.hpp"
#include "sunsky.hpp"
#include "traverse.hpp"
#include "vecmath.hpp"

namespace tyr {

// ---- RNG, kernel.cu:23-41 ----------------------------------------------------------------
__device__ __forceinline__ uint32_t rng_int(uint32_t& s) {
	s ^= s << 13;
	s ^= s >> 17;
	s ^= s << 5;
	return s;
}
__device__ __forceinline__ float rng_float(uint32_t& s) { return (float)rng_int(s) * 2.3283064365387e-10f; }
__device__ __forceinline__ float rng_float2(uint32_t& s) { return (float)(rng_int(s) >> 16) / 65535.0f; }
__device__ __forceinline__ int rng_int_0_max(uint32_t& s, int max) { return (int)(rng_float(s) * ((float)max + 0.99999f)); }

// kernel.cu:44-65 (chosenStratum is 0..16: stratum 16 aliases (0,0))
__device__ __forceinline__ void stratified_sample(uint32_t& s, float& sx, float& sy) {
	constexpr int width2D = 4, height2D = 4;
	constexpr float pixelWidth = 1.0f / width2D, pixelHeight = 1.0f / height2D;
	const int chosenStratum = rng_int_0_max(s, width2D * height2D);
	const int stratumX = chosenStratum % width2D;
	const int stratumY = (chosenStratum / width2D) % height2D;
	const float stratumXStart = pixelWidth * stratumX;
	const float stratumYStart = pixelHeight * stratumY;
	sx = stratumXStart + (rng_float(s) * pixelWidth);
	sy = stratumYStart + (rng_float(s) * pixelHeight);
}

// kernel.cu:190-208
__device__ __forceinline__ void concentric_sample_disk(float ux, float uy, float& dx, float& dy) {
	const float ox = 2.f * ux - 1.0f, oy = 2.f * uy - 1.0f;
	if (ox == 0 && oy == 0) {
		dx = 0;
		dy = 0;
		return;
	}
	float theta, r;
	if (fabsf(ox) > fabsf(oy)) {
		r = ox;
		theta = kPi / 4 * (oy / ox);
	} else {
		r = oy;
		theta = kPi / 2 - kPi / 4 * (ox / oy);
	}
	float s, c;
	dm::sincosf_det(theta, s, c);
	dx = r * c;
	dy = r * s;
}

// kernel.cu:181-189
__device__ __forceinline__ void orthonormal_basis_naive(f3 w, f3& u, f3& v) {
	if ((double)fabsf(w.x) > .9)
		u = mk3(0.0f, 1.0f, 0.0f);
	else
		u = mk3(1.0f, 0.0f, 0.0f);
	u = normalize(cross(u, w));
	v = cross(w, u);
}

// sunsky.cu:170-185 with the basis precomputed per sun change
__device__ __forceinline__ f3 cone_sample(const SunParams& S, uint32_t& seed) {
	float rx = rng_float2(seed);
	float ry = rng_float2(seed);
	rx = rx * 2.f * kPi;
	ry = 1.0f - ry * S.coneExtent;
	const float oneminus = sqrtf(1.0f - ry * ry);
	float s, c;
	dm::sincosf_det(rx, s, c);
	return (c * oneminus) * ld3(S.coneO1) + (s * oneminus) * ld3(S.coneO2) + ry * ld3(S.coneDir);
}

// kernel.cu:83-93 / 95-105
__device__ __forceinline__ float sphere_intersect(const tyr_sphere& sp, f3 origin, f3 direction) {
	const f3 op = ld3(sp.position) - origin;
	float t;
	const float b = dot(op, direction);
	float disc = b * b - dot(op, op) + sp.radius * sp.radius;
	if (disc < 0)
		return 0;
	disc = sqrtf(disc);
	return (t = b - disc) > kEpsilon ? t : ((t = b + disc) > kEpsilon ? t : 0);
}

// traversal-stack storage of one thread: LDS column + private arrays, bound to a TravStack<STACK_LDS>
#define TYR_DECLARE_STACK(st)                                                        \
	__shared__ uint2 smem_[STACK_LDS ? STACK_LDS * kBlock : 1];                      \
	uint32_t spillRef_[kStackSize - STACK_LDS];                                      \
	float spillT_[kStackSize - STACK_LDS];                                           \
	TravStack<STACK_LDS> st;                                                         \
	st.bind(smem_ + threadIdx.x, spillRef_, spillT_);                                \
	st.reset();

// the flat kernels' stack (hip/traverse.hpp LdsStack), same storage
#define TYR_DECLARE_FLAT_STACK(st, WITH_T)                                           \
	__shared__ typename LdsStack<STACK_LDS, WITH_T>::entry_t smem_[STACK_LDS ? STACK_LDS * kBlock : 1]; \
	uint32_t spillRef_[kStackSize - STACK_LDS];                                      \
	float spillT_[(WITH_T) ? kStackSize - STACK_LDS : 1];                            \
	LdsStack<STACK_LDS, WITH_T> st;                                                  \
	st.bind(smem_ + threadIdx.x, spillRef_, spillT_);                                \
	st.reset();

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// wave-aggregated 64-bit counter add (one atomic per wave)
__device__ __forceinline__ void wave_add_u64(unsigned long long* p, uint32_t v) {
	unsigned long long sum = v;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
		sum += __shfl_xor(sum, o, 64);
	if (lane_id() == 0 && sum)
		atomicAdd(p, sum);
}

// ======================================================================================
// primary_rays, kernel.cu:247-297.  One thread per new queue slot.
// ======================================================================================
// the sphere half of intersect_scene (kernel.cu:127-136): closest of the seven spheres, or VERY_FAR
__device__ __forceinline__ float2 sphere_hit_record(const FrameParams& P, f3 o, f3 d) {
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
	return make_float2(dist, __uint_as_float(id));
}

__global__ void __launch_bounds__(kBlock) k_primary(const FrameParams P) {
	const uint32_t index = blockIdx.x * kBlock + threadIdx.x;
	const uint32_t cnt = P.k->primary_ray_cnt; // survivors already in the buffer (kernel.cu:253)
	const unsigned long long room = (unsigned long long)(P.N - cnt);
	const unsigned long long budget = P.k->budget_remaining;
	const uint32_t nNew = (uint32_t)(room < budget ? room : budget);
	if (index >= nNew)
		return;
	const uint32_t slot = index + cnt;
	uint32_t seed = (P.frame * 147565741u) * 720898027u * index; // kernel.cu:258

	const uint32_t start = P.k->start_position;
	const int x = (int)((start + index) % P.W);
	const int yl = (int)(((start + index) / P.W) % P.localRows);
	const int y = yl * (int)P.nranks + (int)P.rank; // nranks == 1: kernel.cu:264

	float sx, sy;
	stratified_sample(seed, sx, sy);
	const float rand_point_pixelX = (float)x - sx; // kernel.cu:268-269 (jitter is subtracted)
	const float rand_point_pixelY = (float)y - sy;
	const float normalized_i = (rand_point_pixelX / (float)P.W) - 0.5f;
	const float normalized_j = (((float)P.H - rand_point_pixelY) / (float)P.H) - 0.5f;

	const f3 O = ld3(P.camPos), camera_direction = ld3(P.camDir), camera_right = ld3(P.camRight), camera_up = ld3(P.camUp);
	f3 directionToFocalPlane = camera_direction + normalized_i * camera_right + normalized_j * camera_up;
	directionToFocalPlane = normalize(directionToFocalPlane);
	const int ImGui_slider_hack = 3; // kernel.cu:286
	const f3 convergencePoint = O + (P.focalDistance * (float)ImGui_slider_hack) * directionToFocalPlane;

	const float l0 = rng_float(seed);
	const float l1 = rng_float(seed);
	float dx, dy;
	concentric_sample_disk(l0, l1, dx, dy);
	const float pLx = P.lensRadius * dx, pLy = P.lensRadius * dy;
	const f3 newOrigin = O + camera_right * pLx + camera_up * pLy;
	const f3 direction = normalize(convergencePoint - newOrigin);

	// kernel.cu:295: {origin, direction, {1,1,1}, 0, 0, 0, pixel}; lastSpecular defaults to true (variables.h:33)
	P.work.o_dx[slot] = make_float4(newOrigin.x, newOrigin.y, newOrigin.z, direction.x);
	P.work.dyz[slot] = make_float2(direction.y, direction.z);
	P.work.direct_ix[slot] = make_float4(1.0f, 1.0f, 1.0f, __int_as_float(y * (int)P.W + x));
	P.work.flags[slot] = 0u | (1u << 8);
	// extend's sphere pre-pass for this ray, while it is in registers (k_extend_spheres then only has the
	// survivors of the last iteration to do: nothing at all in a render's first, largest wavefront)
	P.work.hit[slot] = sphere_hit_record(P, newOrigin, direction);
}

// ======================================================================================
// set_wavefront_globals, kernel.cu:227-244 (+ reset of the compaction descriptors)
// ======================================================================================
__global__ void __launch_bounds__(kBlock) k_globals(const FrameParams P, uint32_t nDesc) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i < nDesc)
		P.scanDesc[i] = 0ull;
	if (i < kTicketWords) {
		P.k->extend_chunks[i * 32] = 0;
		P.kc->chunks[i * 32] = 0;
		P.k->shade_tiles[i * 32] = 0;
	}
	if (i == 0) {
		DevCounters* k = P.k;
		const uint32_t cnt = k->primary_ray_cnt;
		const unsigned long long room = (unsigned long long)(P.N - cnt);
		const unsigned long long budget = k->budget_remaining;
		const uint32_t nNew = (uint32_t)(room < budget ? room : budget);
		k->start_position = (uint32_t)(((unsigned long long)k->start_position + nNew) % P.localPixels);
		k->n_live = cnt + nNew;
		k->first_fresh = cnt;
		k->shadow_ray_cnt = 0;
		k->primary_ray_cnt = 0;
		k->extend_ticket = 0;
		P.kc->ticket = 0;
		P.kc->shadow_cnt = 0;
		if (budget != ~0ull)
			k->budget_remaining = budget - nNew;
		k->total_primary_rays += nNew;
		k->total_extend_rays += cnt + nNew;
	}
}

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
// shade, kernel.cu:347-627
// ======================================================================================
struct ShadeOut {
	bool survive, shadow;
	f3 origin, direction, direct; // survivor state
	uint32_t flags;
	f3 sOrigin, sDir, sColor;      // shadow ray
	float sClosest;
	f3 color;                      // kernel.cu:622-625: contribution to the pixel, added at the end of the kernel
	int newFrame;
};

// NEE toward spheres[6], kernel.cu:419-447 / 559-590 (common part)
struct LightSample {
	f3 lightDir, lightVector;
	float cosSurfaceToLight, cosLightToSurface;
	bool valid;
};
__device__ __forceinline__ LightSample sample_sphere_light(const tyr_sphere& ls, uint32_t& seed, f3 origin, f3 normal) {
	LightSample L;
	const float cosPhi = 2.0f * rng_float(seed) - 1.0f;
	const float sinPhi = sqrtf(1.0f - cosPhi * cosPhi);
	const float theta = 2.0f * kPi * rng_float(seed);
	float st, ct;
	dm::sincosf_det(theta, st, ct);
	const float x = ls.position[0] + ls.radius * sinPhi * st;
	const float y = ls.position[1] + ls.radius * cosPhi;
	const float z = ls.position[2] + ls.radius * sinPhi * ct;
	const f3 p = mk3(x, y, z);
	L.lightVector = p - origin;
	const f3 nL = normalize(p - ld3(ls.position));
	L.lightDir = normalize(L.lightVector);
	L.cosSurfaceToLight = dot(normal, L.lightDir);
	L.cosLightToSurface = dot(nL, -L.lightDir);
	L.valid = L.cosSurfaceToLight > 0 && L.cosLightToSurface > 0;
	return L;
}

// The emitter a next-event sample goes to.  LIGHTS = TYR_FLAG_LIGHT_LIST (extension, SURVEY.md 8f-3: the reference's
// "TODO Use light array", kernel.cu:420 / 560): with emissive triangles in the scene one of nLights + 1 emitters is
// picked uniformly -- k == nLights is spheres[6], otherwise triangle lights[k], sampled uniformly over its area and
// emitting from its front side (e1 x e2, loader.h:28).  `emission` carries the 1/(pick probability), `area` is what
// the solid-angle term of kernel.cu:438-440 multiplies.  Without LIGHTS (a separate instantiation of the shade
// kernel), or with no emissive triangle, this is the sphere sample and draws nothing extra.
struct EmitterSample {
	LightSample L;
	f3 emission;
	float area;
};
template <bool LIGHTS>
__device__ __forceinline__ EmitterSample sample_emitter(const FrameParams& P, uint32_t& seed, f3 origin, f3 normal) {
	const tyr_sphere& ls = P.spheres[6]; // kernel.cu:421, 561
	EmitterSample E;
	float pick = 1.0f;
	if (LIGHTS && P.nLights != 0) {
		const int k = rng_int_0_max(seed, (int)P.nLights);
		pick = (float)(P.nLights + 1u);
		if (k < (int)P.nLights) {
			const uint32_t id = P.lights[k];
			const float4 t0 = P.scene.tris[3 * id + 0];
			const float4 t1 = P.scene.tris[3 * id + 1];
			const float4 t2 = P.scene.tris[3 * id + 2];
			const float u1 = rng_float(seed);
			const float u2 = rng_float(seed);
			const float su = sqrtf(u1);
			const float b1 = su * (1.0f - u2);
			const float b2 = su * u2;
			const f3 e1 = mk3(t0.w, t1.x, t1.y), e2 = mk3(t1.z, t1.w, t2.x);
			const f3 p = (mk3(t0.x, t0.y, t0.z) + e1 * b1) + e2 * b2;
			const f3 cr = cross(e1, e2);
			E.L.lightVector = p - origin;
			const f3 nL = normalize(cr);
			E.L.lightDir = normalize(E.L.lightVector);
			E.L.cosSurfaceToLight = dot(normal, E.L.lightDir);
			E.L.cosLightToSurface = dot(nL, -E.L.lightDir);
			E.L.valid = E.L.cosSurfaceToLight > 0 && E.L.cosLightToSurface > 0;
			E.emission = mk3(P.triEmission[0], P.triEmission[1], P.triEmission[2]) * pick;
			E.area = 0.5f * length(cr);
			return E;
		}
	}
	E.L = sample_sphere_light(ls, seed, origin, normal);
	E.emission = (LIGHTS && P.nLights != 0) ? ld3(ls.emmission) * pick : ld3(ls.emmission);
	E.area = 4 * kPi * ls.radius * ls.radius;
	return E;
}

// `afterLoads` runs once, for every lane, at the point where this ray's last vector load has been consumed and
// only arithmetic follows: the place to issue memory traffic nobody waits for (k_shade: the pixel atomics of the tile
// before).
// Lanes past the end of the queue come along with valid = false (they load nothing and produce nothing) so that
// afterLoads is reached by the whole wave.
template <bool LIGHTS, class AfterLoads>
__device__ __forceinline__ void shade_ray(const FrameParams& P, uint32_t slot, bool valid, ShadeOut& out, AfterLoads&& afterLoads) {
	float4 a = make_float4(0.f, 0.f, 0.f, 0.f), dq = a;
	float2 b = make_float2(0.f, 0.f), h = make_float2(kVeryFar, 0.f);
	uint32_t fl = 0;
	if (valid) {
		a = P.work.o_dx[slot];
		b = P.work.dyz[slot];
		h = P.work.hit[slot];
		dq = P.work.direct_ix[slot];
		fl = P.work.flags[slot];
	}

	f3 origin = mk3(a.x, a.y, a.z), direction = mk3(a.w, b.x, b.y), direct = mk3(dq.x, dq.y, dq.z);
	const int pixel = __float_as_int(dq.w);
	const float distance = h.x;
	const uint32_t ident = __float_as_uint(h.y);
	int bounces = (int)(fl & 0xffu);
	bool lastSpecular = ((fl >> 8) & 1u) != 0;

	int new_frame = 0;
	f3 color = mk3(0.f, 0.f, 0.f);
	f3 object_color = mk3(0.f, 0.f, 0.f);
	uint32_t seed = (P.frame * (uint32_t)pixel * 147565741u) * 720898027u * slot; // kernel.cu:363
	int reflection_type = TYR_DIFF;
	out.survive = false;
	out.shadow = false;

	enum { kAtmoNone = 0, kAtmoSun, kAtmoSky, kAtmoSunSky };
	int atmo = kAtmoNone;   // what this ray wants from the atmosphere model, evaluated once for the whole wave below
	float atmoScale = 0.0f;
	const bool hit = valid && distance < kVeryFar;
	f3 normal = mk3(0.f, 0.f, 0.f);
	if (hit) {
		origin = origin + direction * distance;
		if (ident & kHitSphere) {
			const tyr_sphere& object = P.spheres[ident & 7u];
			normal = (origin - ld3(object.position)) / object.radius;
			reflection_type = object.refl;
			if (reflection_type != TYR_REFR && reflection_type != TYR_LIGHT)
				direct = direct * ld3(object.color);
			object_color = ld3(object.color);
		} else {
			// kernel.cu:380-383: normal from e1 x e2, white DIFF
			const float4 t0 = P.scene.tris[3 * ident + 0];
			const float4 t1 = P.scene.tris[3 * ident + 1];
			const float4 t2 = P.scene.tris[3 * ident + 2];
			normal = normalize(cross(mk3(t0.w, t1.x, t1.y), mk3(t1.z, t1.w, t2.x)));
			reflection_type = TYR_DIFF;
			object_color = mk3(1.f, 1.f, 1.f);
			if (P.flags & TYR_FLAG_TRIANGLE_MATERIALS) {
				const uint32_t m = __float_as_uint(t2.y);
				reflection_type = m <= (uint32_t)(LIGHTS ? TYR_LIGHT : TYR_PHONG) ? (int)m : TYR_DIFF;
			}
		}
	}
	afterLoads();
	if (hit) {
		const bool outside = dot(normal, direction) < 0;
		normal = outside ? normal : normal * -1.f;
		origin = origin + normal * kEpsilon;

		if (reflection_type == TYR_LIGHT) {
			if (lastSpecular) {
				if (LIGHTS && !(ident & kHitSphere))
					color = direct * mk3(P.triEmission[0], P.triEmission[1], P.triEmission[2]);
				else
					color = direct * ld3(P.spheres[ident & 7u].emmission);
			} else {
				color = mk3(0.f, 0.f, 0.f);
				direct = mk3(0.f, 0.f, 0.f);
			}
		}
		lastSpecular = false;
		constexpr float phongexponent = 40.0f;
		switch (reflection_type) {
		case TYR_LIGHT:
			break;
		case TYR_DIFF: {
			const f3 sunSampleDir = cone_sample(P.sun, seed);
			const float sunLight = dot(normal, sunSampleDir);
			if (rng_float(seed) < 0.5f) {
				if (sunLight > 0.f) {
					out.shadow = true;
					out.sOrigin = origin;
					out.sDir = sunSampleDir;
					out.sColor = 2.0f * direct; // x ((sun(sunSampleDir) * sunLight) * 1E-5f) below, kernel.cu:414
					atmo = kAtmoSun;
					atmoScale = sunLight;
					out.sClosest = 1e20f; // variables.h:41
				}
			} else {
				const EmitterSample E = sample_emitter<LIGHTS>(P, seed, origin, normal);
				const LightSample& L = E.L;
				if (L.valid) {
					const float closestAllowed = length(L.lightVector);
					const float solidAngle = (L.cosLightToSurface * E.area) / dot(L.lightVector, L.lightVector);
					out.shadow = true;
					out.sOrigin = origin;
					out.sDir = L.lightDir;
					out.sColor = ((((E.emission * 2.0f) * direct) * solidAngle) * kInvPi) * L.cosSurfaceToLight;
					out.sClosest = closestAllowed;
				}
			}
			if (bounces < kMaxBounces) {
				const float r1 = 2.f * kPi * rng_float(seed);
				const float r2 = rng_float(seed);
				const float r2s = sqrtf(r2);
				f3 u, v;
				orthonormal_basis_naive(normal, u, v);
				float s1, c1;
				dm::sincosf_det(r1, s1, c1);
				direction = normalize((u * c1) * r2s + (v * s1) * r2s + normal * sqrtf(1 - r2));
			}
			break;
		}
		case TYR_SPEC: {
			lastSpecular = true;
			direction = reflect(direction, normal);
			break;
		}
		case TYR_REFR: {
			// kernel.cu:476-515 (n1/n2 = 1.2/1.0 "defying convention")
			const float n1 = outside ? 1.2f : 1.0f;
			const float n2 = outside ? 1.0f : 1.2f;
			float fresnel = 0;
			float r0 = (n1 - n2) / (n1 + n2);
			r0 *= r0;
			const float cosI = -dot(normal, direction);
			const float n = n2 / n1;
			const float sinT2 = n * n * (1.0f - cosI * cosI);
			if (sinT2 > 1.0f) {
				fresnel = 1.0f;
			} else {
				const float x = 1.0f - cosI;
				fresnel = r0 + (1.0f - r0) * x * x * x * x * x;
			}
			if (rng_float(seed) < fresnel) {
				lastSpecular = true;
				direction = reflect(direction, normal);
			} else {
				origin = origin - (normal * 2.f) * kEpsilon;
				const float cosT = sqrtf(1.0f - sinT2);
				direction = n * direction + (n * cosI - cosT) * normal;
			}
			if (!outside) {
				const f3 e = (-object_color) * distance;
				direct = direct * mk3(dm::expf_det(e.x), dm::expf_det(e.y), dm::expf_det(e.z));
			}
			break;
		}
		case TYR_PHONG: {
			f3 w, u, v, d;
			do {
				const float phi = 2 * kPi * rng_float(seed);
				const float r2 = rng_float(seed);
				const float cosTheta = dm::powf_det(1.0f - r2, 1.0f / (phongexponent + 1.0f));
				const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
				w = direction - (normal * 2.0f) * dot(normal, direction);
				w = normalize(w);
				orthonormal_basis_naive(w, u, v);
				float sp, cp;
				dm::sincosf_det(phi, sp, cp);
				d = (u * cp) * sinTheta + (v * sp) * sinTheta + w * cosTheta;
				d = normalize(d);
			} while (dot(d, normal) <= kEpsilon);

			const f3 sunSampleDir = cone_sample(P.sun, seed);
			float sunLight = dot(normal, sunSampleDir);
			if (rng_float(seed) < 0.5f) {
				if (sunLight > 0.f) {
					const float phongCos = dot(sunSampleDir, w);
					if (phongCos > kEpsilon) {
						sunLight *= dm::powf_det(phongCos, phongexponent);
						out.shadow = true;
						out.sOrigin = origin;
						out.sDir = sunSampleDir;
						out.sColor = (2.0f * direct) * ((phongexponent + 2) * 0.5f * kInvPi); // x ((sun(..) * sunLight) * 1E-5f) below
						atmo = kAtmoSun;
						atmoScale = sunLight;
						out.sClosest = 1e20f;
					}
				}
			} else {
				const EmitterSample E = sample_emitter<LIGHTS>(P, seed, origin, normal);
				const LightSample& L = E.L;
				if (L.valid) {
					float phongCos = dot(L.lightDir, w);
					if (phongCos > kEpsilon) {
						phongCos = dm::powf_det(phongCos, phongexponent);
						const float closestAllowed = length(L.lightVector);
						const float solidAngle = (L.cosLightToSurface * E.area) / dot(L.lightVector, L.lightVector);
						f3 sc = (E.emission * 2.0f) * direct;
						sc = sc * solidAngle;
						sc = sc * (phongexponent + 2);
						sc = sc * 0.5f;
						sc = sc * kInvPi;
						sc = sc * phongCos;
						sc = sc * L.cosSurfaceToLight;
						out.shadow = true;
						out.sOrigin = origin;
						out.sDir = L.lightDir;
						out.sColor = sc;
						out.sClosest = closestAllowed;
					}
				}
			}
			origin = origin + w * kEpsilon;
			direction = d;
			break;
		}
		}

	} else if (valid) {
		atmo = lastSpecular ? kAtmoSunSky : kAtmoSky; // kernel.cu:613-617: nothing hit
	}

	// The atmosphere (sunsky.cu) is the most expensive thing a ray can ask for here, and three kinds of lanes ask:
	// a diffuse or Phong hit whose next-event sample went to the sun (sun(sunSampleDir), kernel.cu:414 / 553), and a
	// miss (sky / sunsky(direction), kernel.cu:613-617).  A ray asks at most once, nothing random is drawn in
	// between, so all of them evaluate it HERE, in one pass of the wave, instead of one pass per place of call; every
	// lane still performs exactly the operations the reference's order of evaluation prescribes.
	if (atmo != kAtmoNone) {
		const bool miss = !hit;
		const f3 viewDir = miss ? direction : out.sDir;
		if (atmo == kAtmoSunSky && P.sun.sunAngularDiameterCos == 1.0f) {
			color = color + direct * mk3(1.0f, 0.0f, 0.0f); // sunsky.cu:118-119
		} else {
			const Atmosphere a = atmosphere(P.sun, viewDir);
			if (atmo == kAtmoSun)
				out.sColor = out.sColor * ((sun_radiance(P.sun, a) * atmoScale) * 1E-5f);
			else
				color = color + (atmo == kAtmoSky ? direct * sky_radiance(a) : direct * sunsky_radiance(P.sun, a));
		}
	}

	if (hit) {
		// Russian roulette, kernel.cu:599-611
		const float p = gmin(1.0f, gmax(direct.z, gmax(direct.x, direct.y)));
		if (bounces < kMaxBounces && p > (0 + kEpsilon) && rng_float(seed) <= p) {
			bounces++;
			direct = direct * (1.0f / p);
			out.survive = true;
			out.origin = origin;
			out.direction = direction;
			out.direct = direct;
			out.flags = (uint32_t)bounces | ((lastSpecular ? 1u : 0u) << 8);
		} else {
			new_frame++;
		}
	} else if (valid) {
		new_frame++;
	}

	out.color = color;
	out.newFrame = new_frame;
}

// kernel.cu:622-625.  Adding +0 leaves the pixel unchanged, so zero terms are skipped (the reference's
// own TODO at kernel.cu:621).  One lane, one pixel: used by the per-slot and first persistent kernels (variants
// 0-1); the production kernels add a whole wave's contributions at once (accumulate_pixels_wave below).  vmcnt
// retires loads, stores and atomics in issue order and __syncthreads() waits for vmcnt(0), so where these are
// issued matters: in front of a barrier every wave sits out its own scattered atomics (~3000 cycles under load).
__device__ __forceinline__ void accumulate_pixel(float4* blit, int pixel, f3 color, int new_frame) {
	float* px = reinterpret_cast<float*>(&blit[pixel]);
#ifdef TYR_WHATIF_NO_ATOMICS
	if (pixel != 12345)
		return;
#endif
	if (color.x != 0.0f)
		atomicAdd(px + 0, color.x);
	if (color.y != 0.0f)
		atomicAdd(px + 1, color.y);
	if (color.z != 0.0f)
		atomicAdd(px + 2, color.z);
	if (new_frame)
		atomicAdd(px + 3, (float)new_frame);
}

// The same for a whole wave at once (every lane must call it; lanes without a contribution pass zeros).  A pixel is
// 16 bytes, so "lane l adds its red" spreads one instruction over 64 pixels = eight 128-byte lines with four useful
// bytes in sixteen, four times over for r, g, b and the count.  Here the wave transposes first: instruction j covers
// the pixels of lanes 16j .. 16j + 15, lane l adding component l % 4 of lane 16j + l / 4 -- consecutive queue slots are
// (mostly) consecutive pixels, so an instruction now touches two lines instead of eight.  Same sums, same skipping
// of zero terms.
__device__ __forceinline__ void accumulate_pixels_wave(float4* blit, int pixel, f3 color, int new_frame) {
	const uint32_t lane = lane_id();
	const uint32_t c = lane & 3u;
	const float w = (float)new_frame;
#pragma unroll
	for (uint32_t j = 0; j < 4; ++j) {
		const int src = (int)(16u * j + (lane >> 2));
		const float x = __shfl(color.x, src, 64), y = __shfl(color.y, src, 64), z = __shfl(color.z, src, 64), n = __shfl(w, src, 64);
		const int px = __shfl(pixel, src, 64);
		const float v = c == 0u ? x : (c == 1u ? y : (c == 2u ? z : n));
		if (v != 0.0f)
			atomicAdd(reinterpret_cast<float*>(&blit[px]) + c, v);
	}
}

// look-back descriptor: [63:62] status, [61:31] survivors, [30:0] shadow rays
constexpr unsigned long long kDescAggregate = 1ull << 62;
constexpr unsigned long long kDescInclusive = 2ull << 62;
__device__ __forceinline__ unsigned long long desc_pack(uint32_t s, uint32_t h) { return ((unsigned long long)s << 31) | (unsigned long long)h; }
__device__ __forceinline__ uint32_t desc_s(unsigned long long d) { return (uint32_t)((d >> 31) & 0x7fffffffull); }
__device__ __forceinline__ uint32_t desc_h(unsigned long long d) { return (uint32_t)(d & 0x7fffffffull); }

// Tile order without a ticket.  The stable compaction needs every tile's predecessors to be running
// (or done) while it looks back.  The first version drew a virtual tile id from an atomic counter at
// block start -- 8192 returning atomics on one word per launch, ~0.2 ms of a 0.3 ms kernel (one word
// serves ~88 of them per microsecond; measured by replacing the ticket: 0.30 -> 0.096 ms per launch).
// Now the grid is small enough to be entirely co-resident (at most 4 blocks of 256 threads per CU) and
// block b shades tiles b, b + G, b + 2G, ...: the predecessor of any tile belongs to a block that is
// resident, whatever order the hardware dispatched them in, so the look-back cannot starve.
//
// Deferred look-back.  With "shade tile i, look back for tile i, write tile i" every block waited, tile after
// tile, for the SLOWEST of its predecessors to publish an aggregate (all resident blocks shade the same
// generation of tiles at the same time and shading time has a long tail): an s_memtime build showed two thirds
// of a tile's time inside the look-back, and fetching more descriptors per round trip did not help.  Now the
// tile's compacted records wait in LDS (survivors and shadow rays at their rank inside the tile) and the
// look-back for tile i runs AFTER tile i+G has been shaded: by then every predecessor has long published, the
// look-back is pure round trips (kWindows x 64 descriptors each), and the records leave LDS as coalesced stores
// (thread t writes record t).
struct ShadeStage { // one tile's compacted output, 23 KB
	float4 sv_o_dx[kBlock];
	float2 sv_dyz[kBlock];
	float4 sv_direct_ix[kBlock];
	uint32_t sv_flags[kBlock];
	float4 sh_o_dx[kBlock];
	float4 sh_dyz_cd_ix[kBlock];
	float4 sh_color[kBlock];
};

// Exclusive prefix of tile vb over all lower tiles, computed by the whole block; publishes the tile's inclusive
// prefix.  Wave w inspects descriptors vb-1-512w ... vb-512(w+1) (kWindows x 64, lane i of window k reads one), so
// one memory round trip covers 2048 predecessors -- more than the distance to the nearest inclusive prefix, which
// is one to two generations of resident tiles (<= 1024 each) because the look-back is deferred by one tile.
// With wave 0 alone and 512 descriptors per step it took three steps, 36 % of a tile's time.
__device__ __forceinline__ void shade_lookback(const FrameParams& P, uint32_t vb, uint32_t totS, uint32_t totH, uint32_t tid, uint32_t nTiles, uint32_t* sh, uint32_t& esOut, uint32_t& ehOut) {
	constexpr int kWindows = 2;
	constexpr int kPerWave = 64 * kWindows, kPerStep = kPerWave * (int)(kBlock / 64);
	const uint32_t lane = tid & 63u, wave = tid >> 6;
	uint32_t es = 0, eh = 0;
	if (vb > 0) { // block-uniform
		int base = (int)vb - 1; // nearest predecessor of this step
		for (;;) {
			const int first = base - kPerWave * (int)wave; // nearest descriptor of this wave's share
			unsigned long long d[kWindows];
#pragma unroll
			for (int k = 0; k < kWindows; ++k) {
				const int idx = first - 64 * k - (int)lane;
				d[k] = kDescInclusive; // below tile 0: an inclusive prefix of zero
				if (idx >= 0)
					d[k] = __hip_atomic_load(&P.scanDesc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			bool found = false, timeout = false;
			uint32_t ps = 0, ph = 0;
#pragma unroll
			for (int k = 0; k < kWindows; ++k) {
				if (found)
					continue; // (wave-uniform) an inclusive prefix was found in a nearer window
				const int idx = first - 64 * k - (int)lane;
				uint32_t spins = 0;
				const unsigned long long t0_ = __builtin_amdgcn_s_memrealtime();
				while (!(d[k] >> 62)) { // not published yet: poll this one
					if (++spins > (1u << 25)) { // every wait is bounded (~4 s): report, never hang
						timeout = true;
						d[k] = kDescInclusive;
						// what timed out, for TYR_VERBOSE's report (host/driver.cpp check_device_error)
						P.k->debug[0] = vb;
						P.k->debug[1] = (unsigned long long)idx;
						P.k->debug[2] = blockIdx.x;
						P.k->debug[3] = gridDim.x;
						P.k->debug[4] = __builtin_amdgcn_s_memrealtime() - t0_; // 100 MHz ticks
						break;
					}
					__builtin_amdgcn_s_sleep(1);
					d[k] = __hip_atomic_load(&P.scanDesc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
				const unsigned long long inclMask = __ballot((d[k] >> 62) == 2ull);
				// lanes up to and including the nearest inclusive descriptor contribute
				const uint32_t stop = inclMask ? (uint32_t)__ffsll((long long)inclMask) - 1u : 63u;
				unsigned long long v = (lane <= stop) ? (d[k] & ~(3ull << 62)) : 0ull;
#pragma unroll
				for (int o = 32; o > 0; o >>= 1)
					v += __shfl_xor(v, o, 64);
				ps += desc_s(v);
				ph += desc_h(v);
				found = (inclMask != 0ull);
			}
			if (__ballot(timeout) != 0ull && lane == 0)
				atomicOr(&P.k->device_error, kErrScanTimeout);
			if (lane == 0) {
				sh[16 + 3 * wave + 0] = ps;
				sh[16 + 3 * wave + 1] = ph;
				sh[16 + 3 * wave + 2] = found ? 1u : 0u;
			}
			__syncthreads();
			bool any = false;
#pragma unroll
			for (uint32_t w = 0; w < kBlock / 64; ++w) {
				if (!any) {
					es += sh[16 + 3 * w + 0];
					eh += sh[16 + 3 * w + 1];
					any = sh[16 + 3 * w + 2] != 0u;
				}
			}
			__syncthreads(); // sh[16..] may be rewritten by another step
			if (any)
				break;
			base -= kPerStep;
		}
	}
	if (tid == 0) {
		__hip_atomic_store(&P.scanDesc[vb], kDescInclusive | desc_pack(es + totS, eh + totH), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (vb == nTiles - 1) {
			// kernel.cu:607 / 416: the totals the next top-up and connect read
			P.k->primary_ray_cnt = es + totS;
			P.k->shadow_ray_cnt = eh + totH;
			P.kc->shadow_cnt = eh + totH;
			P.k->total_shadow_rays += eh + totH;
			P.k->n_survive += es + totS;
		}
	}
	esOut = es;
	ehOut = eh;
}

template <bool LIGHTS>
__global__ void __launch_bounds__(kBlock) k_shade(const FrameParams P, uint32_t nTiles) {
	__shared__ uint32_t sh[32];
	__shared__ ShadeStage stage;
	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63u, wave = tid >> 6;
	const uint32_t nLive = P.k->n_live;
	const unsigned long long below = (1ull << lane) - 1ull;
#ifdef TYR_SHADE_TIMING
	// diagnostic build: where a tile's time goes, in s_memtime ticks summed over this block's tiles (thread 0;
	// debug[0] shade, [1] ranks + barrier, [2] look-back, [3] barrier after it, [4] copy out + barrier, [5] stage +
	// pixel atomics, [7] tiles)
	unsigned long long tacc_[6] = { 0, 0, 0, 0, 0, 0 }, t_ = __builtin_amdgcn_s_memtime(), ntiles_ = 0;
#define TYR_STAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc_[i] += now_ - t_; t_ = now_; }
#else
#define TYR_STAMP(i)
#endif
	bool havePrev = false;            // a shaded tile whose records wait in `stage`
	uint32_t pendPixel = 0;           // this lane's pixel contribution of that tile, not yet added
	int pendNew = 0;
	f3 pendColor = mk3(0.f, 0.f, 0.f);
	uint32_t prevVb = 0, prevS = 0, prevH = 0;

	// finish the waiting tile: look back, then move its records from LDS to their slots (kernel.cu:607-608, 416-417)
	auto flush_prev = [&]() {
		uint32_t es, eh;
		shade_lookback(P, prevVb, prevS, prevH, tid, nTiles, sh, es, eh);
		TYR_STAMP(2)
		// one array at a time (the compiler barrier keeps it from loading all seven records first): this copy is where
		// the kernel's register count peaks
		if (tid < prevS) {
			P.next.o_dx[es + tid] = stage.sv_o_dx[tid];
			__asm__ volatile("" ::: "memory");
			P.next.direct_ix[es + tid] = stage.sv_direct_ix[tid];
			__asm__ volatile("" ::: "memory");
			P.next.dyz[es + tid] = stage.sv_dyz[tid];
			P.next.flags[es + tid] = stage.sv_flags[tid];
		}
		__asm__ volatile("" ::: "memory");
		if (tid < prevH) {
			P.shadow.o_dx[eh + tid] = stage.sh_o_dx[tid];
			__asm__ volatile("" ::: "memory");
			P.shadow.dyz_cd_ix[eh + tid] = stage.sh_dyz_cd_ix[tid];
			__asm__ volatile("" ::: "memory");
			P.shadow.color[eh + tid] = stage.sh_color[tid];
		}
		__syncthreads(); // `stage` and sh[] are free again
		TYR_STAMP(4)
	};

	// Tiles are drawn from eight tickets (word w hands out tiles w, w + 8, ...; a block starts at word
	// blockIdx % 8 and moves on when a word is used up).  A tile is only ever started after every lower tile of
	// its word, and the lowest tile not yet started always belongs to a word whose running tiles are lower still,
	// so every tile a look-back waits for is being shaded by some block: no starvation, whatever the grid size.
	// Unlike the fixed assignment b, b + G, ... a slow block simply shades fewer tiles instead of holding up every
	// look-back of its generation; one word per tile id would be a single ticket again (88 draws/us: 0.74 ms for
	// the 64.8 k tiles of a full queue).
	uint32_t word = blockIdx.x % kTicketWords, tried = 0;
	auto draw_tile = [&]() -> uint32_t { // block-uniform; nTiles when nothing is left
		uint32_t vbNext = nTiles;
		if (tid == 0) {
			while (tried < kTicketWords) {
				const uint32_t t = atomicAdd(&P.k->shade_tiles[word * 32], 1u);
				const unsigned long long cand = (unsigned long long)t * kTicketWords + word;
				if (cand < nTiles) {
					vbNext = (uint32_t)cand;
					break;
				}
				word = (word + 1) % kTicketWords;
				++tried;
			}
			sh[3] = vbNext;
		}
		__syncthreads();
		vbNext = sh[3];
		__syncthreads();
		return vbNext;
	};
	for (uint32_t vb = draw_tile(); vb < nTiles; vb = draw_tile()) { // vb = tile id = queue order
		const uint32_t slot = vb * kBlock + tid;
		ShadeOut out = {};
		uint32_t pixelBits = 0;
		// kernel.cu:622-625 for the tile BEFORE this one.  vmcnt retires loads and atomics in issue order, and an
		// atomic that has to reach the memory side takes thousands of cycles under load: issued at the end of a
		// tile they sat in front of the next tile's ray loads (0.34 ms of a render's 1.77 ms of shade, measured by
		// leaving them out).  Issued here -- this tile's loads are back, ~1000 instructions of arithmetic follow --
		// nobody waits for them.
		auto flush_pixels = [&]() { // reached by every lane of every wave: lanes with nothing pending add nothing
			accumulate_pixels_wave(P.blit, (int)pendPixel, pendColor, pendNew);
			pendColor = mk3(0.f, 0.f, 0.f);
			pendNew = 0;
		};
		const bool valid = slot < nLive;
		if (valid)
			pixelBits = __float_as_uint(P.work.direct_ix[slot].w);
		shade_ray<LIGHTS>(P, slot, valid, out, flush_pixels);
		TYR_STAMP(0)

		// ---- stable compaction of survivors and shadow rays: ranks inside the tile ----
		const unsigned long long bs = __ballot(out.survive);
		const unsigned long long bh = __ballot(out.shadow);
		const uint32_t rs = __popcll(bs & below), rh = __popcll(bh & below);
		if (lane == 0) {
			sh[4 + wave] = __popcll(bs);
			sh[8 + wave] = __popcll(bh);
		}
		__syncthreads();
		uint32_t ws = 0, wh = 0, totS = 0, totH = 0;
#pragma unroll
		for (uint32_t w = 0; w < kBlock / 64; ++w) {
			const uint32_t cs = sh[4 + w], ch = sh[8 + w];
			if (w < wave) {
				ws += cs;
				wh += ch;
			}
			totS += cs;
			totH += ch;
		}
		totS = (uint32_t)__builtin_amdgcn_readfirstlane((int)totS); // block-uniform: keep them out of the vector registers
		totH = (uint32_t)__builtin_amdgcn_readfirstlane((int)totH);
		// the aggregate goes out at once: later tiles can add it up long before this tile knows its own prefix.
		// Descriptor = 8 bytes {status, survivors, shadows} written by one relaxed agent-scope store: payload and
		// flag travel together, no fence needed.  Tile 0 publishes one too: its inclusive prefix only appears when its
		// deferred look-back runs, and every look-back of the first generation would sit waiting for it.
		if (tid == 0)
			__hip_atomic_store(&P.scanDesc[vb], kDescAggregate | desc_pack(totS, totH), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		TYR_STAMP(1)
		if (havePrev)
			flush_prev(); // ends with a barrier: sh[4..11] have been read by every thread
		else
			__syncthreads();
		if (out.survive) {
			const uint32_t k = ws + rs;
			stage.sv_o_dx[k] = make_float4(out.origin.x, out.origin.y, out.origin.z, out.direction.x);
			stage.sv_dyz[k] = make_float2(out.direction.y, out.direction.z);
			stage.sv_direct_ix[k] = make_float4(out.direct.x, out.direct.y, out.direct.z, __uint_as_float(pixelBits));
			stage.sv_flags[k] = out.flags;
		}
		if (out.shadow) {
			const uint32_t k = wh + rh;
			stage.sh_o_dx[k] = make_float4(out.sOrigin.x, out.sOrigin.y, out.sOrigin.z, out.sDir.x);
			stage.sh_dyz_cd_ix[k] = make_float4(out.sDir.y, out.sDir.z, out.sClosest, __uint_as_float(pixelBits));
			stage.sh_color[k] = make_float4(out.sColor.x, out.sColor.y, out.sColor.z, 0.0f);
		}
		havePrev = true;
		prevVb = vb;
		prevS = totS;
		prevH = totH;
		// goes to the pixel under the next tile's arithmetic (or after the loop); zeros for lanes past the end
		pendPixel = pixelBits;
		pendColor = out.color;
		pendNew = out.newFrame;
		// no barrier here: the next tile's first barrier orders these LDS writes before flush_prev reads them
		TYR_STAMP(5)
#ifdef TYR_SHADE_TIMING
		++ntiles_;
#endif
	}
	accumulate_pixels_wave(P.blit, (int)pendPixel, pendColor, pendNew);
	if (havePrev) {
		__syncthreads();
		flush_prev();
	}
#ifdef TYR_SHADE_TIMING
	if (tid == 0) {
		for (int i = 0; i < 6; ++i)
			atomicAdd(&P.k->debug[i], tacc_[i]);
		atomicAdd(&P.k->debug[7], ntiles_);
	}
#endif
#undef TYR_STAMP
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

// extend pre-pass: kernel.cu:125-136 (spheres first; their distance bounds the BVH search)
__global__ void __launch_bounds__(kBlock) k_extend_spheres(const FrameParams P) {
	const uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
	if (slot == 0)
		P.k->extend_ticket = 0; // the persistent kernel that follows on the stream starts from slot 0
	if (slot >= P.k->first_fresh) // slots from there to n_live are this iteration's primary rays: k_primary has done them
		return;
	const float4 a = P.work.o_dx[slot];
	const float2 b = P.work.dyz[slot];
	P.work.hit[slot] = sphere_hit_record(P, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y));
}

__device__ __forceinline__ uint32_t root_ref(const DevScene& sc, const RayConst& r, float bound) {
	float t0;
	const bool ok = slab_test(r, r.nx ? sc.rootMax[0] : sc.rootMin[0], r.nx ? sc.rootMin[0] : sc.rootMax[0], r.ny ? sc.rootMax[1] : sc.rootMin[1], r.ny ? sc.rootMin[1] : sc.rootMax[1],
		r.nz ? sc.rootMax[2] : sc.rootMin[2], r.nz ? sc.rootMin[2] : sc.rootMax[2], bound, t0);
	return ok ? sc.rootRef : kRefDone;
}

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
// Register budget of the flat traversal kernels, as waves per SIMD.  Five (<= 96 VGPRs) measured 7-10 % faster in
// extend than the four the allocator picks by itself (99 VGPRs); six (80 VGPRs) spills 32 registers in the descent
// loop and is 40 % slower.  Deeper LDS stacks cap the occupancy below five anyway.
#ifndef TYR_CONNECT_ORDERED
#define TYR_CONNECT_ORDERED false
#endif
#ifndef TYR_FLAT_WAVES_PER_EU
#define TYR_FLAT_WAVES_PER_EU (STACK_LDS <= 8 ? 6 : STACK_LDS <= 12 ? 5 : STACK_LDS <= 16 ? 3 : 2)
#endif

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

// connect pre-pass: the sphere half of intersect_scene_simple (kernel.cu:168-172).  Any-hit does not
// depend on test order, so spheres go first and an occluded ray never enters the BVH.
// color.w (unused by the reference's 44-byte record) carries the flag.
__global__ void __launch_bounds__(kBlock) k_connect_spheres(const FrameParams P) {
	const uint32_t index = blockIdx.x * kBlock + threadIdx.x;
	if (index == 0)
		P.kc->ticket = 0;
	if (index >= P.kc->shadow_cnt)
		return;
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


// ======================================================================================
// Flat traversal (variant 2): persistent waves, lane refill, and NO nested divergent loops.
//
// Measured on variant 1 with the counting build (tools/loop_occupancy.py, C2 at 1080p): the
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
		const uint32_t waves = gridDim.x * (kBlock / 64);
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

// slots of the queue that each block of a persistent grid owns outright (a multiple of 64; 0 for thin queues)
__device__ __forceinline__ uint32_t static_range(uint32_t nItems, uint32_t sixteenths) {
	const unsigned long long share = (unsigned long long)nItems * sixteenths / 16ull;
	return (uint32_t)(share / gridDim.x) & ~63u;
}

template <bool COUNT, int STACK_LDS, bool QUAD, bool PERSIST>
__global__ void __launch_bounds__(kBlock, TYR_FLAT_WAVES_PER_EU) k_extend_flat(const FrameParams P) {
	static_assert(!(COUNT && QUAD), "only pair nodes reproduce the reference's visit counts");
	TYR_DECLARE_FLAT_STACK(st, true)
	// variant 4: the top of the tree lives in LDS for the lifetime of the (persistent) block
	__shared__ float4 stagedNodes[PERSIST ? 7 * kStagedNodes : 1];
	const uint32_t nStaged = (PERSIST && QUAD) ? P.scene.nStaged : 0u;
	if (PERSIST && QUAD) {
		for (uint32_t i = threadIdx.x; i < 7 * nStaged; i += kBlock) {
			const uint32_t v = i / nStaged, n = i - v * nStaged;
			stagedNodes[v * kStagedNodes + n] = P.scene.quads[8 * n + v];
		}
		// visible to the block after the __syncthreads() that precedes the main loop
	}
	const uint32_t lane = lane_id();
	const unsigned long long below = (1ull << lane) - 1ull;
	const uint32_t nLive = P.k->n_live;
	const DevScene& sc = P.scene;
	// the ray of this lane as plain scalars: kept as one RayConst object across the refill branch, its first 16
	// bytes (origin + direction.x) stayed in a private-memory slot that every descent and leaf phase re-read
	float rox = 0.f, roy = 0.f, roz = 0.f, rdx = 0.f, rdy = 0.f, rdz = 0.f, rix = 0.f, riy = 0.f, riz = 0.f;
	bool regular = true;      // this lane's ray has a finite 1/d in all three components
	bool allRegular = true;   // ... and so has every live ray of the wave (wave-uniform; refreshed at refills)
	float dist = 0.0f;
	uint32_t ref = kRefDone, slot = 0;
	int prim = 0;
	bool hitTri = false, live = false, overflow = false;
	VisitCount vc{ 0, 0 };
	uint32_t dbg[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
	// Work distribution: this block owns queue slots [blockIdx.x * raysPerBlock, +raysPerBlock) and hands
	// them to the free lanes of its four waves through a counter in LDS.  Balancing ACROSS blocks is the
	// hardware dispatcher's (grid = slots / raysPerBlock blocks, more than fit at once).  The first
	// versions pulled from one device-wide ticket: a single word serves only ~88 returning atomics per
	// microsecond (MI355X_MICROARCH.md "dequeue"), and with >= 2 pulls per wave that alone was a
	// 0.26 ms floor per launch, whatever the traversal cost.
	__shared__ uint32_t blockNext;
	// variant 4 (PERSIST): the first staticShare/16 of the queue is dealt to the blocks as fixed ranges, handed out
	// through LDS exactly like variant 3 (no device-wide atomic: a launch of short rays -- the primary rays --
	// would spend a third of its time on ticket round trips); the rest goes out in ticketed chunks to whoever is
	// free, which evens out the blocks and leaves no block waiting for its longest ray.
	const uint32_t perBlock = PERSIST ? static_range(nLive, P.staticShare) : P.raysPerBlock;
	const uint32_t dynBase = PERSIST ? perBlock * gridDim.x : 0u;
	const uint32_t blockBegin = blockIdx.x * perBlock;
	const uint32_t blockEnd = PERSIST ? blockBegin + perBlock : ((blockBegin + perBlock) < nLive ? (blockBegin + perBlock) : nLive);
	ChunkFeed feed;
	feed.init(nLive - dynBase, P.ticketChunk);
	bool staticDone = (perBlock == 0); // wave-uniform
	if (threadIdx.x == 0)
		blockNext = blockBegin;
	__syncthreads();
	bool exhausted = (sc.rootRef == kRefDone) || (PERSIST ? nLive == 0 : blockBegin >= nLive);
	uint32_t passes = 0; // see kMaxPasses

	for (;;) {
		if (kGuardPasses && ++passes > kMaxPasses)
			break;
		// ---- refill free lanes from the queue ----
		const unsigned long long idleMask = __ballot(!live);
		const uint32_t nIdle = __popcll(idleMask);
		if (!exhausted && nIdle >= P.refillMinIdle) {
			const uint32_t rank = __popcll(idleMask & below);
			uint32_t s = 0;
			bool fed = false;
			if (PERSIST) {
				uint32_t got = 0; // idle lanes served so far; one refill may take from the fixed range and from two chunks
				if (!staticDone) {
					uint32_t base = 0;
					if (lane == 0)
						base = atomicAdd(&blockNext, nIdle); // LDS
					base = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl(base, 0, 64));
					const uint32_t avail = base < blockEnd ? blockEnd - base : 0u;
					got = avail < nIdle ? avail : nIdle;
					if (!live && rank < got) {
						s = base + rank;
						fed = true;
					}
					staticDone = (base + nIdle >= blockEnd);
				}
				while (staticDone && got < nIdle) {
					if (!feed.refill(P.k->extend_chunks, nLive - dynBase, lane)) {
						exhausted = true;
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
			} else {
				uint32_t base = 0;
				if (lane == 0)
					base = atomicAdd(&blockNext, nIdle); // LDS
				base = __shfl(base, 0, 64);
				const uint32_t avail = base < blockEnd ? blockEnd - base : 0u;
				const uint32_t take = avail < nIdle ? avail : nIdle;
				exhausted = (base + nIdle >= blockEnd);
				s = base + rank;
				fed = !live && rank < take;
			}
			{
				if (fed) {
					TYR_DBG(6)
					const float4 a = P.work.o_dx[s];
					const float2 b = P.work.dyz[s];
					const float2 h = P.work.hit[s];
					const RayConst nr = make_ray(mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y));
					rox = nr.o.x, roy = nr.o.y, roz = nr.o.z, rdx = nr.d.x, rdy = nr.d.y, rdz = nr.d.z, rix = nr.inv.x, riy = nr.inv.y, riz = nr.inv.z;
					regular = ray_is_regular(nr);
					dist = h.x;
					slot = s;
					hitTri = false;
					st.reset();
					ref = root_ref(sc, nr, dist);
					if (QUAD && ref != kRefDone)
						ref = sc.quadRootRef;
					// a ray that misses the root box (or is already stopped short of it by a sphere) is finished here:
					// the pre-pass's answer stands, nothing to write, the lane stays free
					live = (ref != kRefDone);
					if (COUNT)
						vc.nodes += 1;
				}
			}
			// Primary rays mostly end right there (three in four on C3): top the wave up again rather than run
			// the descent loop a quarter full.  Every pass consumes queue slots, so this terminates.
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
			// leave the descent when few lanes are still descending and there is anything else to do (leaves, or a
			// refill).  (Also leaving once many lanes hold a leaf, so that triangle tests run wide, was measured at
			// every threshold and never paid; the test cost eight instructions per trip.)
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
				if (QUAD) {
					const QuadHits q = allRegular ? test_quad<true, true, PERSIST>(sc.quads, ref, r, dist, stagedNodes, nStaged) : test_quad<false, true, PERSIST>(sc.quads, ref, r, dist, stagedNodes, nStaged);
					// the earliest hit in visit order is entered now, the later ones are pushed latest first:
					// entry k is pushed iff it hit and an earlier entry hit too
					const lanemask any01 = q.hit[0] | q.hit[1], any012 = any01 | q.hit[2];
					st.push3(q.hit[3] & any012, q.ref[3], q.t[3], q.hit[2] & any01, q.ref[2], q.t[2], q.hit[1] & q.hit[0], q.ref[1], q.t[1]);
					ref = lane_in(q.hit[0]) ? q.ref[0] : lane_in(q.hit[1]) ? q.ref[1] : lane_in(q.hit[2]) ? q.ref[2] : lane_in(q.hit[3]) ? q.ref[3] : kRefPop;
				} else {
					const PairTest p = allRegular ? test_pair_fast(sc.nodes, ref, r, dist) : test_pair(sc.nodes, ref, r, dist);
					if (COUNT && !p.synthetic)
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
		}
		// ---- leaves: bvh.h:129-140 ----
		if (ref_is_leaf(ref)) {
			const uint32_t off = ref & (kMaxPrimOffset - 1);
			const uint32_t cnt = ((ref >> 26) & 31u) + 1u;
			TriData tri = triangle_load(sc.tris, off);
			for (uint32_t i = 0; i < cnt; ++i) {
				TYR_DBG(4)
				// the next primitive of the leaf is on its way while this one is tested (a leaf is 1..4 consecutive records)
				const TriData cur = tri;
				if (i + 1 < cnt)
					tri = triangle_load(sc.tris, off + i + 1);
				const float t = triangle_test(cur, r);
				if (COUNT)
					vc.tris += 1;
				if (t > kEpsilon && t < dist && ((dist - t) > kEpsilon)) {
					prim = (int)(off + i);
					dist = t;
					hitTri = true;
				}
			}
			ref = kRefPop;
		}
		// ---- finished rays: a triangle hit replaces the sphere answer of the pre-pass (kernel.cu:138-140).
		// (Holding the record back until the wave's next refill, one store for all lanes that finished in between,
		// was measured: +1 %.) ----
		if (live && ref == kRefDone) {
			if (hitTri)
				P.work.hit[slot] = make_float2(dist, __uint_as_float((uint32_t)prim));
			overflow = overflow || st.overflow;
			live = false;
		}
	}
	if (overflow)
		atomicOr(&P.k->device_error, kErrStackOverflow);
	if (kGuardPasses && passes > kMaxPasses)
		atomicOr(&P.k->device_error, kErrNoProgress);
	if (COUNT) {
		wave_add_u64(&P.k->nodes_extend, vc.nodes);
		wave_add_u64(&P.k->tris_extend, vc.tris);
	}
	if (COUNT || kLoopStats) {
		for (int i = 0; i < 8; ++i)
			wave_add_u64(&P.k->debug[i], dbg[i]);
	}
}

template <bool COUNT, int STACK_LDS, bool QUAD, bool PERSIST>
__global__ void __launch_bounds__(kBlock, TYR_FLAT_WAVES_PER_EU) k_connect_flat(const FrameParams P) {
	static_assert(!(COUNT && QUAD), "only pair nodes reproduce the reference's visit counts");
	constexpr bool kKeepT = !(QUAD && !COUNT); // the pair / counting path marks failed boxes through the entry distance
	TYR_DECLARE_FLAT_STACK(st, kKeepT)
	// variant 4: the top of the tree lives in LDS for the lifetime of the (persistent) block
	__shared__ float4 stagedNodes[PERSIST ? 7 * kStagedNodes : 1];
	const uint32_t nStaged = (PERSIST && QUAD) ? P.scene.nStaged : 0u;
	if (PERSIST && QUAD) {
		for (uint32_t i = threadIdx.x; i < 7 * nStaged; i += kBlock) {
			const uint32_t v = i / nStaged, n = i - v * nStaged;
			stagedNodes[v * kStagedNodes + n] = P.scene.quads[8 * n + v];
		}
		// visible to the block after the __syncthreads() that precedes the main loop
	}
	const uint32_t lane = lane_id();
	const unsigned long long below = (1ull << lane) - 1ull;
	const uint32_t nRays = P.kc->shadow_cnt;
	const DevScene& sc = P.scene;
	const bool haveBvh = (sc.rootRef != kRefDone);
	float rox = 0.f, roy = 0.f, roz = 0.f, rdx = 0.f, rdy = 0.f, rdz = 0.f, rix = 0.f, riy = 0.f, riz = 0.f; // see k_extend_flat
	bool regular = true;      // this lane's ray has a finite 1/d in all three components
	bool allRegular = true;   // ... and so has every live ray of the wave (wave-uniform; refreshed at refills)
	float closest = 0.0f;
	uint32_t ref = kRefDone, index = 0;
	bool live = false, occluded = false, overflow = false;
	VisitCount vc{ 0, 0 };
	uint32_t visible = 0;
	// kernel.cu:640-644, deferred: a lane whose ray came through unoccluded notes the slot and goes idle; the wave
	// adds all such colours to their pixels at its next refill (and once after the loop), loads batched with the new
	// rays' loads and the atomics transposed (accumulate_pixels_wave) -- instead of two dependent loads and three
	// scattered atomics in the middle of the descent every time some lane finishes.
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
	const uint32_t perBlock = PERSIST ? static_range(nRays, P.staticShare) : P.raysPerBlock; // see k_extend_flat
	const uint32_t dynBase = PERSIST ? perBlock * gridDim.x : 0u;
	const uint32_t blockBegin = blockIdx.x * perBlock;
	const uint32_t blockEnd = PERSIST ? blockBegin + perBlock : ((blockBegin + perBlock) < nRays ? (blockBegin + perBlock) : nRays);
	ChunkFeed feed;
	feed.init(nRays - dynBase, P.ticketChunk);
	bool staticDone = (perBlock == 0); // wave-uniform
	if (threadIdx.x == 0)
		blockNext = blockBegin;
	__syncthreads();
	bool exhausted = PERSIST ? nRays == 0 : (blockBegin >= nRays);
	const float kFailed = __builtin_inff();
	uint32_t passes = 0; // see kMaxPasses

	for (;;) {
		if (kGuardPasses && ++passes > kMaxPasses)
			break;
		const unsigned long long idleMask = __ballot(!live);
		const uint32_t nIdle = __popcll(idleMask);
		if (!exhausted && nIdle >= P.refillMinIdle) {
			const uint32_t rank = __popcll(idleMask & below);
			uint32_t s = 0;
			bool fed = false;
			if (PERSIST) {
				uint32_t got = 0;
				if (!staticDone) {
					uint32_t base = 0;
					if (lane == 0)
						base = atomicAdd(&blockNext, nIdle); // LDS
					base = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl(base, 0, 64));
					const uint32_t avail = base < blockEnd ? blockEnd - base : 0u;
					got = avail < nIdle ? avail : nIdle;
					if (!live && rank < got) {
						s = base + rank;
						fed = true;
					}
					staticDone = (base + nIdle >= blockEnd);
				}
				while (staticDone && got < nIdle) {
					if (!feed.refill(P.kc->chunks, nRays - dynBase, lane)) {
						exhausted = true;
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
			} else {
				uint32_t base = 0;
				if (lane == 0)
					base = atomicAdd(&blockNext, nIdle); // LDS
				base = __shfl(base, 0, 64);
				const uint32_t avail = base < blockEnd ? blockEnd - base : 0u;
				const uint32_t take = avail < nIdle ? avail : nIdle;
				exhausted = (base + nIdle >= blockEnd);
				s = base + rank;
				fed = !live && rank < take;
			}
			if (__ballot(pendIdx != kNoPending) != 0ull)
				flush_visible();
			{
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
					if (haveBvh && (COUNT || !occluded)) {
						const RayConst nr = make_ray(mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y));
						rox = nr.o.x, roy = nr.o.y, roz = nr.o.z, rdx = nr.d.x, rdy = nr.d.y, rdz = nr.d.z, rix = nr.inv.x, riy = nr.inv.y, riz = nr.inv.z;
						regular = ray_is_regular(nr);
						ref = root_ref(sc, nr, closest);
						if (QUAD && ref != kRefDone)
							ref = sc.quadRootRef;
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
		allRegular = (__ballot(live && !regular) == 0ull);
		const RayConst r = { mk3(rox, roy, roz), mk3(rdx, rdy, rdz), mk3(rix, riy, riz), rix < 0, riy < 0, riz < 0 }; // bvh.h:120-121
		for (;;) {
			const uint32_t nTrav = __popcll(lanes_traversing(ref));
			if (nTrav == 0)
				break;
			// leave the descent when few lanes are still descending and there is anything else to do (leaves, or a
			// refill).  (Also leaving once many lanes hold a leaf, so that triangle tests run wide, was measured at
			// every threshold and never paid; the test cost eight instructions per trip.)
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
					if (COUNT)
						vc.nodes += 1; // the reference fetches the popped node before testing its box (bvh.h:222-224)
					if (pt < closest)
						ref = pr;
				} else {
					ref = kRefDone;
				}
			}
			if ((int)ref >= 0) {
				if (QUAD) {
					const QuadHits q = allRegular ? test_quad<true, TYR_CONNECT_ORDERED, PERSIST>(sc.quads, ref, r, closest, stagedNodes, nStaged) : test_quad<false, TYR_CONNECT_ORDERED, PERSIST>(sc.quads, ref, r, closest, stagedNodes, nStaged);
					const lanemask any01 = q.hit[0] | q.hit[1], any012 = any01 | q.hit[2];
					st.push3(q.hit[3] & any012, q.ref[3], q.t[3], q.hit[2] & any01, q.ref[2], q.t[2], q.hit[1] & q.hit[0], q.ref[1], q.t[1]);
					ref = lane_in(q.hit[0]) ? q.ref[0] : lane_in(q.hit[1]) ? q.ref[1] : lane_in(q.hit[2]) ? q.ref[2] : lane_in(q.hit[3]) ? q.ref[3] : kRefPop;
					continue;
				}
				const PairTest p = allRegular ? test_pair_fast(sc.nodes, ref, r, closest) : test_pair(sc.nodes, ref, r, closest);
				if (COUNT && !p.synthetic) {
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
				if (COUNT)
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
	if (COUNT) {
		wave_add_u64(&P.k->nodes_connect, vc.nodes);
		wave_add_u64(&P.k->tris_connect, vc.tris);
	}
}
#undef TYR_DBG


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

// ---- launch wrappers ---------------------------------------------------------------------
static inline uint32_t blocks_for(uint32_t n) { return (n + kBlock - 1) / kBlock; }

void launch_primary(const FrameParams& P, uint32_t maxNew, hipStream_t stream) {
	if (maxNew == 0)
		return;
	hipLaunchKernelGGL(k_primary, dim3(blocks_for(maxNew)), dim3(kBlock), 0, stream, P);
}
void launch_globals(const FrameParams& P, uint32_t nDesc, hipStream_t stream) {
	hipLaunchKernelGGL(k_globals, dim3(blocks_for(nDesc ? nDesc : 1)), dim3(kBlock), 0, stream, P, nDesc);
}
// persistent grids: as many 256-thread blocks as stay resident (no inter-block dependency, so a
// larger grid would only queue), never more waves than there are rays
template <class K>
static uint32_t persistent_blocks(K kernel, uint32_t nItems, const Tuning& t, int numCUs) {
	// the occupancy query is a host-side call of ~0.1-0.2 ms: ask once per kernel, not once per launch
	// (at every launch it sat between the pre-pass and the persistent kernel with the GPU idle)
	static int cachedPerCU = 0; // one instance per template instantiation = per kernel
	int perCU = 0;
	if (t.wavesPerSimd > 0) {
		perCU = t.wavesPerSimd; // 4 SIMDs x w waves = w blocks of 4 waves
	} else {
		if (cachedPerCU == 0) {
			int q = 0;
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, kernel, kBlock, 0) != hipSuccess || q <= 0)
				q = 4;
			cachedPerCU = q;
		}
		perCU = cachedPerCU;
	}
	const uint32_t resident = (uint32_t)perCU * (uint32_t)numCUs;
	const uint32_t needed = (nItems + kBlock - 1) / kBlock;
	return needed < resident ? (needed ? needed : 1) : resident;
}

template <bool COUNT, int STACK_LDS>
static void launch_extend_t(const FrameParams& P, uint32_t maxLive, uint32_t nSurvivors, const Tuning& t, int numCUs, hipStream_t stream) {
	if (t.traversalVariant == 0) {
		hipLaunchKernelGGL((k_extend<COUNT, STACK_LDS>), dim3(blocks_for(maxLive)), dim3(kBlock), 0, stream, P);
		return;
	}
	if (nSurvivors != 0)
		hipLaunchKernelGGL(k_extend_spheres, dim3(blocks_for(nSurvivors)), dim3(kBlock), 0, stream, P);
	const uint32_t flatBlocks = (maxLive + P.raysPerBlock - 1) / P.raysPerBlock;
	if (t.traversalVariant == 4 && !COUNT)
		hipLaunchKernelGGL((k_extend_flat<false, STACK_LDS, true, true>), dim3(persistent_blocks(k_extend_flat<false, STACK_LDS, true, true>, maxLive, t, numCUs)), dim3(kBlock), 0, stream, P);
	else if (t.traversalVariant == 3 && !COUNT)
		hipLaunchKernelGGL((k_extend_flat<false, STACK_LDS, true, false>), dim3(flatBlocks), dim3(kBlock), 0, stream, P);
	else if (t.traversalVariant >= 2)
		hipLaunchKernelGGL((k_extend_flat<COUNT, STACK_LDS, false, false>), dim3(flatBlocks), dim3(kBlock), 0, stream, P);
	else
		hipLaunchKernelGGL((k_extend_persistent<COUNT, STACK_LDS>), dim3(persistent_blocks(k_extend_persistent<COUNT, STACK_LDS>, maxLive, t, numCUs)), dim3(kBlock), 0, stream, P);
}
template <bool COUNT, int STACK_LDS>
static void launch_connect_t(const FrameParams& P, uint32_t maxShadow, const Tuning& t, int numCUs, hipStream_t stream) {
	if (t.traversalVariant == 0) {
		hipLaunchKernelGGL((k_connect<COUNT, STACK_LDS>), dim3(blocks_for(maxShadow)), dim3(kBlock), 0, stream, P);
		return;
	}
	hipLaunchKernelGGL(k_connect_spheres, dim3(blocks_for(maxShadow)), dim3(kBlock), 0, stream, P);
	const uint32_t flatBlocks = (maxShadow + P.raysPerBlock - 1) / P.raysPerBlock;
	if (t.traversalVariant >= 2) {
		if (t.traversalVariant == 4 && !COUNT)
			hipLaunchKernelGGL((k_connect_flat<false, STACK_LDS, true, true>), dim3(persistent_blocks(k_connect_flat<false, STACK_LDS, true, true>, maxShadow, t, numCUs)), dim3(kBlock), 0, stream, P);
		else if (t.traversalVariant == 3 && !COUNT)
			hipLaunchKernelGGL((k_connect_flat<false, STACK_LDS, true, false>), dim3(flatBlocks), dim3(kBlock), 0, stream, P);
		else
			hipLaunchKernelGGL((k_connect_flat<COUNT, STACK_LDS, false, false>), dim3(flatBlocks), dim3(kBlock), 0, stream, P);
	} else
		hipLaunchKernelGGL((k_connect_persistent<COUNT, STACK_LDS>), dim3(persistent_blocks(k_connect_persistent<COUNT, STACK_LDS>, maxShadow, t, numCUs)), dim3(kBlock), 0, stream, P);
}

// the stack depths that are compiled in (tyr_set_tuning TYR_TUNE_STACK_LDS_DEPTH)
#define TYR_DISPATCH_STACK(FN, COUNT, ...)          \
	switch (t.stackLdsDepth) {                      \
	case 0: FN<COUNT, 0>(__VA_ARGS__); break;       \
	case 8: FN<COUNT, 8>(__VA_ARGS__); break;       \
	case 10: FN<COUNT, 10>(__VA_ARGS__); break;     \
	case 12: FN<COUNT, 12>(__VA_ARGS__); break;     \
	case 24: FN<COUNT, 24>(__VA_ARGS__); break;     \
	default: FN<COUNT, 16>(__VA_ARGS__); break;     \
	}

void launch_extend(const FrameParams& P, uint32_t maxLive, uint32_t nSurvivors, bool countVisits, const Tuning& t, int numCUs, hipStream_t stream) {
	if (maxLive == 0)
		return;
	if (countVisits) {
		TYR_DISPATCH_STACK(launch_extend_t, true, P, maxLive, nSurvivors, t, numCUs, stream)
	} else {
		TYR_DISPATCH_STACK(launch_extend_t, false, P, maxLive, nSurvivors, t, numCUs, stream)
	}
}
void launch_shade(const FrameParams& P, uint32_t maxLive, int numCUs, hipStream_t stream) {
	if (maxLive == 0)
		return;
	const uint32_t nTiles = blocks_for(maxLive);
	// A persistent grid: as many blocks as stay resident (more would only wait for a slot and then find no tile
	// left; the tile tickets make any grid size safe).  Asked once: the occupancy query is a slow host call.
	const bool lights = (P.flags & TYR_FLAG_LIGHT_LIST) != 0; // its own instantiation: the default kernel keeps its registers
	static int perCU[2] = { 0, 0 };
	if (perCU[lights] == 0) {
		int q = 0;
		const hipError_t e = lights ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, k_shade<true>, kBlock, 0) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, k_shade<false>, kBlock, 0);
		if (e != hipSuccess || q < 1)
			q = 2;
		perCU[lights] = q > 6 ? 6 : q;
	}
	const uint32_t resident = (uint32_t)perCU[lights] * (uint32_t)numCUs;
	const dim3 grid(nTiles < resident ? nTiles : resident);
	if (lights)
		hipLaunchKernelGGL(k_shade<true>, grid, dim3(kBlock), 0, stream, P, nTiles);
	else
		hipLaunchKernelGGL(k_shade<false>, grid, dim3(kBlock), 0, stream, P, nTiles);
}
void launch_connect(const FrameParams& P, uint32_t maxShadow, bool countVisits, const Tuning& t, int numCUs, hipStream_t stream) {
	if (maxShadow == 0)
		return;
	if (countVisits) {
		TYR_DISPATCH_STACK(launch_connect_t, true, P, maxShadow, t, numCUs, stream)
	} else {
		TYR_DISPATCH_STACK(launch_connect_t, false, P, maxShadow, t, numCUs, stream)
	}
}
void launch_resolve(const float4* blit, float4* out, uint32_t nPixels, hipStream_t stream) {
	hipLaunchKernelGGL(k_resolve, dim3(blocks_for(nPixels)), dim3(kBlock), 0, stream, blit, out, nPixels);
}

} // namespace tyr
