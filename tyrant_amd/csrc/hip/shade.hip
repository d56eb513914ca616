// shade.hip -- shade (kernel.cu:347-627).  A ray's random numbers are seeded from its VIRTUAL slot -- the slot the
// reference's serial ticket order gives it (kernel.cu:363, 607) -- which every record carries (kernels.hpp "Queues");
// survivors and shadow rays are appended a tile at a time to one of eight queue segments (one atomic per tile and
// queue), in whatever order the tiles finish.  Rounds 1-2 kept the serial order physically (a stable compaction with a
// decoupled look-back); that chained every tile to all tiles before it.
#include "device_common.hpp"
#include "prologue.hpp"

namespace tyr {

// ======================================================================================
// shade, kernel.cu:347-627
// ======================================================================================
struct ShadeOut {
	bool survive, shadow;
	bool tree; // a survivor that may enter the tree (class 0 of the next queue)
	f3 origin, direction, direct; // survivor state
	uint32_t flags;
	f3 sOrigin, sDir, sColor;      // shadow ray
	float sClosest;
	f3 color;                      // kernel.cu:622-625: contribution to the pixel, added at the end of the kernel
	int newFrame;
	// P.foldSpheres: the sphere half of the NEXT stage for the rays this one emits, while they are in registers (what
	// k_extend_spheres / k_connect_spheres would re-read them for: kernel.cu:127-136, 168-172)
	float2 hitRec;                 // survivor: closest sphere (distance, id | kHitSphere) or VERY_FAR
	float sBlocked;                // shadow ray: 1 = a sphere occludes it
	// P.resolveShadows: a shadow ray whose answer is known here -- a sphere occludes it, or it cannot enter the tree (it fails
	// the root box for its bound: the traversal kernel's own first test) -- is answered here and never queued
	bool sResolved, sVisible;
	// P.retireGhosts: a survivor that will hit nothing -- no sphere, not the root box -- is finished here: what the next
	// iteration's shade would do with it is fixed (kernel.cu:613-617: sky or sunsky of its direction times its throughput,
	// the path ends; no random number).  It still SURVIVES this iteration (its survive byte, the counts, its slot in the
	// next iteration's order -- every other ray's random numbers depend on that), it just never enters a queue.
	bool ghost;
};

// NEE toward spheres[6], kernel.cu:419-447 / 559-590 (common part)
struct LightSample {
	f3 toLightUnit, toLight;
	float cosAtSurface, cosAtLight;
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
	L.toLight = p - origin;
	const f3 nL = normalize(p - ld3(ls.position));
	L.toLightUnit = normalize(L.toLight);
	L.cosAtSurface = dot(normal, L.toLightUnit);
	L.cosAtLight = dot(nL, -L.toLightUnit);
	L.valid = L.cosAtSurface > 0 && L.cosAtLight > 0;
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
// the emission of an emissive triangle (its third record vector in hand): its palette entry with
// TYR_FLAG_TRIANGLE_COLORS, else the one colour of all (tyr_set_triangle_emission)
__device__ __forceinline__ f3 triangle_emission(const FrameParams& P, float4 t2) {
	if (P.flags & TYR_FLAG_TRIANGLE_COLORS) {
		const float4 e = P.palette[2u * (__float_as_uint(t2.z) & 255u) + 1u];
		return mk3(e.x, e.y, e.z);
	}
	return mk3(P.triEmission[0], P.triEmission[1], P.triEmission[2]);
}

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
			E.L.toLight = p - origin;
			const f3 nL = normalize(cr);
			E.L.toLightUnit = normalize(E.L.toLight);
			E.L.cosAtSurface = dot(normal, E.L.toLightUnit);
			E.L.cosAtLight = dot(nL, -E.L.toLightUnit);
			E.L.valid = E.L.cosAtSurface > 0 && E.L.cosAtLight > 0;
			E.emission = triangle_emission(P, t2) * pick;
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
__device__ __forceinline__ void shade_ray(const FrameParams& P, uint32_t slot, bool valid, float2 hitRecord, ShadeOut& out, uint32_t& vslotOut, AfterLoads&& afterLoads) {
	float4 a = make_float4(0.f, 0.f, 0.f, 0.f), dq = a;
	float2 b = make_float2(0.f, 0.f), h = make_float2(kVeryFar, 0.f);
	uint32_t fl = 0, key = 0;
	if (valid) {
		key = P.work.key[slot];
		a = P.work.o_dx[slot];
		b = P.work.dyz[slot];
		h = hitRecord; // loaded by the caller (an early launch reads it past the caches, once)
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
	f3 albedo = mk3(0.f, 0.f, 0.f);
	const uint32_t vslot = valid ? v_lookup(P.vPrev, key) : 0u; // the ray's slot by the serial order
	vslotOut = vslot;
	uint32_t seed = (P.frame * (uint32_t)pixel * 147565741u) * 720898027u * vslot; // kernel.cu:363
	int material = TYR_DIFF;
	out.survive = false;
	out.shadow = false;
	out.sResolved = false;
	out.sVisible = false;
	out.ghost = false;

	enum { kAtmoNone = 0, kAtmoSun, kAtmoSky, kAtmoSunSky };
	int atmo = kAtmoNone;   // what this ray wants from the atmosphere model, evaluated once for the whole wave below
	float atmoScale = 0.0f;
	const bool hit = valid && distance < kVeryFar;
	f3 normal = mk3(0.f, 0.f, 0.f);
	f3 triEmit = mk3(P.triEmission[0], P.triEmission[1], P.triEmission[2]); // what a LIGHT triangle hit head-on emits
	if (hit) {
		origin = origin + direction * distance;
		if (ident & kHitSphere) {
			const tyr_sphere& object = P.spheres[ident & 7u];
			normal = (origin - ld3(object.position)) / object.radius;
			material = object.refl;
			if (material != TYR_REFR && material != TYR_LIGHT)
				direct = direct * ld3(object.color);
			albedo = ld3(object.color);
		} else {
			// kernel.cu:380-383: normal from e1 x e2, white DIFF
			const float4 t0 = P.scene.tris[3 * ident + 0];
			const float4 t1 = P.scene.tris[3 * ident + 1];
			const float4 t2 = P.scene.tris[3 * ident + 2];
			normal = normalize(cross(mk3(t0.w, t1.x, t1.y), mk3(t1.z, t1.w, t2.x)));
			material = TYR_DIFF;
			albedo = mk3(1.f, 1.f, 1.f);
			if (P.flags & TYR_FLAG_TRIANGLE_MATERIALS) {
				const uint32_t m = __float_as_uint(t2.y);
				material = m <= (uint32_t)(LIGHTS ? TYR_LIGHT : TYR_PHONG) ? (int)m : TYR_DIFF;
			}
			if (P.flags & TYR_FLAG_TRIANGLE_COLORS) {
				// Scene.cpp:44's `tempTriangle.color`, treated like a sphere's colour (kernel.cu:375-377)
				const float4 c = P.palette[2u * (__float_as_uint(t2.z) & 255u)];
				albedo = mk3(c.x, c.y, c.z);
				if (material != TYR_REFR && material != TYR_LIGHT)
					direct = direct * albedo;
				triEmit = triangle_emission(P, t2);
			}
		}
	}
	afterLoads();
	if (hit) {
		const bool outside = dot(normal, direction) < 0;
		normal = outside ? normal : normal * -1.f;
		origin = origin + normal * kEpsilon;

		if (material == TYR_LIGHT) {
			if (lastSpecular) {
				if (LIGHTS && !(ident & kHitSphere))
					color = direct * triEmit;
				else
					color = direct * ld3(P.spheres[ident & 7u].emmission);
			} else {
				color = mk3(0.f, 0.f, 0.f);
				direct = mk3(0.f, 0.f, 0.f);
			}
		}
		lastSpecular = false;
		constexpr float kPhongPower = 40.0f;
		switch (material) {
		case TYR_LIGHT:
			break;
		case TYR_DIFF: {
			const f3 toSun = cone_sample(P.sun, seed);
			const float cosSun = dot(normal, toSun);
			if (rng_float(seed) < 0.5f) {
				if (cosSun > 0.f) {
					out.shadow = true;
					out.sOrigin = origin;
					out.sDir = toSun;
					out.sColor = 2.0f * direct; // x ((sun(toSun) * cosSun) * 1E-5f) below, kernel.cu:414
					atmo = kAtmoSun;
					atmoScale = cosSun;
					out.sClosest = 1e20f; // variables.h:41
				}
			} else {
				const EmitterSample E = sample_emitter<LIGHTS>(P, seed, origin, normal);
				const LightSample& L = E.L;
				if (L.valid) {
					const float reach = length(L.toLight);
					const float subtended = (L.cosAtLight * E.area) / dot(L.toLight, L.toLight);
					out.shadow = true;
					out.sOrigin = origin;
					out.sDir = L.toLightUnit;
					out.sColor = ((((E.emission * 2.0f) * direct) * subtended) * kInvPi) * L.cosAtSurface;
					out.sClosest = reach;
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
				const f3 e = (-albedo) * distance;
				direct = direct * mk3(dm::expf_det(e.x), dm::expf_det(e.y), dm::expf_det(e.z));
			}
			break;
		}
		case TYR_PHONG: {
			f3 w, u, v, d;
			do {
				const float phi = 2 * kPi * rng_float(seed);
				const float r2 = rng_float(seed);
				const float cosLobe = dm::powf_det(1.0f - r2, 1.0f / (kPhongPower + 1.0f));
				const float sinLobe = sqrtf(1.0f - cosLobe * cosLobe);
				w = direction - (normal * 2.0f) * dot(normal, direction);
				w = normalize(w);
				orthonormal_basis_naive(w, u, v);
				float sp, cp;
				dm::sincosf_det(phi, sp, cp);
				d = (u * cp) * sinLobe + (v * sp) * sinLobe + w * cosLobe;
				d = normalize(d);
			} while (dot(d, normal) <= kEpsilon);

			const f3 toSun = cone_sample(P.sun, seed);
			float cosSun = dot(normal, toSun);
			if (rng_float(seed) < 0.5f) {
				if (cosSun > 0.f) {
					const float lobeCos = dot(toSun, w);
					if (lobeCos > kEpsilon) {
						cosSun *= dm::powf_det(lobeCos, kPhongPower);
						out.shadow = true;
						out.sOrigin = origin;
						out.sDir = toSun;
						out.sColor = (2.0f * direct) * ((kPhongPower + 2) * 0.5f * kInvPi); // x ((sun(..) * cosSun) * 1E-5f) below
						atmo = kAtmoSun;
						atmoScale = cosSun;
						out.sClosest = 1e20f;
					}
				}
			} else {
				const EmitterSample E = sample_emitter<LIGHTS>(P, seed, origin, normal);
				const LightSample& L = E.L;
				if (L.valid) {
					float lobeCos = dot(L.toLightUnit, w);
					if (lobeCos > kEpsilon) {
						lobeCos = dm::powf_det(lobeCos, kPhongPower);
						const float reach = length(L.toLight);
						const float subtended = (L.cosAtLight * E.area) / dot(L.toLight, L.toLight);
						f3 sc = (E.emission * 2.0f) * direct;
						sc = sc * subtended;
						sc = sc * (kPhongPower + 2);
						sc = sc * 0.5f;
						sc = sc * kInvPi;
						sc = sc * lobeCos;
						sc = sc * L.cosAtSurface;
						out.shadow = true;
						out.sOrigin = origin;
						out.sDir = L.toLightUnit;
						out.sColor = sc;
						out.sClosest = reach;
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
	// a diffuse or Phong hit whose next-event sample went to the sun (sun(toSun), kernel.cu:414 / 553), and a
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
			// the traversal's own first test on the new ray: a ray that fails it can never enter the tree and goes to class 1.
			// Without the sphere pre-pass's distance as bound when a pre-pass follows (a bound only takes rays away); with it
			// when this kernel does the pre-pass's work itself (k_primary's form: same function, same bound, same answer
			// as root_ref at the traversal's refill)
			float bound = kVeryFar;
			if (P.foldSpheres) {
				out.hitRec = sphere_hit_record(P, origin, direction);
				bound = out.hitRec.x;
			}
			out.tree = P.scene.rootRef != kRefDone && root_ref(P.scene, make_ray(origin, direction), bound) != kRefDone;
			out.origin = origin;
			out.direction = direction;
			out.direct = direct;
			out.flags = (uint32_t)bounces | ((lastSpecular ? 1u : 0u) << 8);
			out.ghost = P.retireGhosts != 0u && !out.tree && !(out.hitRec.x < kVeryFar); // (retireGhosts implies foldSpheres: hitRec is the sphere answer)
		} else {
			new_frame++;
		}
	} else if (valid) {
		new_frame++;
	}
	if (out.ghost) {
		// the next iteration's shade of this ray, now (kernel.cu:613-617 with its lastSpecular, its throughput): one more
		// evaluation of the atmosphere for the lanes that need it; the pixel receives this iteration's and the next one's
		// contribution as one sum, and the path is finished
		f3 seen;
		if (lastSpecular && P.sun.sunAngularDiameterCos == 1.0f) {
			seen = mk3(1.0f, 0.0f, 0.0f); // sunsky.cu:118-119
		} else {
			const Atmosphere a = atmosphere(P.sun, direction);
			seen = lastSpecular ? sunsky_radiance(P.sun, a) : sky_radiance(a);
		}
		color = color + direct * seen;
		new_frame++;
	}

	out.sBlocked = 0.0f;
	if (P.foldSpheres && out.shadow) { // kernel.cu:168-172 (k_connect_spheres' test, on the ray in registers)
		bool occluded = false;
#pragma unroll
		for (int i = TYR_NUM_SPHERES; i--;) {
			const float t = sphere_intersect(P.spheres[i], out.sOrigin, out.sDir);
			occluded = occluded || (t && (t + kEpsilon) < out.sClosest);
		}
		out.sBlocked = occluded ? 1.0f : 0.0f;
		if (P.resolveShadows) {
			// connect for this ray (kernel.cu:630-646) comes down to what is known now: blocked by a sphere -> nothing; not
			// blocked and unable to enter the tree -> visible, its colour goes to the pixel (kernel.cu:640-644) with this
			// ray's own contribution; only a ray that may meet a triangle is left to the traversal kernel
			const bool mayEnter = P.scene.rootRef != kRefDone && root_ref(P.scene, make_ray(out.sOrigin, out.sDir), out.sClosest) != kRefDone;
			if (occluded || !mayEnter) {
				out.shadow = false;
				out.sResolved = true;
				out.sVisible = !occluded;
				if (!occluded)
					color = color + out.sColor;
			}
		}
	}
	out.color = color;
	out.newFrame = new_frame;
}

struct ShadeStage { // one tile's output, waiting for the tile's place in the queues: 25 KB
	float4 sv_o_dx[kBlock];
	float2 sv_dyz[kBlock];
	float4 sv_direct_ix[kBlock];
	uint32_t sv_flags[kBlock];
	uint32_t sv_key[kBlock];
	float2 sv_hit[kBlock];   // (P.foldSpheres) the survivor's sphere record
	float4 sh_o_dx[kBlock];
	float4 sh_dyz_cd_ix[kBlock];
	float4 sh_color[kBlock];
	uint32_t sh_key[kBlock];
};

#define TYR_SHADE_BLOCKS_PER_CU 5 // tiles in flight per CU: 102 vector registers each (96 used; 5 x 27.8 KB of the CU's 160 KB LDS)
// One tile = 256 consecutive physical slots of the work queue = four 64-slot chunks (one per wave), each the tail or
// the middle of one segment: a wave's valid lanes are the first chunk_valid() of its chunk.
//
// Tiles are drawn from eight tickets (word w hands out tiles w, w + 8, ...; a block starts at word blockIdx % 8 and
// moves on when a word is used up): one word per tile id would be a single ticket (88 draws/us: 0.74 ms for the 64.8 k
// tiles of a full queue).  Nothing waits for anything here: any grid size is safe.
//
// A tile's survivors (kernel.cu:607-608) and shadow rays (kernel.cu:416-417 ...) get their place with ONE atomic per
// queue on the counter of segment (tile / 2) % 8, issued as soon as the tile's counts are known; the records wait in LDS at
// their rank inside the tile and leave as coalesced stores (thread t writes record t) AFTER the next tile has been
// shaded -- by then the atomics have long returned.
template <bool LIGHTS>
__global__ void __launch_bounds__(kBlock, TYR_SHADE_BLOCKS_PER_CU) k_shade(const FrameParams P_) {
	const FrameParams& P = P_;
	__shared__ uint32_t sh[32];
	__shared__ ShadeStage stage;
	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63u, wave = tid >> 6;
	// class 0's tiles, then class 1's (from the device's counts: the host may have sized the grid from an upper bound)
	const uint32_t tiles0 = queue_extent(P.segWork) / kBlock;
	const uint32_t nTiles = tiles0 + queue_extent(P.segWork + kClassWords) / kBlock;
#ifdef TYR_SHADE_TIMING
	// diagnostic build: where a tile's time goes, in s_memtime ticks summed over this block's tiles (thread 0;
	// debug[0] shade, [1] ranks + barrier, [2] place, [4] copy out + barrier, [5] stage + pixel atomics, [7] tiles)
	unsigned long long tacc_[6] = { 0, 0, 0, 0, 0, 0 }, t_ = __builtin_amdgcn_s_memtime(), ntiles_ = 0;
#define TYR_STAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc_[i] += now_ - t_; t_ = now_; }
#else
#define TYR_STAMP(i)
#endif
	bool havePrev = false;            // a shaded tile whose records wait in `stage`
	uint32_t pendPixel = 0;           // this lane's pixel contribution of that tile, not yet added
	int pendNew = 0;
	f3 pendColor = mk3(0.f, 0.f, 0.f);
	uint32_t prevSeg = 0, prevS = 0, prevT = 0, prevH = 0;
	uint32_t mySurvivors = 0, myShadows = 0; // thread 0: what this block appended
	[[maybe_unused]] bool lastBlock = false;  // thread 0: this block finished last (the kernel's end)
	[[maybe_unused]] uint32_t survivors = 0;  // ... and the iteration's survivor count it read there
	[[maybe_unused]] uint32_t snapShadows = 0;
	uint32_t waveGhosts = 0; // (wave-uniform) survivors finished in place (P.retireGhosts): they count as survivors
	uint32_t waveResolved = 0, waveVisible = 0; // (wave-uniform) shadow rays answered in place (P.resolveShadows): they count as emitted, the visible ones as visible

	// finish the waiting tile: its places have arrived (sh[12], sh[13]); move its records from LDS to the queues
	auto flush_prev = [&]() {
		const FrameParams& P = kernarg_view<FrameParams>();
		const uint32_t baseT = sh[12], baseH = sh[13], baseK = sh[18];
		TYR_STAMP(2)
		// one array at a time (the compiler barrier keeps it from loading all records first): this copy is where the
		// kernel's register count peaks
		{
			const bool isTree = tid < prevT;
			const uint32_t base = isTree ? baseT : baseK;
			if (tid < prevS && base != 0xffffffffu) {
				const uint32_t d = isTree ? seg_phys(prevSeg, base + tid) : P.classStride + seg_phys(prevSeg, base + (tid - prevT));
				P.next.o_dx[d] = stage.sv_o_dx[tid];
				__asm__ volatile("" ::: "memory");
				P.next.direct_ix[d] = stage.sv_direct_ix[tid];
				__asm__ volatile("" ::: "memory");
				P.next.dyz[d] = stage.sv_dyz[tid];
				if (P.foldSpheres)
					P.next.hit[d] = stage.sv_hit[tid];
				P.next.flags[d] = stage.sv_flags[tid];
				P.next.key[d] = stage.sv_key[tid];
			}
		}
		__asm__ volatile("" ::: "memory");
		if (tid < prevH && baseH != 0xffffffffu) {
			const uint32_t d = seg_phys(prevSeg, baseH + tid);
			P.shadow.o_dx[d] = stage.sh_o_dx[tid];
			__asm__ volatile("" ::: "memory");
			P.shadow.dyz_cd_ix[d] = stage.sh_dyz_cd_ix[tid];
			__asm__ volatile("" ::: "memory");
			P.shadow.color[d] = stage.sh_color[tid];
			P.shadow.key[d] = stage.sh_key[tid];
		}
		__syncthreads(); // `stage` and sh[] are free again
		TYR_STAMP(4)
	};

	uint32_t word = blockIdx.x % kTicketWords, tried = 0;
	uint32_t* const tickets = P.k->shade_tiles;
	auto draw_tile = [&]() -> uint32_t { // block-uniform; nTiles when nothing is left
		uint32_t vbNext = nTiles;
		if (tid == 0) {
			while (tried < kTicketWords) {
				const uint32_t t = atomicAdd(&tickets[word * 32], 1u);
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
	// (no tile at all -- a launch for an iteration without rays: no ticket is drawn, eight round trips saved per block)
	for (uint32_t vb = nTiles != 0u ? draw_tile() : nTiles; vb < nTiles; vb = draw_tile()) { // vb = tile id = 256 physical slots of one class
		TYR_STAMP(3) // (the draw: ticket)
		const FrameParams& P = kernarg_view<FrameParams>(); // this tile's reads of the arguments: loaded where they are used (device_common.hpp; 113 scalar spills -> 2, 128 vector registers -> 96)
		const uint32_t cls = vb >= tiles0 ? 1u : 0u;
		const uint32_t inClass = (vb - cls * tiles0) * kBlock + tid; // slot inside the class
		const uint32_t slot = cls * P.classStride + inClass;
		ShadeOut out; // (not zeroed: every field is written before the flag that admits its reading is set)
		uint32_t pixelBits = 0, vslot = 0;
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
		const bool valid = lane < chunk_valid(P.segWork + cls * kClassWords, inClass & ~63u);
		float2 hitRecord = make_float2(kVeryFar, 0.f);
		if (valid)
			hitRecord = P.work.hit[slot];
		if (valid)
			pixelBits = __float_as_uint(P.work.direct_ix[slot].w);
		shade_ray<LIGHTS>(P, slot, valid, hitRecord, out, vslot, flush_pixels);
		if (valid)
			P.survFlag[vslot] = out.survive ? 1 : 0; // what k_scan_words turns into next iteration's slots
		TYR_STAMP(0)

		// ---- ranks inside the tile: survivors that may enter the tree, survivors that cannot, shadow rays ----
		const bool sT = out.survive && out.tree, sS = out.survive && !out.tree && !out.ghost;
		if (P.retireGhosts)
			waveGhosts += (uint32_t)__popcll(__ballot(out.ghost));
		const unsigned long long bt = __ballot(sT), bk = __ballot(sS);
		const unsigned long long bh = __ballot(out.shadow);
		if (P.resolveShadows) {
			waveResolved += (uint32_t)__popcll(__ballot(out.sResolved));
			waveVisible += (uint32_t)__popcll(__ballot(out.sVisible));
		}
		const uint32_t rt = lanes_below(bt), rk = lanes_below(bk), rh = lanes_below(bh);
		if (lane == 0) {
			sh[4 + wave] = __popcll(bt);
			sh[8 + wave] = __popcll(bh);
			sh[14 + wave] = __popcll(bk);
		}
		__syncthreads();
		uint32_t wt = 0, wk = 0, wh = 0, totT = 0, totK = 0, totH = 0;
#pragma unroll
		for (uint32_t w = 0; w < kBlock / 64; ++w) {
			const uint32_t ct = sh[4 + w], ch = sh[8 + w], ck = sh[14 + w];
			if (w < wave) {
				wt += ct;
				wk += ck;
				wh += ch;
			}
			totT += ct;
			totK += ck;
			totH += ch;
		}
		totT = (uint32_t)__builtin_amdgcn_readfirstlane((int)totT); // block-uniform: keep them out of the vector registers
		totK = (uint32_t)__builtin_amdgcn_readfirstlane((int)totK);
		totH = (uint32_t)__builtin_amdgcn_readfirstlane((int)totH);
		const uint32_t totS = totT + totK;
		TYR_STAMP(1)
		if (havePrev)
			flush_prev(); // ends with a barrier: sh[4..13] have been read by every thread
		else
			__syncthreads();
		// tiles 2k and 2k + 1 hold records [64k, 64k + 64) of all eight segments between them and both append to segment
		// k % 8: whichever rays survive, a segment receives at most an eighth of the queue's records (+ 64 per segment)
		const uint32_t seg = (vb >> 1) & (kSegs - 1u);
		if (tid == 0) {
			// this tile's places: consumed by flush_prev one tile later
			uint32_t bT = 0, bK = 0, bH = 0;
			if (totT) {
				bT = atomicAdd(&P.segNext[seg * kSegStride], totT);
				if (bT + totT > P.segCap)
					bT = 0xffffffffu;
			}
			if (totK) {
				bK = atomicAdd(&P.segNext[kClassWords + seg * kSegStride], totK);
				if (bK + totK > P.segCap)
					bK = 0xffffffffu;
			}
			if (totH) {
				bH = atomicAdd(&P.kc->seg[seg * kSegStride], totH);
				if (bH + totH > P.segCap)
					bH = 0xffffffffu;
			}
			if (bT == 0xffffffffu || bK == 0xffffffffu || bH == 0xffffffffu)
				atomicOr(&P.k->device_error, kErrQueueOverflow);
			sh[12] = bT;
			sh[13] = bH;
			sh[18] = bK;
			mySurvivors += (bT == 0xffffffffu ? 0u : totT) + (bK == 0xffffffffu ? 0u : totK);
			myShadows += bH == 0xffffffffu ? 0u : totH;
		}
		if (out.survive && !out.ghost) {
			const uint32_t k = out.tree ? wt + rt : totT + wk + rk; // the tile's class-0 survivors first, then its class-1 ones
			stage.sv_o_dx[k] = make_float4(out.origin.x, out.origin.y, out.origin.z, out.direction.x);
			stage.sv_dyz[k] = make_float2(out.direction.y, out.direction.z);
			stage.sv_direct_ix[k] = make_float4(out.direct.x, out.direct.y, out.direct.z, __uint_as_float(pixelBits));
			stage.sv_flags[k] = out.flags;
			stage.sv_key[k] = vslot | kKeyIndirect; // next iteration's slot = rank of vslot among this iteration's survivors
			if (P.foldSpheres)
				stage.sv_hit[k] = out.hitRec;
		}
		if (out.shadow) {
			const uint32_t k = wh + rh;
			stage.sh_o_dx[k] = make_float4(out.sOrigin.x, out.sOrigin.y, out.sOrigin.z, out.sDir.x);
			stage.sh_dyz_cd_ix[k] = make_float4(out.sDir.y, out.sDir.z, out.sClosest, __uint_as_float(pixelBits));
			stage.sh_color[k] = make_float4(out.sColor.x, out.sColor.y, out.sColor.z, out.sBlocked); // .w: the connect pre-pass's verdict (0 when a pre-pass follows and writes it)
			stage.sh_key[k] = vslot;
		}
		havePrev = true;
		prevSeg = seg;
		prevS = totS;
		prevT = totT;
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
	const FrameParams& PE = kernarg_view<FrameParams>(); // the kernel's end: its reads of the arguments, not held through the loop
	accumulate_pixels_wave(PE.blit, (int)pendPixel, pendColor, pendNew);
	if (havePrev) {
		__syncthreads();
		flush_prev();
	}
	uint32_t myResolved = 0;
	if (PE.resolveShadows || PE.retireGhosts) { // (block-uniform)
		if (lane == 0) {
			sh[20 + wave] = waveResolved;
			sh[24 + wave] = waveVisible;
			sh[28 + wave] = waveGhosts;
		}
		__syncthreads();
		if (tid == 0) {
			myResolved = sh[20] + sh[21] + sh[22] + sh[23];
			const uint32_t vis = sh[24] + sh[25] + sh[26] + sh[27];
			if (vis)
				atomicAdd(&PE.k->n_shadow_visible, (unsigned long long)vis);
			mySurvivors += sh[28] + sh[29] + sh[30] + sh[31]; // survivors all the same: the next iteration's ray count, the totals
		}
	}
	// kernel.cu:607 / 416: the totals the next top-up and connect read.  Every block adds what it appended; the block
	// that finishes last publishes the per-iteration figures.
#ifdef TYR_SHADE_TIMING
	if (tid == 0) {
		for (int i = 0; i < 6; ++i)
			atomicAdd(&PE.k->debug[i], tacc_[i]);
		atomicAdd(&PE.k->debug[7], ntiles_);
	}
#endif
	if (tid == 0) {
		if (mySurvivors)
			atomicAdd(&PE.k->primary_ray_cnt, mySurvivors);
		if (myShadows + myResolved)
			atomicAdd(&PE.kc->shadow_cnt, myShadows + myResolved);
		// the counts above must have arrived before this block counts as done; they are device-scope atomics, so waiting for
		// them is enough (a release fence here is an L2 write-back per block)
		__asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
		if (atomicAdd(&PE.k->shade_blocks_done, 1u) + 1u == PE.shadeBlocks) {
			const uint32_t s = __hip_atomic_load(&PE.k->primary_ray_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const uint32_t h = __hip_atomic_load(&PE.kc->shadow_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			PE.k->shadow_ray_cnt = h;
			PE.k->total_shadow_rays += h;
			PE.k->n_survive += s;
			for (uint32_t c = 0; c < kClasses; ++c) // what the next iteration's sphere pre-pass has to do (a top-up appends behind it)
				for (uint32_t w = 0; w < kSegs; ++w)
					PE.k->segSurv[c][w] = __hip_atomic_load(&PE.segNext[c * kClassWords + w * kSegStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			lastBlock = true;
			survivors = s;
			snapShadows = h;
		}
	}
	// P.shadeOpensNext (tyr_render one iteration ahead, the next iteration known to come without a top-up; TYR_TUNE_SCAN_IN_TRACE): the
	// block that finishes last opens that iteration -- the hole padding and set_wavefront_globals, what P.foldNextPrologue has
	// k_scan_words' last block do -- so that the scan of THIS iteration's survive bytes, which only the next SHADE launch reads, needs no
	// launch in front of the next traversal launch: that launch's waves do it on their way in (hip/scan_wave.hpp).  Everything read here was written by
	// agent-scope atomics (the counters) or by this thread; the scan finds its ray count in scan_live[] (n_live is reset below).
	if (PE.shadeOpensNext != 0u) { // (wave-uniform: a kernel argument)
		__syncthreads(); // sh[] is free: every wave has left the tile loop
		if (tid == 0) {
			// (an iteration without rays opens nothing: it is the one a run-ahead render queued behind its last real iteration, the host
			// never queues a successor behind it, and the counters of the iteration before -- kcPrev's segments and shadow count, which
			// set_wavefront_globals would zero -- are what tyr_shadow_export hands out after the render; k_scan_words' fold skips it likewise)
			sh[16] = (lastBlock && PE.k->n_live != 0u) ? 1u : 0u;
			sh[17] = survivors;
		}
		__syncthreads();
		if (sh[16] != 0u) {
			if (tid < kSegs)
				sh[tid] = __hip_atomic_load(&PE.segNext[tid * kSegStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // class 0 of the next work queue
			else if (tid < 2u * kSegs)
				sh[tid] = __hip_atomic_load(&PE.kc->seg[(tid - kSegs) * kSegStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // this iteration's shadow queue
			if (tid == 0)
				PE.k->scan_live[PE.scanSet & 1u] = PE.k->n_live;
			__syncthreads();
			pad_work_holes_counts(PE.next, sh);
			pad_shadow_holes_counts(PE.shadow, sh + kSegs);
			__syncthreads();
			wavefront_globals_for(PE, PE.segWork, PE.kcPrev, true, sh[17]);
		}
	}
	// the counts the host's render loop waits for, straight into its (pinned) memory: four stores, a fence, the stamp
	if (tid == 0 && lastBlock && PE.hostSnap != nullptr) {
		HostSnap* const hs = PE.hostSnap;
		__hip_atomic_store(&hs->survivors, survivors, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		__hip_atomic_store(&hs->shadows, snapShadows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		__hip_atomic_store(&hs->device_error, __hip_atomic_load(&PE.k->device_error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		__hip_atomic_store(&hs->seq, PE.snapSeq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
#undef TYR_STAMP
}

// blocks of a shade launch over (at most) maxSlots physical slots
uint32_t shade_grid(const FrameParams& P, uint32_t maxSlots, int numCUs, LaunchCache& lc) {
	const uint32_t nTiles = blocks_for(maxSlots); // an upper bound is fine: the kernel takes the tile count from the device
	// A persistent grid: as many blocks as stay resident (more would only wait for a slot and then find no tile
	// left; the tile tickets make any grid size safe).  Asked once: the occupancy query is a slow host call.
	const bool lights = (P.flags & TYR_FLAG_LIGHT_LIST) != 0; // its own instantiation: the default kernel keeps its registers
	int* perCU = lc.perCU[kLcShade]; // [0] default kernel, [1] the light-list instantiation
	if (perCU[lights] == 0) {
		int q = 0;
		const hipError_t e = lights ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, k_shade<true>, kBlock, 0) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, k_shade<false>, kBlock, 0);
		if (e != hipSuccess || q < 1)
			q = 2;
		perCU[lights] = q > 6 ? 6 : q;
	}
	const uint32_t resident = (uint32_t)perCU[lights] * (uint32_t)numCUs;
	return nTiles < resident ? (nTiles ? nTiles : 1u) : resident;
}
void launch_shade(const FrameParams& P0, uint32_t maxSlots, int numCUs, LaunchCache& lc, hipStream_t stream) {
	FrameParams P = P0;
	P.shadeBlocks = shade_grid(P, maxSlots, numCUs, lc);
	if (P.flags & TYR_FLAG_LIGHT_LIST)
		hipLaunchKernelGGL((k_shade<true>), dim3(P.shadeBlocks), dim3(kBlock), 0, stream, P);
	else
		hipLaunchKernelGGL((k_shade<false>), dim3(P.shadeBlocks), dim3(kBlock), 0, stream, P);
}

} // namespace tyr
