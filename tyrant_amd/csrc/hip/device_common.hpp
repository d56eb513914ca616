// device_common.hpp -- small device helpers shared by the kernel translation units (frame.hip, shade.hip,
// traverse_flat.hip): the reference's RNG and sampling routines (kernel.cu:23-65, 181-208,
// sunsky.cu:170-185), Sphere::intersect (kernel.cu:83-105), wave-level helpers.
#pragma once

#include <hip/hip_runtime.h>

#include "detmath.hpp"
#include "kernels.hpp"
#include "sunsky.hpp"
#include "traverse.hpp"
#include "vecmath.hpp"

namespace tyr {

// The kernel's by-value argument, read where it lies: a view of the kernarg segment (the argument is the kernel's only
// one: offset 0) behind a pointer the compiler cannot see through.  Fields read through `P` are loaded in the kernel's
// first block and stay in scalar registers for its whole life (and are spilled to vector lanes when those run out); read
// through this view they are s_load-ed where they are used, from the scalar cache, and die there.  Call it INSIDE the loop
// whose body should reload.
template <class Params>
__device__ __forceinline__ const Params& kernarg_view() {
	auto p = (const __attribute__((address_space(4))) Params*)__builtin_amdgcn_kernarg_segment_ptr();
	__asm__ volatile("" : "+s"(p));
	return *(const Params*)p;
}


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

// ---- segmented queues (kernels.hpp "Queues") -------------------------------------------------------------------------
// physical slot of record j of segment seg
__device__ __forceinline__ uint32_t seg_phys(uint32_t seg, uint32_t j) { return ((((j >> 6) * kSegs) + seg) << 6) | (j & 63u); }
// slots [0, extent) cover every record of the queue (a multiple of 512; wave-uniform: eight scalar loads)
__device__ __forceinline__ uint32_t queue_extent(const uint32_t* cnt) {
	uint32_t m = 0;
#pragma unroll
	for (uint32_t w = 0; w < kSegs; ++w) {
		const uint32_t c = cnt[w * kSegStride];
		m = c > m ? c : m;
	}
	return ((m + 63u) >> 6) * (kSegs * 64u);
}
__device__ __forceinline__ uint32_t queue_records(const uint32_t* cnt) {
	uint32_t n = 0;
#pragma unroll
	for (uint32_t w = 0; w < kSegs; ++w)
		n += cnt[w * kSegStride];
	return n;
}
// does physical slot s hold a record?  (per lane: one 4-byte load, eight distinct addresses)
__device__ __forceinline__ bool slot_valid(const uint32_t* cnt, uint32_t s) { return (((s >> 9) << 6) | (s & 63u)) < cnt[((s >> 6) & (kSegs - 1u)) * kSegStride]; }
// records in the 64-slot chunk that starts at slot s0 (s0 % 64 == 0; wave-uniform when s0 is)
__device__ __forceinline__ uint32_t chunk_valid(const uint32_t* cnt, uint32_t s0) {
	const uint32_t c = cnt[((s0 >> 6) & (kSegs - 1u)) * kSegStride], first = (s0 >> 9) << 6;
	return c > first ? (c - first < 64u ? c - first : 64u) : 0u;
}
// what a slot that holds no record looks like to the traversal kernel: a ray that cannot enter any box (k_pad_holes
// writes it into the holes at the segments' ends, so that k_trace_flat hands out slots without asking)
__device__ __forceinline__ void write_dead_ray(const RayQ& q, uint32_t slot) {
	q.o_dx[slot] = make_float4(3e38f, 3e38f, 3e38f, 1.0f);
	q.dyz[slot] = make_float2(0.0f, 0.0f);
	q.hit[slot] = make_float2(0.0f, 0.0f);
}

// this iteration's virtual slot of a ray from its key
__device__ __forceinline__ uint32_t v_lookup(const VTable& T, uint32_t key) {
	const uint32_t v = key & kKeyMask;
	if (!(key & kKeyIndirect))
		return v;
	const uint32_t e = v >> 6;
	return T.blk[e >> 8] + T.pre[e] + (uint32_t)__popcll(T.word[e] & ((1ull << (v & 63u)) - 1ull));
}

// bits of a wave mask below this lane (v_mbcnt: no per-lane copy of the 64-bit "lanes below me" mask to keep in two vector registers)
__device__ __forceinline__ uint32_t lanes_below(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
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

__device__ __forceinline__ uint32_t root_ref(const DevScene& sc, const RayConst& r, float bound) {
	float t0;
	const bool ok = slab_test(r, r.nx ? sc.rootMax[0] : sc.rootMin[0], r.nx ? sc.rootMin[0] : sc.rootMax[0], r.ny ? sc.rootMax[1] : sc.rootMin[1], r.ny ? sc.rootMin[1] : sc.rootMax[1],
		r.nz ? sc.rootMax[2] : sc.rootMin[2], r.nz ? sc.rootMin[2] : sc.rootMax[2], bound, t0);
	return ok ? sc.rootRef : kRefDone;
}

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


// kernel.cu:622-625.  Adding +0 leaves the pixel unchanged, so zero terms are skipped (the reference's
// own TODO at kernel.cu:621).  One lane, one pixel: used by the per-slot and first persistent kernels (variants
// 0-1); the production kernels add a whole wave's contributions at once (accumulate_pixels_wave below).  vmcnt
// retires loads, stores and atomics in issue order and __syncthreads() waits for vmcnt(0), so where these are
// issued matters: in front of a barrier every wave sits out its own scattered atomics (~3000 cycles under load).
__device__ __forceinline__ void accumulate_pixel(float4* blit, int pixel, f3 color, int new_frame) {
	float* px = reinterpret_cast<float*>(&blit[pixel]);
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

// the flat kernels' stack (hip/traverse.hpp LdsStack): LDS column + private arrays
#define TYR_DECLARE_FLAT_STACK(st, WITH_T)                                           \
	__shared__ typename LdsStack<STACK_LDS, WITH_T>::entry_t smem_[STACK_LDS ? STACK_LDS * kBlock : 1]; \
	uint32_t spillRef_[kStackSize - STACK_LDS];                                      \
	float spillT_[(WITH_T) ? kStackSize - STACK_LDS : 1];                            \
	LdsStack<STACK_LDS, WITH_T> st;                                                  \
	st.bind(smem_ + threadIdx.x, spillRef_, spillT_);                                \
	st.reset();

static inline uint32_t blocks_for(uint32_t n) { return (n + kBlock - 1) / kBlock; }

// persistent grids: as many 256-thread blocks as stay resident (no inter-block dependency, so a
// larger grid would only queue), never more blocks than there are rays to fill them.  The occupancy query is a
// host-side call of ~0.1-0.2 ms: asked once per kernel and context (`cachedPerCU` lives in the ctx's LaunchCache),
// not once per launch -- there it sat between the pre-pass and the persistent kernel with the GPU idle.
template <class K>
static inline uint32_t persistent_blocks(K kernel, uint32_t nItems, const Tuning& t, int numCUs, int& cachedPerCU, uint32_t blockThreads = kBlock) {
	int perCU = 0;
	if (t.wavesPerSimd > 0) {
		perCU = (int)((4u * (uint32_t)t.wavesPerSimd * 64u + blockThreads - 1u) / blockThreads); // 4 SIMDs x w waves, in blocks of blockThreads / 64 waves (rounded up)
	} else {
		if (cachedPerCU == 0) {
			int q = 0;
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, kernel, (int)blockThreads, 0) != hipSuccess || q <= 0)
				q = 4;
			cachedPerCU = q;
		}
		perCU = cachedPerCU;
	}
	const uint32_t resident = (uint32_t)perCU * (uint32_t)numCUs;
	const uint32_t needed = (nItems + blockThreads - 1) / blockThreads;
	return needed < resident ? (needed ? needed : 1) : resident;
}

} // namespace tyr
