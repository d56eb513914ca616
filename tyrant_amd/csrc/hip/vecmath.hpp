// vecmath.hpp -- float3 arithmetic for host and gfx950 device code, in the evaluation
// order of the glm functions the reference calls (Dependencies/glm-0.9.9.3/detail):
//   dot        func_geometric.inl:54-61    (x + y) + z
//   cross      func_geometric.inl:74-85
//   normalize  func_geometric.inl:88-96    v * (1 / sqrt(dot(v, v)))
//   reflect    func_geometric.inl:110-116  I - N * dot(N, I) * 2
//   min / max  func_common.inl:16-29
//   clamp      func_common.inl:566         min(max(x, lo), hi)
//   mix        func_common.inl:103-111     x + a * (y - x)
//   smoothstep func_common.inl:257-265     t = clamp((x - e0) / (e1 - e0), 0, 1); t * t * (3 - 2 * t)
// Pinned to the vendored glm itself: tests/golden/ref_glm.npz (oracle/ref_harness.cpp ref_glm) through tyr_vecmath_probe.
// Every operation is one IEEE binary32 op; the library is built with -ffp-contract=off
// so nothing is fused (DESIGN.md "Numeric contract").
#pragma once

#include <hip/hip_runtime.h>

// one IEEE operation per source operation, also if a build forgets -ffp-contract=off
#pragma clang fp contract(off)

#define TYR_HD __host__ __device__ __forceinline__

namespace tyr {

struct f3 {
	float x, y, z;
};

TYR_HD f3 mk3(float x, float y, float z) { return f3{ x, y, z }; }
TYR_HD f3 ld3(const float* p) { return f3{ p[0], p[1], p[2] }; }
TYR_HD f3 operator+(f3 a, f3 b) { return f3{ a.x + b.x, a.y + b.y, a.z + b.z }; }
TYR_HD f3 operator-(f3 a, f3 b) { return f3{ a.x - b.x, a.y - b.y, a.z - b.z }; }
TYR_HD f3 operator*(f3 a, f3 b) { return f3{ a.x * b.x, a.y * b.y, a.z * b.z }; }
TYR_HD f3 operator/(f3 a, f3 b) { return f3{ a.x / b.x, a.y / b.y, a.z / b.z }; }
TYR_HD f3 operator*(f3 a, float s) { return f3{ a.x * s, a.y * s, a.z * s }; }
TYR_HD f3 operator*(float s, f3 a) { return f3{ s * a.x, s * a.y, s * a.z }; }
TYR_HD f3 operator/(f3 a, float s) { return f3{ a.x / s, a.y / s, a.z / s }; }
TYR_HD f3 operator-(f3 a) { return f3{ -a.x, -a.y, -a.z }; }

TYR_HD float dot(f3 a, f3 b) {
	const float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
	return tx + ty + tz;
}
TYR_HD f3 cross(f3 a, f3 b) { return f3{ a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y }; }
TYR_HD float rsqrt_glm(float x) { return 1.0f / sqrtf(x); }
TYR_HD f3 normalize(f3 v) { return v * rsqrt_glm(dot(v, v)); }
TYR_HD float length(f3 v) { return sqrtf(dot(v, v)); }
TYR_HD f3 reflect(f3 I, f3 N) { return I - N * dot(N, I) * 2.0f; }
TYR_HD float gmin(float x, float y) { return (y < x) ? y : x; }
TYR_HD float gmax(float x, float y) { return (x < y) ? y : x; }
TYR_HD float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
TYR_HD f3 gmix(f3 x, f3 y, float a) { return x + a * (y - x); }
TYR_HD float gsmoothstep(float edge0, float edge1, float x) {
	const float t = gclamp((x - edge0) / (edge1 - edge0), 0.0f, 1.0f);
	return t * t * (3.0f - 2.0f * t);
}

constexpr float kPi = 3.1415926535897932f; // variables.h:3
constexpr float kInvPi = 1.0f / kPi;        // variables.h:4
constexpr float kEpsilon = 0.001f;          // variables.h:14
constexpr float kVeryFar = 1e20f;           // kernel.cu:15
constexpr int kMaxBounces = 5;              // kernel.cu:16

} // namespace tyr
