// sunsky.hpp -- analytic sun / sky radiance and sun-cone sampling on the device
// (reference: sunsky.cu:10-185, sunsky.cuh:26-43).
//
// Hoisting: the reference recomputes, per call and per thread, values that depend only
// on the sun direction -- SunIntensity (acos + exp), totalMie (three pows), the mix
// factor (pow), and the cone basis (three normalisations).  Here they are computed once
// per sun change on the host (host/sun_setup.cpp) and arrive as kernel-argument
// constants in SGPRs; per call only the view-dependent part remains: three exps, the
// two phase functions (binary64 where the reference's unsuffixed literals promote,
// sunsky.cu:11, 21) and three square roots.
#pragma once

#include "detmath.hpp"
#include "vecmath.hpp"

namespace tyr {

// sunsky.cu:4-8 device globals + per-sun-change constants
struct SunParams {
	float sunDirection[3];
	float sunAngularDiameterCos;
	float sunE;             // SunIntensity(dot(sunDirection, up))          sunsky.cu:24-26
	float rayleighAtX[3];   //                                               sunsky.cu:41
	float mieAtX[3];        // totalMie(...) * mieCoefficient                sunsky.cu:15-19, 44
	float totalLightAtX[3]; // rayleighAtX + mieAtX                          sunsky.cu:63
	float mixFactor;        // clamp(pow(1 - dot(up, sunDirection), 5), 0, 1) sunsky.cu:66-67
	float coneDir[3];       // normalize(sunDirection)                       sunsky.cu:173
	float coneO1[3];        // normalize(ortho(dir))                         sunsky.cu:174
	float coneO2[3];        // normalize(cross(dir, o1))                     sunsky.cu:175
	float coneExtent;       // 1 - sunAngularDiameterCos                     kernel.cu:410
};

#ifdef __HIPCC__

constexpr float kRayleighZenithLength = 8.4E3f; // sunsky.cuh:38
constexpr float kMieZenithLength = 1.25E3f;     // sunsky.cuh:39
constexpr float kMieDirectionalG = 0.80f;       // sunsky.cuh:33

struct Atmosphere {
	float cosViewSunAngle;
	f3 Fex;
	f3 sky;
};

// the block shared by sun(), sky() and sunsky(): sunsky.cu:33-67 / 77-111 / 117-152
__device__ __forceinline__ Atmosphere atmosphere(const SunParams& S, f3 viewDir) {
	const f3 up = mk3(0.0f, 0.0f, 1.0f); // sunsky.cu:5
	const f3 sunDirection = ld3(S.sunDirection);
	const f3 rayleighAtX = ld3(S.rayleighAtX);
	const f3 mieAtX = ld3(S.mieAtX);

	const float cosViewSunAngle = dot(viewDir, sunDirection);
	const float cosZenith = dot(up, viewDir);
	const float cosZenithPos = gmax(0.0f, cosZenith);
	const float pathRayleigh = kRayleighZenithLength / cosZenithPos;
	const float pathMie = kMieZenithLength / cosZenithPos;

	const f3 ext = rayleighAtX * pathRayleigh + mieAtX * pathMie;
	const f3 Fex = mk3(dm::expf_det(-ext.x), dm::expf_det(-ext.y), dm::expf_det(-ext.z));

	// RayleighPhase, sunsky.cu:10-12: (3.0 / (16.0 * pi)) * (1.0 + powf(c, 2.0))
	const float c2 = cosViewSunAngle * cosViewSunAngle;
	const float rayleighPhase = (float)((3.0 / (16.0 * (double)kPi)) * (1.0 + (double)c2));
	// hgPhase, sunsky.cu:20-22: pow(x, 1.5) is x * sqrt(x) in binary64
	const float g2 = kMieDirectionalG * kMieDirectionalG;
	const double hb = 1.0 - 2.0 * (double)kMieDirectionalG * (double)cosViewSunAngle + (double)g2;
	const double hp = hb * sqrt(hb);
	const float hg = (float)((1.0 / (4.0 * (double)kPi)) * ((1.0 - (double)g2) / hp));

	const f3 phased = rayleighAtX * rayleighPhase + mieAtX * hg;
	const f3 inscatter = S.sunE * (phased / ld3(S.totalLightAtX));

	f3 sky = inscatter * mk3(1.0f - Fex.x, 1.0f - Fex.y, 1.0f - Fex.z);
	// mix(vec3(1), pow(inscatter * Fex, vec3(0.5)), a) = 1 + a * (y - 1)  (func_common.inl:103-111)
	const f3 sf = inscatter * Fex;
	sky = sky * gmix(mk3(1.0f, 1.0f, 1.0f), mk3(sqrtf(sf.x), sqrtf(sf.y), sqrtf(sf.z)), S.mixFactor);
	return Atmosphere{ cosViewSunAngle, Fex, sky };
}

// sunsky.cu:32-74.  Line 70's `adc < (cos ? 1.0 : 0.0)` precedence quirk is kept: the
// disk term is 1 for every non-zero cosine.  (These take the shared block as an argument: k_shade evaluates it once
// per ray, for whichever of the three a ray needs.)
__device__ __forceinline__ f3 sun_radiance(const SunParams& S, const Atmosphere& a) {
	const float sundisk = ((double)S.sunAngularDiameterCos < (a.cosViewSunAngle ? 1.0 : 0.0)) ? 1.0f : 0.0f;
	const f3 sun = ((S.sunE * 19000.0f) * a.Fex) * sundisk;
	return 0.01f * sun;
}

// sunsky.cu:76-114
__device__ __forceinline__ f3 sky_radiance(const Atmosphere& a) { return (1.f * 0.01f) * a.sky; }

// sunsky.cu:116-161; the caller handles the early return of sunsky.cu:118-119 (sunAngularDiameterCos == 1)
__device__ __forceinline__ f3 sunsky_radiance(const SunParams& S, const Atmosphere& a) {
	const float e0 = S.sunAngularDiameterCos;
	const float e1 = S.sunAngularDiameterCos + 0.00002f;
	const float sundisk = gsmoothstep(e0, e1, a.cosViewSunAngle);
	const f3 sun = (((S.sunE * 19000.0f) * a.Fex) * sundisk) * 1E-5f;
	return 0.01f * (sun + a.sky);
}

#endif // __HIPCC__

} // namespace tyr
