// sun_setup.cpp -- per-sun-change constants for the device sun/sky model.
//
// Reference: the host prologue of launch_kernels (kernel.cu:683-684, 704-709) plus the
// sun-direction-only sub-expressions of sunsky.cu that the reference re-evaluates per
// call on the device: SunIntensity (sunsky.cu:24-26), totalMie (15-19), the mix factor
// (66-67) and getConeSample's basis (170-175).  They run here once per sun change, on
// the host, with <cmath> in the precision the reference's expressions have.
#include <cmath>

#include "host.hpp"

namespace tyr {

namespace {
// sunsky.cuh:26-43
constexpr float sunSize = 1.5f;
constexpr float cutoffAngle = kPi / 1.95f;
constexpr float steepness = 1.5f;
constexpr float turbidity = 1.0f;
constexpr float mieCoefficient = 0.005f;
constexpr float v = 4.0f;
constexpr float sunIntensity = 1000.0f;

// sunsky.cu:24-26: acos/exp resolve to the float overloads under nvcc; 1.0 - x and max(0.0, x) are binary64
float SunIntensity(float zenithAngleCos) {
	const float e = std::exp(-((cutoffAngle - std::acos(zenithAngleCos)) / steepness));
	const double m = 1.0 - static_cast<double>(e);
	return static_cast<float>(static_cast<double>(sunIntensity) * ((0.0 < m) ? m : 0.0));
}

// sunsky.cu:15-19
f3 totalMie(f3 primaryWavelengths, f3 K, float T) {
	const float c = static_cast<float>((0.2 * static_cast<double>(T)) * 10E-18);
	const float s = 0.434f * c * kPi;
	const float ex = static_cast<float>(static_cast<double>(v) - 2.0);
	const f3 q = mk3((2.0f * kPi) / primaryWavelengths.x, (2.0f * kPi) / primaryWavelengths.y, (2.0f * kPi) / primaryWavelengths.z);
	const f3 p = mk3(std::pow(q.x, ex), std::pow(q.y, ex), std::pow(q.z, ex));
	return (s * p) * K;
}

// sunsky.cu:163-166
f3 ortho(f3 a) { return std::fabs(a.x) > std::fabs(a.z) ? mk3(-a.y, a.x, 0.0f) : mk3(0.0f, -a.z, a.y); }
} // namespace

void sun_setup(float sun_x, float sun_y, SunParams& S) {
	// kernel.cu:683: float sun_angular = cos(sunSize * pi / 180.f)
	S.sunAngularDiameterCos = std::cos(sunSize * kPi / 180.f); // cos(float): the float overload

	// kernel.cu:708: normalize(fromSpherical((sun_position - vec2(0.0, 0.5)) * vec2(6.28f, 3.14f))); sunsky.cu:28-30
	const float px = (sun_x - 0.0f) * 6.28f;
	const float py = (sun_y - 0.5f) * 3.14f;
	// float arguments pick the float overloads and the products are binary32 (tests/golden/ref_sunsky.npz, made by the
	// reference's own fromSpherical, tells this from a binary64 evaluation at sun position (0.3, 0.12))
	f3 d = mk3(std::cos(px) * std::sin(py), std::sin(px) * std::sin(py), std::cos(py));
	d = normalize(d);
	S.sunDirection[0] = d.x;
	S.sunDirection[1] = d.y;
	S.sunDirection[2] = d.z;

	const f3 up = mk3(0.0f, 0.0f, 1.0f); // sunsky.cu:5
	S.sunE = SunIntensity(dot(d, up));

	const f3 rayleighAtX = mk3(static_cast<float>(5.176821E-6), static_cast<float>(1.2785348E-5), static_cast<float>(2.8530756E-5)); // sunsky.cu:41
	const f3 K = mk3(static_cast<float>(0.686), static_cast<float>(0.678), static_cast<float>(0.666));                               // sunsky.cu:4
	const f3 wavelengths = mk3(static_cast<float>(680E-9), static_cast<float>(550E-9), static_cast<float>(450E-9));                  // sunsky.cuh:43
	const f3 mieAtX = totalMie(wavelengths, K, turbidity) * mieCoefficient;                                                           // sunsky.cu:44
	const f3 total = rayleighAtX + mieAtX;
	for (int i = 0; i < 3; ++i) {
		S.rayleighAtX[i] = (&rayleighAtX.x)[i];
		S.mieAtX[i] = (&mieAtX.x)[i];
		S.totalLightAtX[i] = (&total.x)[i];
	}
	// sunsky.cu:66-67
	S.mixFactor = gclamp(std::pow(1.0f - dot(up, d), 5.0f), 0.0f, 1.0f);

	// sunsky.cu:172-175
	const f3 dir = normalize(d);
	const f3 o1 = normalize(ortho(dir));
	const f3 o2 = normalize(cross(dir, o1));
	for (int i = 0; i < 3; ++i) {
		S.coneDir[i] = (&dir.x)[i];
		S.coneO1[i] = (&o1.x)[i];
		S.coneO2[i] = (&o2.x)[i];
	}
	S.coneExtent = 1.0f - S.sunAngularDiameterCos; // kernel.cu:410
}

} // namespace tyr
