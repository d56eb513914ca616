// detmath.hpp -- deterministic sinf / cosf / expf / powf for the shading kernels.
//
// The reference calls CUDA's sinf/cosf/powf/expf (kernel.cu:207, 427-429, 465, 512,
// 527, 542, 554; sunsky.cu:56, 181-184).  No two libm implementations agree bit for
// bit, and one flipped Russian-roulette decision renumbers every later queue slot
// (the RNG seed depends on the slot, kernel.cu:363), so this path fixes ONE
// definition: each function is a fixed polynomial kernel evaluated in binary64 with
// plain multiply/add Horner steps and a single final rounding to binary32.
// MI355X runs FP64 vector ops at half the FP32 rate, so the fp64 core costs little,
// the results are within 1 ulp of libm (<= 0.5000003 ulp), and a CPU that evaluates
// the same sequence of IEEE operations reproduces them exactly (DESIGN.md "Numeric
// contract"; the test oracle carries its own implementation of the same contract).
//
// Contract: x86 or gfx950, round-to-nearest-even, no FMA contraction, no fast-math.
#pragma once

#include <hip/hip_runtime.h>

#include "vecmath.hpp"

namespace tyr {
namespace dm {

// Taylor coefficients 1/n!, n = 0..16 (correctly rounded binary64)
__device__ static constexpr double kInvFact[17] = {
	1.0,
	1.0,
	0x1.0000000000000p-1,
	0x1.5555555555555p-3,
	0x1.5555555555555p-5,
	0x1.1111111111111p-7,
	0x1.6c16c16c16c17p-10,
	0x1.a01a01a01a01ap-13,
	0x1.a01a01a01a01ap-16,
	0x1.71de3a556c734p-19,
	0x1.27e4fb7789f5cp-22,
	0x1.ae64567f544e4p-26,
	0x1.1eed8eff8d898p-29,
	0x1.6124613a86d09p-33,
	0x1.93974a8c07c9dp-37,
	0x1.ae7f3e733b81fp-41,
	0x1.ae7f3e733b81fp-45,
};
// atanh series 2/(2k+1), k = 1..11
__device__ static constexpr double kAtanh[11] = {
	0x1.5555555555555p-1, 0x1.999999999999ap-2, 0x1.2492492492492p-2, 0x1.c71c71c71c71cp-3,
	0x1.745d1745d1746p-3, 0x1.3b13b13b13b14p-3, 0x1.1111111111111p-3, 0x1.e1e1e1e1e1e1ep-4,
	0x1.af286bca1af28p-4, 0x1.8618618618618p-4, 0x1.642c8590b2164p-4,
};

constexpr double kLog2e = 0x1.71547652b82fep+0;
constexpr double kLn2 = 0x1.62e42fefa39efp-1;
constexpr double kLn2Hi = 0x1.62e42fee00000p-1;
constexpr double kLn2Lo = 0x1.a39ef35793c76p-33;
constexpr double kTwoOverPi = 0x1.45f306dc9c883p-1;
constexpr double kPio2Hi = 0x1.921fb54400000p+0;
constexpr double kPio2Lo = 0x1.0b4611a626331p-34;
constexpr double kSqrt2 = 0x1.6a09e667f3bcdp+0;

// round to nearest even; rint lowers to v_rndne_f64 and equals the 1.5*2^52 add/subtract
// trick for every |t| < 2^51
__device__ __forceinline__ double rne(double t) { return __builtin_rint(t); }
__device__ __forceinline__ double pow2i(int k) { return __longlong_as_double((long long)(k + 1023) << 52); }

// exp(r), |r| <= 0.35, degree 11
__device__ __forceinline__ double exp_poly(double r) {
	double p = kInvFact[11];
#pragma unroll
	for (int n = 10; n >= 0; --n)
		p = p * r + kInvFact[n];
	return p;
}

__device__ __forceinline__ float expf_det(float xf) {
	if (xf != xf)
		return xf;
	const double x = (double)xf;
	if (x > 89.0)
		return __builtin_inff();
	if (x < -104.0)
		return 0.0f;
	const double kd = rne(x * kLog2e);
	const double r = (x - kd * kLn2Hi) - kd * kLn2Lo;
	return (float)(exp_poly(r) * pow2i((int)kd));
}

// sin(r) = r + r*(z*P(z)), cos(r) = 1 + z*Q(z), z = r*r, |r| <= pi/4
__device__ __forceinline__ double sin_poly(double r) {
	const double z = r * r;
	double p = -kInvFact[15];
	p = p * z + kInvFact[13];
	p = p * z - kInvFact[11];
	p = p * z + kInvFact[9];
	p = p * z - kInvFact[7];
	p = p * z + kInvFact[5];
	p = p * z - kInvFact[3];
	return r + r * (z * p);
}
__device__ __forceinline__ double cos_poly(double r) {
	const double z = r * r;
	double p = kInvFact[16];
	p = p * z - kInvFact[14];
	p = p * z + kInvFact[12];
	p = p * z - kInvFact[10];
	p = p * z + kInvFact[8];
	p = p * z - kInvFact[6];
	p = p * z + kInvFact[4];
	p = p * z - kInvFact[2];
	return 1.0 + z * p;
}
__device__ __forceinline__ double reduce_pio2(double x, int& q) {
	const double kd = rne(x * kTwoOverPi);
	q = (int)kd & 3;
	return (x - kd * kPio2Hi) - kd * kPio2Lo;
}
__device__ __forceinline__ float sinf_det(float xf) {
	if (!(fabsf(xf) < 1048576.0f))
		return xf - xf;
	int q;
	const double r = reduce_pio2((double)xf, q);
	const double s = (q & 1) ? cos_poly(r) : sin_poly(r);
	return (float)((q & 2) ? -s : s);
}
__device__ __forceinline__ float cosf_det(float xf) {
	if (!(fabsf(xf) < 1048576.0f))
		return xf - xf;
	int q;
	const double r = reduce_pio2((double)xf, q);
	const double c = (q & 1) ? sin_poly(r) : cos_poly(r);
	return (float)(((q + 1) & 2) ? -c : c);
}
// both at once (shares the reduction; each result is bit-identical to the single calls)
__device__ __forceinline__ void sincosf_det(float xf, float& s_out, float& c_out) {
	if (!(fabsf(xf) < 1048576.0f)) {
		s_out = c_out = xf - xf;
		return;
	}
	int q;
	const double r = reduce_pio2((double)xf, q);
	const double sp = sin_poly(r), cp = cos_poly(r);
	const double s = (q & 1) ? cp : sp;
	const double c = (q & 1) ? sp : cp;
	s_out = (float)((q & 2) ? -s : s);
	c_out = (float)(((q + 1) & 2) ? -c : c);
}

__device__ __forceinline__ double log2_det(double x) {
	unsigned long long u = (unsigned long long)__double_as_longlong(x);
	int e = (int)((u >> 52) & 0x7ff) - 1023;
	u = (u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
	double m = __longlong_as_double((long long)u);
	if (m > kSqrt2) {
		m = m * 0.5;
		e = e + 1;
	}
	const double s = (m - 1.0) / (m + 1.0);
	const double z = s * s;
	double p = kAtanh[10];
#pragma unroll
	for (int k = 9; k >= 0; --k)
		p = p * z + kAtanh[k];
	const double logm = s * (2.0 + z * p);
	return logm * kLog2e + (double)e;
}
__device__ __forceinline__ float powf_det(float xf, float yf) {
	if (yf != yf)
		return yf;
	if (!(xf > 0.0f)) {
		if (xf == 0.0f)
			return (yf > 0.0f) ? 0.0f : ((yf == 0.0f) ? 1.0f : __builtin_inff());
		return xf != xf ? xf : __builtin_nanf("");
	}
	if (xf == __builtin_inff())
		return (yf > 0.0f) ? __builtin_inff() : ((yf == 0.0f) ? 1.0f : 0.0f);
	const double t = (double)yf * log2_det((double)xf);
	if (t > 129.0)
		return __builtin_inff();
	if (t < -152.0)
		return 0.0f;
	const double kd = rne(t);
	const double w = (t - kd) * kLn2;
	return (float)(exp_poly(w) * pow2i((int)kd));
}

} // namespace dm
} // namespace tyr
