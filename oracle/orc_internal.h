/*
 * oracle/orc_internal.h -- vector arithmetic in glm's evaluation order and the
 * deterministic transcendental layer.  TEST INFRASTRUCTURE (see orc.h).
 *
 * Numeric spec (DESIGN.md "Numeric contract"):
 *   - every +,-,*,/ and sqrt is a single IEEE-754 binary32 (or, where the
 *     reference's unsuffixed literals promote, binary64) operation in the
 *     reference's source order; no FMA contraction (-ffp-contract=off);
 *   - glm functions follow Dependencies/glm-0.9.9.3/detail:
 *       dot       func_geometric.inl:54-61   (x + y) + z
 *       cross     func_geometric.inl:74-85
 *       normalize func_geometric.inl:88-96   v * (1 / sqrt(dot(v,v)))  (func_exponential.inl:136-139)
 *       reflect   func_geometric.inl:110-116 I - N * dot(N,I) * 2
 *       length    func_geometric.inl:14-20
 *       min/max   func_common.inl:16-29      (y < x) ? y : x   /   (x < y) ? y : x
 *       mix       func_common.inl:103-111    x + a * (y - x)
 *       smoothstep func_common.inl:257-265
 *   - sinf/cosf/expf/powf are evaluated by fixed polynomial kernels in binary64
 *     (plain mul/add Horner, no FMA) and rounded once to binary32.  The same
 *     operation sequence is implemented independently in the HIP kernels, which
 *     makes GPU and oracle bit-identical; accuracy versus libm is checked in
 *     tests/test_oracle_math.py (<= 1 ulp).
 */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#include "orc.h"

typedef struct {
	float x, y, z;
} v3;

static const float ORC_PI = 3.1415926535897932f; /* variables.h:3 */
#define ORC_INV_PI (1.0f / ORC_PI)               /* variables.h:4 */

static inline v3 v3make(float x, float y, float z) {
	v3 r = { x, y, z };
	return r;
}
static inline v3 v3load(const float* p) { return v3make(p[0], p[1], p[2]); }
static inline void v3store(float* p, v3 a) {
	p[0] = a.x;
	p[1] = a.y;
	p[2] = a.z;
}
static inline v3 v3add(v3 a, v3 b) { return v3make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3sub(v3 a, v3 b) { return v3make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3mul(v3 a, v3 b) { return v3make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3div(v3 a, v3 b) { return v3make(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline v3 v3scale(v3 a, float s) { return v3make(a.x * s, a.y * s, a.z * s); }  /* vec * scalar */
static inline v3 v3rscale(float s, v3 a) { return v3make(s * a.x, s * a.y, s * a.z); } /* scalar * vec */
static inline v3 v3divs(v3 a, float s) { return v3make(a.x / s, a.y / s, a.z / s); }   /* type_vec3.inl:696-702 */
static inline v3 v3neg(v3 a) { return v3make(-a.x, -a.y, -a.z); }
static inline float v3dot(v3 a, v3 b) {
	float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
	return tx + ty + tz;
}
static inline v3 v3cross(v3 x, v3 y) {
	return v3make(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
static inline v3 v3normalize(v3 v) { return v3scale(v, 1.0f / sqrtf(v3dot(v, v))); }
static inline float v3length(v3 v) { return sqrtf(v3dot(v, v)); }
static inline v3 v3reflect(v3 I, v3 N) { return v3sub(I, v3scale(v3scale(N, v3dot(N, I)), 2.0f)); }
static inline float glm_minf(float x, float y) { return (y < x) ? y : x; }
static inline float glm_maxf(float x, float y) { return (x < y) ? y : x; }
static inline float glm_clampf(float x, float lo, float hi) { return glm_minf(glm_maxf(x, lo), hi); }

/* ---------------- deterministic transcendental layer ---------------------- */

static inline double dm_round(double t) {
	/* round-to-nearest-even for |t| < 2^51 via the 1.5*2^52 trick */
	const double M = 6755399441055744.0;
	double u = t + M;
	return u - M;
}
static inline double dm_pow2i(int k) {
	/* exact 2^k for -1022 <= k <= 1023 */
	uint64_t u = (uint64_t)(k + 1023) << 52;
	double d;
	memcpy(&d, &u, 8);
	return d;
}
/* exp(r) for |r| <= 0.35: Taylor to r^11, Horner, plain mul/add */
static inline double dm_exp_poly(double r) {
	double p = 0x1.ae64567f544e4p-26; /* 1/11! */
	p = p * r + 0x1.27e4fb7789f5cp-22;  /* 1/10! */
	p = p * r + 0x1.71de3a556c734p-19;  /* 1/9!  */
	p = p * r + 0x1.a01a01a01a01ap-16;  /* 1/8!  */
	p = p * r + 0x1.a01a01a01a01ap-13;  /* 1/7!  */
	p = p * r + 0x1.6c16c16c16c17p-10;  /* 1/6!  */
	p = p * r + 0x1.1111111111111p-7;   /* 1/5!  */
	p = p * r + 0x1.5555555555555p-5;   /* 1/4!  */
	p = p * r + 0x1.5555555555555p-3;   /* 1/3!  */
	p = p * r + 0x1.0000000000000p-1;   /* 1/2!  */
	p = p * r + 1.0;
	p = p * r + 1.0;
	return p;
}
static inline float dm_expf(float xf) {
	if (xf != xf)
		return xf;
	double x = (double)xf;
	if (x > 89.0)
		return INFINITY;
	if (x < -104.0)
		return 0.0f;
	double kd = dm_round(x * 0x1.71547652b82fep+0);
	double r = (x - kd * 0x1.62e42fee00000p-1) - kd * 0x1.a39ef35793c76p-33;
	return (float)(dm_exp_poly(r) * dm_pow2i((int)kd));
}
/* sin/cos kernels on |r| <= pi/4 (Taylor to r^15 / r^16) */
static inline double dm_sin_poly(double r) {
	double z = r * r;
	double p = -0x1.ae7f3e733b81fp-41; /* -1/15! */
	p = p * z + 0x1.6124613a86d09p-33;   /*  1/13! */
	p = p * z - 0x1.ae64567f544e4p-26;   /* -1/11! */
	p = p * z + 0x1.71de3a556c734p-19;   /*  1/9!  */
	p = p * z - 0x1.a01a01a01a01ap-13;   /* -1/7!  */
	p = p * z + 0x1.1111111111111p-7;    /*  1/5!  */
	p = p * z - 0x1.5555555555555p-3;    /* -1/3!  */
	return r + r * (z * p);
}
static inline double dm_cos_poly(double r) {
	double z = r * r;
	double p = 0x1.ae7f3e733b81fp-45;  /*  1/16! */
	p = p * z - 0x1.93974a8c07c9dp-37;   /* -1/14! */
	p = p * z + 0x1.1eed8eff8d898p-29;   /*  1/12! */
	p = p * z - 0x1.27e4fb7789f5cp-22;   /* -1/10! */
	p = p * z + 0x1.a01a01a01a01ap-16;   /*  1/8!  */
	p = p * z - 0x1.6c16c16c16c17p-10;   /* -1/6!  */
	p = p * z + 0x1.5555555555555p-5;    /*  1/4!  */
	p = p * z - 0x1.0000000000000p-1;    /* -1/2!  */
	return 1.0 + z * p;
}
/* quadrant reduction: x = k*pi/2 + r, valid for |x| < 2^20 */
static inline double dm_reduce_pio2(double x, int* quadrant) {
	double kd = dm_round(x * 0x1.45f306dc9c883p-1);
	double r = (x - kd * 0x1.921fb54400000p+0) - kd * 0x1.0b4611a626331p-34;
	*quadrant = (int)kd & 3;
	return r;
}
static inline float dm_sinf(float xf) {
	if (!(fabsf(xf) < 1048576.0f))
		return xf - xf; /* NaN for inf/NaN; out-of-contract arguments */
	int q;
	double r = dm_reduce_pio2((double)xf, &q);
	double s = (q & 1) ? dm_cos_poly(r) : dm_sin_poly(r);
	return (float)((q & 2) ? -s : s);
}
static inline float dm_cosf(float xf) {
	if (!(fabsf(xf) < 1048576.0f))
		return xf - xf;
	int q;
	double r = dm_reduce_pio2((double)xf, &q);
	double c = (q & 1) ? dm_sin_poly(r) : dm_cos_poly(r);
	return (float)(((q + 1) & 2) ? -c : c);
}
/* log2(x) for finite x > 0 (binary64) */
static inline double dm_log2(double x) {
	uint64_t u;
	memcpy(&u, &x, 8);
	int e = (int)((u >> 52) & 0x7ff) - 1023;
	u = (u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
	double m;
	memcpy(&m, &u, 8);
	if (m > 0x1.6a09e667f3bcdp+0) {
		m = m * 0.5;
		e = e + 1;
	}
	double s = (m - 1.0) / (m + 1.0);
	double z = s * s;
	double p = 0x1.642c8590b2164p-4;  /* 2/23 */
	p = p * z + 0x1.8618618618618p-4; /* 2/21 */
	p = p * z + 0x1.af286bca1af28p-4; /* 2/19 */
	p = p * z + 0x1.e1e1e1e1e1e1ep-4; /* 2/17 */
	p = p * z + 0x1.1111111111111p-3; /* 2/15 */
	p = p * z + 0x1.3b13b13b13b14p-3; /* 2/13 */
	p = p * z + 0x1.745d1745d1746p-3; /* 2/11 */
	p = p * z + 0x1.c71c71c71c71cp-3; /* 2/9  */
	p = p * z + 0x1.2492492492492p-2; /* 2/7  */
	p = p * z + 0x1.999999999999ap-2; /* 2/5  */
	p = p * z + 0x1.5555555555555p-1; /* 2/3  */
	double logm = s * (2.0 + z * p);
	return logm * 0x1.71547652b82fep+0 + (double)e;
}
static inline float dm_powf(float xf, float yf) {
	if (yf != yf)
		return yf;
	if (!(xf > 0.0f)) {
		if (xf == 0.0f)
			return (yf > 0.0f) ? 0.0f : ((yf == 0.0f) ? 1.0f : INFINITY);
		return xf != xf ? xf : NAN; /* negative base: out of contract */
	}
	if (xf == INFINITY)
		return (yf > 0.0f) ? INFINITY : ((yf == 0.0f) ? 1.0f : 0.0f);
	double t = (double)yf * dm_log2((double)xf);
	if (t > 129.0)
		return INFINITY;
	if (t < -152.0)
		return 0.0f;
	double kd = dm_round(t);
	double w = (t - kd) * 0x1.62e42fefa39efp-1;
	return (float)(dm_exp_poly(w) * dm_pow2i((int)kd));
}

/* ---------------- RNG (kernel.cu:23-41) ----------------------------------- */
static inline uint32_t rng_int(uint32_t* seed) {
	uint32_t s = *seed;
	s ^= s << 13;
	s ^= s >> 17;
	s ^= s << 5;
	*seed = s;
	return s;
}
static inline float rng_float(uint32_t* seed) { return (float)rng_int(seed) * 2.3283064365387e-10f; }
static inline float rng_float2(uint32_t* seed) { return (float)(rng_int(seed) >> 16) / 65535.0f; }
static inline int rng_int_0_max(uint32_t* seed, int max) { return (int)(rng_float(seed) * ((float)max + 0.99999f)); }

#endif
