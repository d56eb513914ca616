/*
 * oracle/orc_sunsky.c -- CPU restatement of sunsky.cu (analytic sun/sky, cone
 * sampling) and of the sun-related host prologue of launch_kernels.
 * TEST INFRASTRUCTURE (see orc.h).  Citations: /root/reference/PathTracer/.
 *
 * Everything that depends only on sunDirection is hoisted into orc_sun_setup():
 * the reference recomputes it per call on the device, the value is identical.
 * Host-side transcendentals (called once per sun change, never per ray) use
 * libm exactly where the reference's host/device code calls cos/sin/acos/exp/pow.
 */
#include "orc_internal.h"

/* sunsky.cuh:26-43 */
static const float sunSize = 1.5f;
#define cutoffAngle (ORC_PI / 1.95f)
static const float steepness = 1.5f;
static const float turbidity = 1.0f;
static const float mieCoefficient = 0.005f;
static const float mieDirectionalG = 0.80f;
static const float v_ = 4.0f;
static const float rayleighZenithLength = 8.4E3f;
static const float mieZenithLength = 1.25E3f;
static const float sunIntensity = 1000.0f;

/* sunsky.cu:24-26 -- float acos/exp (CUDA's float overloads), double max/1.0- */
static float SunIntensity(float zenithAngleCos) {
	float e = expf(-((cutoffAngle - acosf(zenithAngleCos)) / steepness));
	double m = 1.0 - (double)e;
	if (0.0 < m) {
		/* glm::max(0.0, m) */
	} else {
		m = 0.0;
	}
	return (float)((double)sunIntensity * m);
}

/* sunsky.cu:15-19 */
static v3 totalMie(v3 primaryWavelengths, v3 K, float T) {
	float c = (float)((0.2 * (double)T) * 10E-18);
	float s = 0.434f * c * ORC_PI;
	float ex = (float)((double)v_ - 2.0);
	v3 q = v3make((2.0f * ORC_PI) / primaryWavelengths.x, (2.0f * ORC_PI) / primaryWavelengths.y, (2.0f * ORC_PI) / primaryWavelengths.z);
	v3 p = v3make(powf(q.x, ex), powf(q.y, ex), powf(q.z, ex));
	return v3mul(v3rscale(s, p), K);
}

/* sunsky.cu:163-166 */
static v3 ortho(v3 v) {
	return fabsf(v.x) > fabsf(v.z) ? v3make(-v.y, v.x, 0.0f) : v3make(0.0f, -v.z, v.y);
}

void orc_sun_setup(const float sun_position[2], orc_sunparams* S) {
	/* kernel.cu:683  float sun_angular = cos(sunSize * pi / 180.f); */
	S->sunAngularDiameterCos = cosf(sunSize * ORC_PI / 180.f); /* cos(float): the float overload, under nvcc and in the reference compiled as C++ (oracle/_ref) */
	/* kernel.cu:708  normalize(fromSpherical((sun_position - vec2(0.0,0.5)) * vec2(6.28f,3.14f)))
	 * sunsky.cu:28-30 fromSpherical: host cos/sin on float arguments = the float overloads, products in binary32
	 * (pinned by tests/golden/ref_sunsky.npz: sun position (0.3, 0.12) tells this from a binary64 evaluation) */
	float px = (sun_position[0] - 0.0f) * 6.28f;
	float py = (sun_position[1] - 0.5f) * 3.14f;
	v3 d = v3make(cosf(px) * sinf(py), sinf(px) * sinf(py), cosf(py));
	d = v3normalize(d);
	v3store(S->sunDirection, d);

	const v3 up = v3make(0.0f, 0.0f, 1.0f); /* sunsky.cu:5 */
	float cosSunUpAngle = v3dot(d, up);
	S->sunE = SunIntensity(cosSunUpAngle);

	/* sunsky.cu:41 / 85 (double literals narrowed by the vec3 constructor) */
	v3 rayleighAtX = v3make((float)5.176821E-6, (float)1.2785348E-5, (float)2.8530756E-5);
	v3 K = v3make((float)0.686, (float)0.678, (float)0.666);                 /* sunsky.cu:4 */
	v3 primaryWavelengths = v3make((float)680E-9, (float)550E-9, (float)450E-9); /* sunsky.cuh:43 */
	v3 mieAtX = v3scale(totalMie(primaryWavelengths, K, turbidity), mieCoefficient);
	v3store(S->rayleighAtX, rayleighAtX);
	v3store(S->mieAtX, mieAtX);
	v3store(S->totalLightAtX, v3add(rayleighAtX, mieAtX));

	/* sunsky.cu:66  clamp(pow(1.0f - dot(up, sunDirection), 5.0f), 0, 1) */
	S->mixFactor = glm_clampf(powf(1.0f - v3dot(up, d), 5.0f), 0.0f, 1.0f);

	/* sunsky.cu:170-175 with dir = sunDirection */
	v3 dir = v3normalize(d);
	v3 o1 = v3normalize(ortho(dir));
	v3 o2 = v3normalize(v3cross(dir, o1));
	v3store(S->coneDir, dir);
	v3store(S->coneO1, o1);
	v3store(S->coneO2, o2);
	S->coneExtent = 1.0f - S->sunAngularDiameterCos;
}

/* the part shared by sun/sky/sunsky: sunsky.cu:33-66 / 77-109 / 117-151 */
typedef struct {
	float cosViewSunAngle;
	v3 Fex;
	v3 sky;
} atmos;

static atmos atmosphere(const orc_sunparams* S, v3 viewDir) {
	atmos a;
	const v3 up = v3make(0.0f, 0.0f, 1.0f);
	v3 sunDirection = v3load(S->sunDirection);
	v3 rayleighAtX = v3load(S->rayleighAtX);
	v3 mieAtX = v3load(S->mieAtX);
	v3 totalLightAtX = v3load(S->totalLightAtX);

	float cosViewSunAngle = v3dot(viewDir, sunDirection);
	float cosUpViewAngle = v3dot(up, viewDir);
	float zenithAngle = glm_maxf(0.0f, cosUpViewAngle);
	float rayleighOpticalLength = rayleighZenithLength / zenithAngle;
	float mieOpticalLength = mieZenithLength / zenithAngle;

	v3 ext = v3add(v3scale(rayleighAtX, rayleighOpticalLength), v3scale(mieAtX, mieOpticalLength));
	v3 Fex = v3make(dm_expf(-ext.x), dm_expf(-ext.y), dm_expf(-ext.z));

	/* sunsky.cu:10-12 RayleighPhase: (3.0/(16.0*pi)) * (1.0 + powf(c, 2.0)) */
	float c2 = cosViewSunAngle * cosViewSunAngle;
	float rayleighPhase = (float)((3.0 / (16.0 * (double)ORC_PI)) * (1.0 + (double)c2));
	/* sunsky.cu:20-22 hgPhase: (1.0/(4.0*pi)) * ((1.0 - powf(g,2.0)) / pow(1.0 - 2.0*g*c + powf(g,2.0), 1.5)) */
	float g2 = mieDirectionalG * mieDirectionalG;
	double hb = 1.0 - 2.0 * (double)mieDirectionalG * (double)cosViewSunAngle + (double)g2;
	double hp = hb * sqrt(hb); /* pow(x, 1.5) */
	float hg = (float)((1.0 / (4.0 * (double)ORC_PI)) * ((1.0 - (double)g2) / hp));

	v3 rayleighXtoEye = v3scale(rayleighAtX, rayleighPhase);
	v3 mieXtoEye = v3scale(mieAtX, hg);
	v3 lightFromXtoEye = v3add(rayleighXtoEye, mieXtoEye);
	v3 somethingElse = v3rscale(S->sunE, v3div(lightFromXtoEye, totalLightAtX));

	v3 sky = v3mul(somethingElse, v3make(1.0f - Fex.x, 1.0f - Fex.y, 1.0f - Fex.z));
	/* mix(vec3(1), pow(somethingElse*Fex, vec3(0.5)), a) = 1 + a * (y - 1) */
	v3 sf = v3mul(somethingElse, Fex);
	v3 y = v3make(sqrtf(sf.x), sqrtf(sf.y), sqrtf(sf.z));
	float a_ = S->mixFactor;
	v3 m = v3make(1.0f + a_ * (y.x - 1.0f), 1.0f + a_ * (y.y - 1.0f), 1.0f + a_ * (y.z - 1.0f));
	sky = v3mul(sky, m);

	a.cosViewSunAngle = cosViewSunAngle;
	a.Fex = Fex;
	a.sky = sky;
	return a;
}

/* sunsky.cu:32-74 */
void orc_sun(const orc_sunparams* S, const float viewDir[3], float out[3]) {
	atmos a = atmosphere(S, v3load(viewDir));
	/* :70  float sundisk = sunAngularDiameterCos < (cosViewSunAngle ? 1.0 : 0.0);  (precedence quirk kept) */
	float sundisk = ((double)S->sunAngularDiameterCos < (a.cosViewSunAngle ? 1.0 : 0.0)) ? 1.0f : 0.0f;
	v3 sun = v3scale(v3rscale(S->sunE * 19000.0f, a.Fex), sundisk);
	v3store(out, v3rscale(0.01f, sun));
}

/* sunsky.cu:76-114 */
void orc_sky(const orc_sunparams* S, const float viewDir[3], float out[3]) {
	atmos a = atmosphere(S, v3load(viewDir));
	v3store(out, v3rscale(1.f * 0.01f, a.sky));
}

/* sunsky.cu:116-161 */
void orc_sunsky(const orc_sunparams* S, const float viewDir[3], float out[3]) {
	if (S->sunAngularDiameterCos == 1.0f) {
		out[0] = 1.0f;
		out[1] = 0.0f;
		out[2] = 0.0f;
		return;
	}
	atmos a = atmosphere(S, v3load(viewDir));
	float e0 = S->sunAngularDiameterCos;
	float e1 = S->sunAngularDiameterCos + 0.00002f;
	float t = glm_clampf((a.cosViewSunAngle - e0) / (e1 - e0), 0.0f, 1.0f);
	float sundisk = t * t * (3.0f - 2.0f * t);
	v3 sun = v3scale(v3scale(v3rscale(S->sunE * 19000.0f, a.Fex), sundisk), 1E-5f);
	v3store(out, v3rscale(0.01f, v3add(sun, a.sky)));
}

/* sunsky.cu:170-185 with dir = sunDirection, extent = 1 - sunAngularDiameterCos (kernel.cu:410, 546) */
void orc_cone_sample(const orc_sunparams* S, uint32_t* seed, float out[3]) {
	float rx = rng_float2(seed);
	float ry = rng_float2(seed);
	rx = rx * 2.f * ORC_PI;
	ry = 1.0f - ry * S->coneExtent;
	float oneminus = sqrtf(1.0f - ry * ry);
	v3 a = v3rscale(dm_cosf(rx) * oneminus, v3load(S->coneO1));
	v3 b = v3rscale(dm_sinf(rx) * oneminus, v3load(S->coneO2));
	v3 c = v3rscale(ry, v3load(S->coneDir));
	v3store(out, v3add(v3add(a, b), c));
}

/* exported wrappers of the math layer and RNG (for known-answer tests) */
float orc_dm_sinf(float x) { return dm_sinf(x); }
float orc_dm_cosf(float x) { return dm_cosf(x); }
float orc_dm_expf(float x) { return dm_expf(x); }
float orc_dm_powf(float x, float y) { return dm_powf(x, y); }
uint32_t orc_random_int(uint32_t* seed) { return rng_int(seed); }
float orc_random_float(uint32_t* seed) { return rng_float(seed); }
float orc_random_float2(uint32_t* seed) { return rng_float2(seed); }
int orc_random_int_between_0_and_max(uint32_t* seed, int max) { return rng_int_0_max(seed, max); }
