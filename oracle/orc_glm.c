/*
 * oracle/orc_glm.c -- the oracle's vector helpers (orc_internal.h v3*, glm's evaluation order) behind one array
 * entry point, so that tests can pin them to the vendored glm itself (oracle/ref_harness.cpp ref_glm, same op codes;
 * golden copy of its answers: tests/golden/ref_glm.npz).  TEST INFRASTRUCTURE (see orc.h).
 *
 * glm references (Dependencies/glm-0.9.9.3/glm/detail): func_geometric.inl:14-20 length, 54-61 dot, 74-85 cross,
 * 88-96 normalize, 110-116 reflect; func_common.inl:16-29 min/max, 103-111 mix, 257-265 smoothstep, 566 clamp;
 * type_vec3.inl operators.  pow / exp go through <cmath> in glm (func_exponential.inl); here they are the deterministic
 * layer's dm_powf / dm_expf, except the exponents the path uses with a closed form (0.5 -> sqrt).
 */
#include "orc_internal.h"

int orc_glm(int op, const float* a, const float* b, const float* c, int n, float* out) {
	for (int i = 0; i < n; ++i) {
		v3 A = v3load(a + 3 * i), B = v3load(b + 3 * i), C = v3load(c + 3 * i);
		v3 r = v3make(0.0f, 0.0f, 0.0f);
		switch (op) {
		case 0: r.x = v3dot(A, B); break;
		case 1: r = v3cross(A, B); break;
		case 2: r = v3normalize(A); break;
		case 3: r.x = v3length(A); break;
		case 4: r = v3reflect(A, B); break;
		case 5: r = v3make(glm_minf(A.x, B.x), glm_minf(A.y, B.y), glm_minf(A.z, B.z)); break;
		case 6: r = v3make(glm_maxf(A.x, B.x), glm_maxf(A.y, B.y), glm_maxf(A.z, B.z)); break;
		case 7: r = v3make(glm_clampf(A.x, B.x, B.y), glm_clampf(A.y, B.x, B.y), glm_clampf(A.z, B.x, B.y)); break;
		case 8: r = v3add(A, v3rscale(C.x, v3sub(B, A))); break; /* func_common.inl:103-111: x + a * (y - x) (as orc_sunsky.c uses it) */
		case 9: { /* func_common.inl:257-265 (as orc_sunsky.c:167) */
			float t = glm_clampf((A.x - C.x) / (C.y - C.x), 0.0f, 1.0f);
			r.x = t * t * (3.0f - 2.0f * t);
			break;
		}
		case 10: /* exponent 0.5 is the path's closed form sqrt (orc_sunsky.c: pow(somethingElse * Fex, vec3(0.5)), sunsky.cu:66) */
			r = v3make(B.x == 0.5f ? sqrtf(A.x) : dm_powf(A.x, B.x), B.y == 0.5f ? sqrtf(A.y) : dm_powf(A.y, B.y), B.z == 0.5f ? sqrtf(A.z) : dm_powf(A.z, B.z));
			break;
		case 11: r = v3divs(A, C.x); break;
		case 12: r = v3scale(A, C.x); break;
		case 13: r = v3rscale(C.x, A); break;
		case 14: r = v3make(dm_expf(A.x), dm_expf(A.y), dm_expf(A.z)); break;
		case 15: r = v3mul(A, B); break;
		case 16: r = v3div(A, B); break;
		case 17: r = v3neg(A); break;
		case 18: r = v3add(A, B); break;
		case 19: r = v3sub(A, B); break;
		default: return -1;
		}
		v3store(out + 3 * i, r);
	}
	return 0;
}
