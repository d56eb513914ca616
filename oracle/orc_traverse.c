/*
 * oracle/orc_traverse.c -- CPU restatement of the device traversal functions:
 *   BBox::intersect            Bbox.h:38-62
 *   Triangle::intersect        loader.h:21-46
 *   CachedBVH::intersect       bvh.h:118-161  (closest hit)
 *   CachedBVH::intersectSimple bvh.h:213-256  (any hit)
 *   Sphere::intersect          kernel.cu:83-93
 * TEST INFRASTRUCTURE (see orc.h).  Checked bit-for-bit against the reference's
 * own headers compiled in oracle/_ref (tests/test_oracle_golden.py).
 */
#include "orc_internal.h"

/* Bbox.h:38-62 */
int orc_bbox_intersect(const orc_bbox* b, const float origin[3], const float invDir[3], const int rayDirNeg[3], float lowestIntersect) {
	float tMin = (b->bounds[rayDirNeg[0]][0] - origin[0]) * invDir[0];
	float tMax = (b->bounds[1 - rayDirNeg[0]][0] - origin[0]) * invDir[0];
	float tyMin = (b->bounds[rayDirNeg[1]][1] - origin[1]) * invDir[1];
	float tyMax = (b->bounds[1 - rayDirNeg[1]][1] - origin[1]) * invDir[1];

	if (tMin > tyMax || tyMin > tMax)
		return 0;
	if (tyMin > tMin)
		tMin = tyMin;
	if (tyMax < tMax)
		tMax = tyMax;

	float tzMin = (b->bounds[rayDirNeg[2]][2] - origin[2]) * invDir[2];
	float tzMax = (b->bounds[1 - rayDirNeg[2]][2] - origin[2]) * invDir[2];

	if (tMin > tzMax || tzMin > tMax)
		return 0;
	if (tzMin > tMin)
		tMin = tzMin;
	if (tzMax < tMax)
		tMax = tzMax;

	return (tMin < lowestIntersect) && (tMax > 0);
}

/* loader.h:21-46 (Moller-Trumbore, back faces culled, 0 = miss) */
float orc_triangle_intersect(const orc_triangle* t, const float origin[3], const float direction[3]) {
	v3 e1 = v3load(t->e1), e2 = v3load(t->e2), vert = v3load(t->vert);
	v3 o = v3load(origin), d = v3load(direction);
	v3 pvec = v3cross(d, e2);
	float det = v3dot(e1, pvec);
	if (det < 0.0000001f)
		return 0.0f;
	float invDet = 1 / det;
	v3 tvec = v3sub(o, vert);
	float u = v3dot(tvec, pvec) * invDet;
	if (u < 0 || u > 1)
		return 0;
	v3 qvec = v3cross(tvec, e1);
	float v = v3dot(d, qvec) * invDet;
	if (v < 0 || u + v > 1)
		return 0;
	return v3dot(e2, qvec) * invDet;
}

/* bvh.h:118-161; counters follow intersect_debug's rule (bvh.h:164-209): one per loop iteration */
int orc_bvh_intersect(const orc_node* nodes, const orc_triangle* prims, orc_ray* ray, uint64_t* counters) {
	int hit = 0;
	float invDir[3] = { 1.f / ray->direction[0], 1.f / ray->direction[1], 1.f / ray->direction[2] };
	int dirIsNeg[3] = { invDir[0] < 0, invDir[1] < 0, invDir[2] < 0 };
	int toVisitOffset = 0, currentNodeIndex = 0;
	int nodesToVisit[64];
	uint64_t nn = 0, nt = 0, entered = 0;
	for (;;) {
		const orc_node* node = &nodes[currentNodeIndex];
		++nn;
		if (orc_bbox_intersect(&node->bbox, ray->origin, invDir, dirIsNeg, ray->distance)) {
			if (nn == 1)
				entered = 1; /* the root's box: this ray enters the tree */
			if (node->primitiveCount > 0) {
				for (int i = 0; i < node->primitiveCount; ++i) {
					++nt;
					float t = orc_triangle_intersect(&prims[node->offset + i], ray->origin, ray->direction);
					if (t > ORC_EPSILON && t < ray->distance && ((ray->distance - t) > ORC_EPSILON)) {
						ray->identifier = node->offset + i;
						ray->distance = t;
						hit = 1;
					}
				}
				if (toVisitOffset == 0)
					break;
				currentNodeIndex = nodesToVisit[--toVisitOffset];
			} else {
				if (dirIsNeg[node->splitAxis]) {
					nodesToVisit[toVisitOffset++] = currentNodeIndex + 1;
					currentNodeIndex = node->offset;
				} else {
					nodesToVisit[toVisitOffset++] = node->offset;
					currentNodeIndex = currentNodeIndex + 1;
				}
			}
		} else {
			if (toVisitOffset == 0)
				break;
			currentNodeIndex = nodesToVisit[--toVisitOffset];
		}
	}
	if (counters) {
		counters[0] += nn;
		counters[1] += nt;
		counters[2] += entered;
	}
	return hit;
}

/* the same over a batch of 60-byte RayQueue records, updated in place (bench.py times this beside the reference's own loop) */
void orc_bvh_intersect_batch(const orc_node* nodes, const orc_triangle* prims, orc_ray* rays, int n, int* hit_out) {
	for (int i = 0; i < n; ++i)
		hit_out[i] = orc_bvh_intersect(nodes, prims, &rays[i], NULL);
}

/* bvh.h:213-256 */
int orc_bvh_intersect_simple(const orc_node* nodes, const orc_triangle* prims, const orc_shadow* ray, float closestAllowed, uint64_t* counters) {
	float closestIntersection = closestAllowed;
	float invDir[3] = { 1.f / ray->direction[0], 1.f / ray->direction[1], 1.f / ray->direction[2] };
	int dirIsNeg[3] = { invDir[0] < 0, invDir[1] < 0, invDir[2] < 0 };
	int toVisitOffset = 0, currentNodeIndex = 0;
	int nodesToVisit[64];
	uint64_t nn = 0, nt = 0, entered = 0;
	int result = 0;
	for (;;) {
		const orc_node* node = &nodes[currentNodeIndex];
		++nn;
		if (orc_bbox_intersect(&node->bbox, ray->origin, invDir, dirIsNeg, closestIntersection)) {
			if (nn == 1)
				entered = 1;
			if (node->primitiveCount > 0) {
				for (int i = 0; i < node->primitiveCount; ++i) {
					++nt;
					float t = orc_triangle_intersect(&prims[node->offset + i], ray->origin, ray->direction);
					if (t > ORC_EPSILON && ((closestIntersection - t) > ORC_EPSILON)) {
						result = 1;
						goto done;
					}
				}
				if (toVisitOffset == 0)
					break;
				currentNodeIndex = nodesToVisit[--toVisitOffset];
			} else {
				if (dirIsNeg[node->splitAxis]) {
					nodesToVisit[toVisitOffset++] = currentNodeIndex + 1;
					currentNodeIndex = node->offset;
				} else {
					nodesToVisit[toVisitOffset++] = node->offset;
					currentNodeIndex = currentNodeIndex + 1;
				}
			}
		} else {
			if (toVisitOffset == 0)
				break;
			currentNodeIndex = nodesToVisit[--toVisitOffset];
		}
	}
done:
	if (counters) {
		counters[0] += nn;
		counters[1] += nt;
		counters[2] += entered;
	}
	return result;
}

/* kernel.cu:83-93 (intersect_simple 95-105 is the same arithmetic on a ShadowQueue) */
float orc_sphere_intersect(const orc_sphere* s, const float origin[3], const float direction[3]) {
	v3 op = v3sub(v3load(s->position), v3load(origin));
	float t;
	float b = v3dot(op, v3load(direction));
	float disc = b * b - v3dot(op, op) + s->radius * s->radius;
	if (disc < 0)
		return 0;
	disc = sqrtf(disc);
	return (t = b - disc) > ORC_EPSILON ? t : ((t = b + disc) > ORC_EPSILON ? t : 0);
}
