/*
 * oracle/orc_bvh.c -- CPU restatement of the host binned-SAH BVH builder
 * (bvh.cpp:3-225, bvh.h:49-108) and of the BBox host operations
 * (Bbox.h:3-36, Bbox.cpp:3-14).  TEST INFRASTRUCTURE (see orc.h).
 *
 * The reference partitions with std::partition (bvh.cpp:171-176); its element
 * order is the classic bidirectional two-pointer swap (libstdc++ and MSVC agree),
 * restated here explicitly so the node array is toolchain independent
 * ("Third-party arithmetic").
 */
#include <stdlib.h>

#include "orc_internal.h"

#define BUCKET_NUMBER 14      /* bvh.h:76 */
#define MAX_PRIM_NUMBER 4     /* bvh.h:78 */
#define TRAVERSAL_COST 1.0f   /* bvh.h:81 */
#define INTERSECTION_COST 1.0f /* bvh.h:84 */

/* glibc fmin/fmax return the first argument on ties; NaN inputs are rejected upstream */
static inline float bmin(float a, float b) { return (b < a) ? b : a; }
static inline float bmax(float a, float b) { return (b > a) ? b : a; }

static orc_bbox bbox_empty(void) {
	/* Bbox.h:5 */
	orc_bbox b = { { { 1e10f, 1e10f, 1e10f }, { -1e10f, -1e10f, -1e10f } } };
	return b;
}
/* Bbox.h:8-14 */
static void bbox_add_vertex(orc_bbox* b, const float v[3]) {
	for (int k = 0; k < 3; ++k) {
		b->bounds[0][k] = bmin(b->bounds[0][k], v[k]);
		b->bounds[1][k] = bmax(b->bounds[1][k], v[k]);
	}
}
/* Bbox.cpp:3-14 */
static orc_bbox bbox_union(const orc_bbox* b1, const orc_bbox* b2) {
	orc_bbox r;
	for (int k = 0; k < 3; ++k) {
		r.bounds[0][k] = bmin(b1->bounds[0][k], b2->bounds[0][k]);
		r.bounds[1][k] = bmax(b1->bounds[1][k], b2->bounds[1][k]);
	}
	return r;
}
/* Bbox.h:18-21 */
static float bbox_surface_area(const orc_bbox* b) {
	float dx = b->bounds[1][0] - b->bounds[0][0];
	float dy = b->bounds[1][1] - b->bounds[0][1];
	float dz = b->bounds[1][2] - b->bounds[0][2];
	return 2 * (dx * dy + dx * dz + dy * dz);
}
/* Bbox.h:28-36 */
static int bbox_largest_extent(const orc_bbox* b) {
	float dx = b->bounds[1][0] - b->bounds[0][0];
	float dy = b->bounds[1][1] - b->bounds[0][1];
	float dz = b->bounds[1][2] - b->bounds[0][2];
	if (dx > dy && dx > dz)
		return 0;
	else if (dy > dz)
		return 1;
	else
		return 2;
}

/* Scene.cpp:22-33: bbox of the three vertices of a {vert,e1,e2} triangle.
 * Scene::Load has the vertices; from the stored form they are vert, vert+e1, vert+e2. */
/* the host BBox operations over n vertices (Bbox.h:8-36): bbox_out = the box, out2 = {surfaceArea, largestExtent};
 * same shape as oracle/ref_harness.cpp ref_bbox_host_ops, which runs the reference's own Bbox.h */
void orc_bbox_host_ops(const float* vertices, int n, orc_bbox* bbox_out, float* out2) {
	orc_bbox b = bbox_empty();
	for (int i = 0; i < n; ++i)
		bbox_add_vertex(&b, vertices + 3 * i);
	*bbox_out = b;
	out2[0] = bbox_surface_area(&b);
	out2[1] = (float)bbox_largest_extent(&b);
}

void orc_triangle_bbox(const orc_triangle* t, orc_bbox* out) {
	float v1[3], v2[3];
	for (int k = 0; k < 3; ++k) {
		v1[k] = t->vert[k] + t->e1[k];
		v2[k] = t->vert[k] + t->e2[k];
	}
	*out = bbox_empty();
	bbox_add_vertex(out, t->vert);
	bbox_add_vertex(out, v1);
	bbox_add_vertex(out, v2);
}

/* bvh.h:88-97 */
typedef struct {
	uint32_t primitiveNumber;
	orc_bbox bbox;
	float centroid[3];
} prim_info;

typedef struct {
	orc_node* nodes;
	int nNodes;
	prim_info* info;
	const orc_triangle* prims;
	orc_triangle* ordered;
	int nOrdered;
	int algo;
} builder;

/* bvh.cpp:44-58 */
static int compute_bucket(const prim_info* p, const float cb[3], const float ct[3], int dim) {
	float distance = p->centroid[dim] - cb[dim];
	if (ct[dim] > cb[dim]) {
		distance = distance / (ct[dim] - cb[dim]);
	}
	int bucket_idx = (int)(BUCKET_NUMBER * distance);
	if (bucket_idx == BUCKET_NUMBER) {
		bucket_idx--;
	}
	return bucket_idx;
}

/* bvh.cpp:214-218 */
static void init_leaf(orc_node* n, int first, int count, const orc_bbox* box) {
	n->bbox = *box;
	n->offset = first;
	n->primitiveCount = (uint16_t)count;
}

static void emit_leaf(builder* B, int node, int start, int end, const orc_bbox* box) {
	int first = B->nOrdered;
	for (int i = start; i < end; ++i) {
		B->ordered[B->nOrdered++] = B->prims[B->info[i].primitiveNumber];
	}
	init_leaf(&B->nodes[node], first, end - start, box);
}

static int cmp_dim;
static int cmp_centroid(const void* a, const void* b) {
	float ca = ((const prim_info*)a)->centroid[cmp_dim], cb = ((const prim_info*)b)->centroid[cmp_dim];
	return (ca < cb) ? -1 : (ca > cb);
}

/* bvh.cpp:61-212 */
static void recursive_build(builder* B, int start, int end) {
	int currentNode = B->nNodes++;

	orc_bbox nodeBBox = bbox_empty();
	for (int i = start; i < end; ++i) {
		nodeBBox = bbox_union(&nodeBBox, &B->info[i].bbox);
	}
	int nPrimitives = end - start;
	if (nPrimitives == 1) {
		emit_leaf(B, currentNode, start, end, &nodeBBox);
		return;
	}

	orc_bbox centroidBBox = bbox_empty();
	for (int i = start; i < end; ++i) {
		bbox_add_vertex(&centroidBBox, B->info[i].centroid);
	}
	int dim = bbox_largest_extent(&centroidBBox);
	int mid = (start + end) / 2;
	const float* cb = centroidBBox.bounds[0];
	const float* ct = centroidBBox.bounds[1];

	if (cb[dim] == ct[dim]) {
		emit_leaf(B, currentNode, start, end, &nodeBBox);
		return;
	}

	if (B->algo == 1) {
		/* EqualCounts (bvh.cpp:108-116).  std::nth_element's permutation is
		 * implementation defined; a full sort satisfies its postcondition. */
		cmp_dim = dim;
		qsort(&B->info[start], (size_t)(end - start), sizeof(prim_info), cmp_centroid);
	} else {
		struct {
			int count;
			orc_bbox bounds;
		} buckets[BUCKET_NUMBER];
		for (int b = 0; b < BUCKET_NUMBER; ++b) {
			buckets[b].count = 0;
			buckets[b].bounds = bbox_empty();
		}
		for (int i = start; i < end; ++i) {
			int b = compute_bucket(&B->info[i], cb, ct, dim);
			buckets[b].count++;
			buckets[b].bounds = bbox_union(&buckets[b].bounds, &B->info[i].bbox);
		}
		float min_split_cost = 3.402823466e+38f; /* FLT_MAX */
		int min_split_bucket = -1;
		float nodeSA = bbox_surface_area(&nodeBBox);
		for (int cur = 0; cur < BUCKET_NUMBER - 1; ++cur) {
			int c1 = 0, c2 = 0;
			orc_bbox b1 = bbox_empty(), b2 = bbox_empty();
			for (int i = 0; i <= cur; ++i) {
				b1 = bbox_union(&b1, &buckets[i].bounds);
				c1 += buckets[i].count;
			}
			for (int i = cur + 1; i < BUCKET_NUMBER; ++i) {
				b2 = bbox_union(&b2, &buckets[i].bounds);
				c2 += buckets[i].count;
			}
			float cost = TRAVERSAL_COST + ((float)c1 * bbox_surface_area(&b1) + (float)c2 * bbox_surface_area(&b2)) / nodeSA;
			if (cost < min_split_cost) {
				min_split_cost = cost;
				min_split_bucket = cur;
			}
		}
		float leaf_cost = INTERSECTION_COST * (float)nPrimitives;
		if (min_split_bucket < 0) {
			/* bvh.cpp:167 assert(min_split_bucket != -1): zero-area node box; keep the tree valid with a leaf */
			emit_leaf(B, currentNode, start, end, &nodeBBox);
			return;
		}
		if (nPrimitives > MAX_PRIM_NUMBER || min_split_cost < leaf_cost) {
			/* std::partition(first, last, bucket <= min_split_bucket), bidirectional form */
			int first = start, last = end;
			for (;;) {
				for (;;) {
					if (first == last)
						goto partitioned;
					if (compute_bucket(&B->info[first], cb, ct, dim) <= min_split_bucket)
						++first;
					else
						break;
				}
				--last;
				for (;;) {
					if (first == last)
						goto partitioned;
					if (!(compute_bucket(&B->info[last], cb, ct, dim) <= min_split_bucket))
						--last;
					else
						break;
				}
				prim_info tmp = B->info[first];
				B->info[first] = B->info[last];
				B->info[last] = tmp;
				++first;
			}
		partitioned:
			mid = first;
		} else {
			emit_leaf(B, currentNode, start, end, &nodeBBox);
			return;
		}
	}

	recursive_build(B, start, mid);
	B->nodes[currentNode].offset = B->nNodes; /* secondChildOffset, bvh.cpp:203 */
	int second = B->nNodes;
	recursive_build(B, mid, end);
	/* initInterior, bvh.cpp:220-225 */
	B->nodes[currentNode].bbox = bbox_union(&B->nodes[currentNode + 1].bbox, &B->nodes[second].bbox);
	B->nodes[currentNode].primitiveCount = 0;
	B->nodes[currentNode].splitAxis = (uint8_t)dim;
}

int orc_bvh_build(orc_triangle* prims, int n, const orc_bbox* bboxes, orc_node* nodes_out, int algo) {
	if (n <= 0)
		return 0; /* bvh.cpp:8-10 */
	if (algo != 1 && algo != 2)
		return -1;
	builder B;
	memset(nodes_out, 0, sizeof(orc_node) * (size_t)(2 * n - 1)); /* vector::resize value-initialises (bvh.cpp:11) */
	B.nodes = nodes_out;
	B.nNodes = 0;
	B.info = (prim_info*)malloc(sizeof(prim_info) * (size_t)n);
	B.ordered = (orc_triangle*)malloc(sizeof(orc_triangle) * (size_t)n);
	B.prims = prims;
	B.nOrdered = 0;
	B.algo = algo;
	if (!B.info || !B.ordered) {
		free(B.info);
		free(B.ordered);
		return -2;
	}
	for (int i = 0; i < n; ++i) {
		/* PrimitiveInfo ctor, bvh.h:93-96: centroid = bounds[0]*0.5f + bounds[1]*0.5f */
		B.info[i].primitiveNumber = (uint32_t)i;
		B.info[i].bbox = bboxes[i];
		for (int k = 0; k < 3; ++k) {
			B.info[i].centroid[k] = bboxes[i].bounds[0][k] * 0.5f + bboxes[i].bounds[1][k] * 0.5f;
		}
	}
	recursive_build(&B, 0, n);
	memcpy(prims, B.ordered, sizeof(orc_triangle) * (size_t)n); /* bvh.cpp:24 */
	free(B.info);
	free(B.ordered);
	return B.nNodes;
}
