// host.hpp -- host-side pieces of libtyrant_hip.so (C++17, compiled by hipcc as host code)
#pragma once

#include <cstdint>
#include <vector>

#include "../../../include/tyr_c.h"
#include "../hip/kernels.hpp"
#include "../hip/sunsky.hpp"
#include "../hip/traverse.hpp"
#include "../hip/vecmath.hpp"

namespace tyr {

// host/sun_setup.cpp
void sun_setup(float sun_x, float sun_y, SunParams& out);

// host/bvh_build.cpp -- class BVH of the reference (bvh.h:49-108, bvh.cpp:3-225)
int bvh_build(tyr_triangle* prims, int32_t n, const tyr_bbox* bboxes, tyr_bvh_node* nodes_out, int32_t algo);
void triangle_bboxes(const tyr_triangle* prims, int32_t n, tyr_bbox* out);
void set_build_threads(int threads); // 0 = automatic (TYR_BUILD_THREADS or min(16, cores))

// host/bvh_layout.cpp -- flat reference nodes -> device pair nodes + 48-byte triangles
struct DeviceLayout {
	std::vector<float> pairNodes; // 16 floats per pair node
	std::vector<float> quadNodes; // 32 floats per quad node (hip/traverse.hpp "QuadNode")
	uint32_t quadRootRef = 0;
	uint32_t nQuads = 0;
	uint32_t nStaged = 0; // leading records that are the top of the tree in breadth-first order
	uint32_t quadMaxStack = 0; // the most entries a traversal of the quad tree can ever have on its stack (whatever the ray)
	std::vector<float> tris;      // 12 floats per triangle
	float rootMin[3], rootMax[3];
	uint32_t rootRef;
	uint32_t nPairs;
};
// returns TYR_OK or TYR_ERR_INVALID (malformed tree, non-finite geometry, too many primitives)
int build_device_layout(const tyr_bvh_node* nodes, int32_t nNodes, const tyr_triangle* prims, int32_t nPrims, DeviceLayout& out);

} // namespace tyr
