// tyrant/bvh.h -- class BVH (bvh.h:49-108) and CachedBVH (bvh.h:111-117) over the C ABI.
// The constructor runs the binned-SAH build (bvh.cpp:3-225) inside libtyrant_hip.so and, like
// the reference, reorders `primitives` in place.
#pragma once
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../tyr_c.h"
#include "Bbox.h"
#include "loader.h"

namespace tyrant {

enum class PartitionAlgorithm { Middle, EqualCounts, SAH }; // bvh.h:45-47

// Where `BVH`'s constructor builds (SAH only): -1 (default) = on the host's threads (tyr_bvh_build), d >= 0 = on device d
// (tyr_bvh_build_device).  Either way the nodes and the primitive order are the reference's bvh.cpp's, byte for byte; a device
// build that cannot finish (TYR_ERR_UNSUPPORTED: a degenerate range beyond what one thread's stack holds) falls back to the host.
// With a build device set, Scene::Load (Scene.h) goes one step further: tyr_scene_build_upload, the tree built AND laid out on the ctx's device.
inline int& build_device_slot() {
	static int device = -1; // (inline function: one object for the whole program)
	return device;
}
inline void set_build_device(int device) { build_device_slot() = device; }

class BVH {
public:
	struct BVHNode { // bvh.h:55-68 (32 B)
		BBox bbox;
		union {
			int primitiveOffset;
			int secondChildOffset;
		};
		uint16_t primitiveCount;
		uint8_t splitAxis;
		char pad[1];
	};
	static_assert(sizeof(BVHNode) == sizeof(tyr_bvh_node), "BVHNode layout");

	BVH(std::vector<Triangle>& primitives, std::vector<BBox> primitivesBBoxes, PartitionAlgorithm partitionAlgo) : partitionAlgorithm(partitionAlgo) {
		std::cout << "Creating BVH, total primitives: " << primitives.size() << "\n"; // bvh.cpp:7
		if (primitives.empty())
			return;
		nodes.resize(2 * primitives.size() - 1);
		int rc = TYR_ERR_UNSUPPORTED;
		if (partitionAlgo == PartitionAlgorithm::SAH && build_device_slot() >= 0)
			rc = tyr_bvh_build_device(build_device_slot(), reinterpret_cast<tyr_triangle*>(primitives.data()), static_cast<int32_t>(primitives.size()),
				reinterpret_cast<const tyr_bbox*>(primitivesBBoxes.data()), reinterpret_cast<tyr_bvh_node*>(nodes.data()), nullptr);
		if (rc == TYR_ERR_UNSUPPORTED)
			rc = tyr_bvh_build(reinterpret_cast<tyr_triangle*>(primitives.data()), static_cast<int32_t>(primitives.size()),
				reinterpret_cast<const tyr_bbox*>(primitivesBBoxes.data()), reinterpret_cast<tyr_bvh_node*>(nodes.data()), static_cast<int32_t>(partitionAlgo));
		if (rc < 0) // Middle is unimplemented in the reference too (bvh.cpp:190-193 prints an error)
			throw std::invalid_argument(std::string("BVH: ") + tyr_status_string(rc));
		nNodes = rc;
		std::cout << "Created BVH, total nodes : " << nNodes << "\n"; // bvh.cpp:27
	}
	~BVH() = default;

	std::vector<BVHNode> nodes;
	int nNodes = 0;
	const PartitionAlgorithm partitionAlgorithm = PartitionAlgorithm::SAH;
};

// bvh.h:111-117: in the reference this holds raw device pointers and is passed by value to the
// kernels; here the device copies live in the tyr_ctx (private layout), so it is a handle.
class CachedBVH {
public:
	CachedBVH() = default;
	tyr_ctx* ctx = nullptr;
};

} // namespace tyrant
