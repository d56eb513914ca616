// device_build.hpp -- what hip/bvh_build_dev.hip leaves on the device for hip/bvh_layout_dev.hip (a header of its own: the builder's
// unit keeps the reference's builder constants under names host.hpp's traversal constants also use)
#pragma once

#include <cstdint>

#include "../../../include/tyr_c.h"

namespace tyr {

struct DeviceBuild { // one device allocation, freed by the destructor
	void* pool = nullptr;
	tyr_bvh_node* nodes = nullptr; // nNodes records, the reference's bytes
	tyr_triangle* prims = nullptr; // n records in their final order
	int32_t nNodes = 0, n = 0;
	DeviceBuild() = default;
	DeviceBuild(const DeviceBuild&) = delete;
	DeviceBuild& operator=(const DeviceBuild&) = delete;
	~DeviceBuild();
};
// the SAH build with its results left on the device; prims / bboxes are host arrays and are not touched.  Returns the node count
// (> 0) or a negative status; seconds_out (may be null): [0] the device's work, [1] the copies in
int bvh_build_device_keep(int device, const tyr_triangle* prims, int32_t n, const tyr_bbox* bboxes, DeviceBuild& out, double* seconds_out);

} // namespace tyr
