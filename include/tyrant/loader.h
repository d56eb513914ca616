// tyrant/loader.h -- struct Triangle (loader.h:13-19, 40 B).  Triangle::intersect (loader.h:21-46)
// runs on the device inside libtyrant_hip.so (hip/traverse.hpp).
#pragma once
#include "../tyr_c.h"
#include "variables.h"
namespace tyrant {
struct Triangle {
	vec3 vert;
	vec3 e1, e2;
	uint8_t materialType{};
};
static_assert(sizeof(Triangle) == sizeof(tyr_triangle), "Triangle layout");
} // namespace tyrant
