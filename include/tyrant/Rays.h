// tyrant/Rays.h -- Rays.h:5-11.  Declared by the reference and used by none of its kernels;
// kept so host code that names it still compiles.
#pragma once
#include "variables.h"
namespace tyrant {
struct Ray {
	vec3 orig;
	vec3 dir;
	Ray(vec3 origin, vec3 direction) : orig(origin), dir(direction) {}
};
} // namespace tyrant
