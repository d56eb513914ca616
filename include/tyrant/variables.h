// tyrant/variables.h -- the reference's configuration and queue records (variables.h:3-44) for
// host code that is recompiled against libtyrant_hip.so.  Compile-time render_width/height and
// ray_queue_buffer_size become runtime parameters of tyr_create; the values below are the
// reference's defaults.
#pragma once

#include <cstdint>

namespace tyrant {

// layout-compatible with glm::vec3 / vec2 / vec4 as the reference uses them
struct vec2 {
	float x, y;
};
struct vec3 {
	float x, y, z;
};
struct vec4 {
	float x, y, z, w;
};

constexpr float pi = 3.1415926535897932f; // variables.h:3
constexpr float inv_pi = 1.0f / pi;       // variables.h:4
constexpr unsigned render_width = 1920;   // variables.h:9
constexpr unsigned render_height = 1080;  // variables.h:10
constexpr float epsilon = 0.001f;         // variables.h:14
constexpr unsigned ray_queue_buffer_size = 1048576u * 2; // variables.h:44

enum class GeometryType { Sphere = 0, Triangle = 1 }; // variables.h:20-22

struct RayQueue { // variables.h:24-34 (60 B)
	vec3 origin;
	vec3 direction;
	vec3 direct;
	float distance;
	int identifier;
	int bounces;
	int index;
	GeometryType geometry_type = GeometryType::Triangle;
	bool lastSpecular = true;
};
static_assert(sizeof(RayQueue) == 60, "RayQueue layout");

struct ShadowQueue { // variables.h:36-42 (44 B)
	vec3 origin;
	vec3 direction;
	vec3 color;
	int buffer_index;
	float closestDistance = 1e20f;
};
static_assert(sizeof(ShadowQueue) == 44, "ShadowQueue layout");

// variables.cpp:3-5 (defined in tyrant/interop.h's implementation section)
extern vec2 sun_position;
extern bool sun_position_changed;

} // namespace tyrant
