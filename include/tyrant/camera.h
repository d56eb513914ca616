// tyrant/camera.h -- struct Camera (camera.h:3-22) and the global `camera` (camera.h:24).
// handle_input (camera.cpp:3-44) is GLFW keyboard/mouse code and is not part of the render path.
#pragma once
#include "../tyr_c.h"
#include "variables.h"
namespace tyrant {
struct Camera {
	vec3 position = { 1, 30, 90 };
	vec3 direction = { 1, 0, 0 };
	vec3 up = { 0, 0, 1 };
	float focalDistance = 1;
	float lensRadius = 0.0f;
	double horizontal_angle = 0.0;
	double vertical_angle = 0.0;
	void update() { // camera.cpp:46-52
		float d[3];
		tyr_camera_update(horizontal_angle, vertical_angle, d);
		direction = { d[0], d[1], d[2] };
	}
};
extern Camera camera;
} // namespace tyrant
