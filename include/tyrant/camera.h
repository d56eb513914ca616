// tyrant/camera.h -- struct Camera (camera.h:3-22) and the global `camera` (camera.h:24).
// handle_input (camera.cpp:3-44) reads a GLFW window in the reference; here it takes the state it would have read
// (tyr_input_state: keys, cursor, window size) -- a pure function, usable on a headless node (examples/flythrough.cpp).
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
	void handle_input(const tyr_input_state& input, double delta) { // camera.cpp:3-44
		tyr_camera_pose p{ { position.x, position.y, position.z }, { direction.x, direction.y, direction.z }, { up.x, up.y, up.z }, horizontal_angle, vertical_angle };
		tyr_camera_handle_input(&p, &input, delta);
		position = { p.position[0], p.position[1], p.position[2] };
		horizontal_angle = p.horizontal_angle;
		vertical_angle = p.vertical_angle;
	}
	void update() { // camera.cpp:46-52
		float d[3];
		tyr_camera_update(horizontal_angle, vertical_angle, d);
		direction = { d[0], d[1], d[2] };
	}
};
extern Camera camera;
} // namespace tyrant
