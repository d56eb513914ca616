// tyrant/Scene.h -- class Scene and Scene::GPUScene (Scene.h:3-18).  Load(path) is the reference's call (Scene.h:9): PLY files
// through tyr_load_ply with Scene::Load's conventions (Scene.cpp:3-47), then the second half of Scene::Load (Scene.cpp:20-67:
// per-face Triangle + BBox, BVH build, upload).  Other formats: the host imports and hands the per-face vertices in.
#pragma once
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "bvh.h"

namespace tyrant {
// The context the handle-less calls use.  The reference's Scene::Load(path) (Scene.h:9, called at main.cpp:113) has no handle to
// pass: its device state is process-wide (kernel.cu:211-224, Scene.cpp:55-67).  A host that wants the reference's one-line
// call sets its context once, where main.cpp:89-102 picks the device; the overloads that name a ctx remain for hosts with
// several (one per GPU).
inline tyr_ctx*& default_ctx_slot() {
	static tyr_ctx* ctx = nullptr; // (inline function: one object for the whole program)
	return ctx;
}
inline void set_default_ctx(tyr_ctx* ctx) { default_ctx_slot() = ctx; }
inline tyr_ctx* default_ctx() { return default_ctx_slot(); }

class Scene {
public:
	struct GPUScene {
		CachedBVH CUDACachedBVH; // the reference's member name, Scene.h:6
	} gpuScene;

	// void Load(const char path[]), Scene.h:9 -- the reference's own signature, on the process-default context
	void Load(const char path[]) { Load(require_default_ctx(), path); }
	void Load(const std::vector<vec3>& faceVertices) { Load(require_default_ctx(), faceVertices); }

	// Scene::Load(path), Scene.cpp:3: PLY files through tyr_load_ply (first mesh, fan triangulation)
	void Load(tyr_ctx* ctx, const char path[]) {
		std::cout << "Loading scene:" << path << "\n"; // Scene.cpp:7
		tyr_triangle* t = nullptr;
		const int n = tyr_load_ply(path, &t);
		if (n < 0)
			throw std::runtime_error(std::string("Scene::Load: ") + tyr_status_string(n));
		std::vector<vec3> v;
		v.reserve(static_cast<size_t>(n) * 3);
		for (int i = 0; i < n; ++i) {
			const vec3 a = { t[i].vert[0], t[i].vert[1], t[i].vert[2] };
			v.push_back(a);
			v.push_back({ a.x + t[i].e1[0], a.y + t[i].e1[1], a.z + t[i].e1[2] });
			v.push_back({ a.x + t[i].e2[0], a.y + t[i].e2[1], a.z + t[i].e2[2] });
		}
		tyr_free(t);
		Load(ctx, v);
	}

	// vertices are {v0, v1, v2} per face, as Scene.cpp:25-27 reads them from the mesh
	void Load(tyr_ctx* ctx, const std::vector<vec3>& faceVertices) {
		for (size_t f = 0; f + 2 < faceVertices.size(); f += 3) {
			const vec3 &a = faceVertices[f], &b = faceVertices[f + 1], &c = faceVertices[f + 2];
			BBox bbox;
			bbox.addVertex(a).addVertex(b).addVertex(c); // Scene.cpp:31-35
			primitiveBBoxes.push_back(bbox);
			Triangle t;
			t.vert = a;
			t.e1 = { b.x - a.x, b.y - a.y, b.z - a.z };
			t.e2 = { c.x - a.x, c.y - a.y, c.z - a.z };
			primitives.push_back(t);
		}
		gpuScene.CUDACachedBVH.ctx = ctx;
		if (primitives.empty()) { // Scene.cpp:49-52
			std::cerr << "No primitives found in scene, loading scene without any \n";
			tyr_scene_upload(ctx, nullptr, 0, nullptr, 0);
			return;
		}
		if (build_device_slot() >= 0) {
			// tyrant::set_build_device: Scene.cpp:53 and :55-67 in one call -- the tree is built on the ctx's device and laid out there, the
			// nodes never leave HBM (the reference's `bvh` is a local of this function too: nobody reads it afterwards).  `primitives` comes
			// back in the builder's order (bvh.cpp:24); what the device leaves to the host is done there, the scene is the same bytes.
			std::cout << "Creating BVH, total primitives: " << primitives.size() << "\n"; // bvh.cpp:7
			int32_t nNodes = 0;
			const int rc = tyr_scene_build_upload(ctx, reinterpret_cast<tyr_triangle*>(primitives.data()), static_cast<int32_t>(primitives.size()),
				reinterpret_cast<const tyr_bbox*>(primitiveBBoxes.data()), nullptr, &nNodes, nullptr);
			if (rc)
				throw std::runtime_error(std::string("Scene::Load: ") + tyr_status_string(rc));
			std::cout << "Created BVH, total nodes : " << nNodes << "\n"; // bvh.cpp:27
			return;
		}
		BVH bvh(primitives, primitiveBBoxes, PartitionAlgorithm::SAH); // Scene.cpp:53
		const int rc = tyr_scene_upload(ctx, reinterpret_cast<const tyr_bvh_node*>(bvh.nodes.data()), bvh.nNodes, reinterpret_cast<const tyr_triangle*>(primitives.data()),
			static_cast<int32_t>(primitives.size()));
		if (rc)
			throw std::runtime_error(std::string("Scene::Load: ") + tyr_status_string(rc));
	}

private:
	static tyr_ctx* require_default_ctx() {
		tyr_ctx* ctx = default_ctx();
		if (!ctx)
			throw std::runtime_error("Scene::Load: no context -- call tyrant::set_default_ctx(ctx) once after tyr_create, or use Load(ctx, ...)");
		return ctx;
	}
	std::vector<Triangle> primitives;
	std::vector<BBox> primitiveBBoxes;
};
} // namespace tyrant
