// tyrant/Scene.h -- class Scene and Scene::GPUScene (Scene.h:3-18).  Load() keeps the second half
// of Scene::Load (Scene.cpp:20-67: per-face Triangle + BBox, BVH build, upload); the assimp
// import of its first half is a "next" row, so triangles are handed in.
#pragma once
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "bvh.h"

namespace tyrant {
class Scene {
public:
	struct GPUScene {
		CachedBVH CUDACachedBVH; // the reference's member name, Scene.h:6
	} gpuScene;

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
		BVH bvh(primitives, primitiveBBoxes, PartitionAlgorithm::SAH); // Scene.cpp:53
		const int rc = tyr_scene_upload(ctx, reinterpret_cast<const tyr_bvh_node*>(bvh.nodes.data()), bvh.nNodes, reinterpret_cast<const tyr_triangle*>(primitives.data()),
			static_cast<int32_t>(primitives.size()));
		if (rc)
			throw std::runtime_error(std::string("Scene::Load: ") + tyr_status_string(rc));
	}

private:
	std::vector<Triangle> primitives;
	std::vector<BBox> primitiveBBoxes;
};
} // namespace tyrant
