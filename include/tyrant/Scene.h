// tyrant/Scene.h -- class Scene and Scene::GPUScene (Scene.h:3-18).  Load() keeps the second half
// of Scene::Load (Scene.cpp:20-67: per-face Triangle + BBox, BVH build, upload); the assimp
// import of its first half is a "next" row, so triangles are handed in.
#pragma once
#include <iostream>
#include <vector>

#include "bvh.h"

namespace tyrant {
class Scene {
public:
	struct GPUScene {
		CachedBVH CUDACachedBVH; // the reference's member name, Scene.h:6
	} gpuScene;

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
