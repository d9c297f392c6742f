// flatten.hpp — scene graph -> pointer-free RtSceneDesc (include/rt_hip.h) and back.
#pragma once
#include <map>
#include <vector>

#include "../../include/rt_hip.h"
#include "scene_api.hpp"

namespace rth {

// Owns every array an RtSceneDesc points into.
struct FlatScene {
    std::vector<RtPrim> prims;
    std::vector<RtMaterial> materials;
    std::vector<RtTexture> textures;
    std::vector<RtPerlin> perlins;
    std::vector<RtImage> images;
    std::vector<std::shared_ptr<const RgbImage>> image_owners;
    std::vector<RtXform> xforms;
    std::vector<RtInstance> world, lights;
    std::vector<double> min_x, min_y, min_z, max_x, max_y, max_z;  // SoA BVH
    std::vector<int32_t> left, right;
    uint32_t n_world_prims = 0;
    RtSceneDesc desc{};
    void seal();  // (re)point desc at the vectors
};

// Flattening rules (DESIGN.md "flat scene form"):
//  * world / lights: one RtInstance per ObjectList entry, in order.
//  * an entry is  [ConstantMedium(] Translate|RotateY ... ( primitive | Bvh | BvhNode ) [)]
//    wrappers become the instance's xform chain, outermost first.
//  * BvhNode trees: internal nodes in pre-order into the SoA arrays; a leaf must be a primitive.
//  * primitive ids = order of first visit (pre-order, left before right) — the id the parity
//    probe reports.  Light-list primitives are appended after the world's.
//  * anything else (nested lists, a medium under a transform, transformed BVH leaves) throws
//    std::runtime_error("unsupported ...").
void flatten(const Scene& scene, FlatScene& out);

// Rebuild a scene graph from a flat description (used by the round-trip check and tests).
Scene unflatten(const RtSceneDesc& d);

// Structural + bit-exact comparison of two scene graphs; returns the number of differences and
// describes the first in `first`.
int compare_scenes(const Scene& a, const Scene& b, std::string& first);

}  // namespace rth
