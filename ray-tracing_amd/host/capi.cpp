// capi.cpp — include/rt_host.h: build a reference scene, flatten it, hand out the RtSceneDesc.
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/rt_host.h"
#include "flatten.hpp"

struct RthScene {
    rth::Scene graph;
    rth::FlatScene flat;
};

namespace {
thread_local std::string g_error;
const char* const kSceneNames[] = {"random_scene", "random_scene_light", "two_checker_spheres", "two_perlin_spheres", "earth",
                                   "simple_light", "cornell_box", "cornell_smoke", "final_scene", nullptr};
}  // namespace

extern "C" {

const char* rth_last_error(void) { return g_error.c_str(); }
const char* const* rth_scene_names(void) { return kSceneNames; }

RthScene* rth_scene_build(const char* name, double aspect_ratio, uint64_t seed, const char* image_path) {
    if (!name) { g_error = "scene name is NULL"; return nullptr; }
    const std::string n = name, img = image_path ? image_path : "";
    RthScene* s = new RthScene();
    try {
        if (n == "random_scene") s->graph = rth::random_scene(aspect_ratio, seed);
        else if (n == "random_scene_light") s->graph = rth::random_scene_light(aspect_ratio, seed);
        else if (n == "two_checker_spheres") s->graph = rth::two_checker_spheres(aspect_ratio);
        else if (n == "two_perlin_spheres") s->graph = rth::two_perlin_spheres(aspect_ratio, seed);
        else if (n == "earth") s->graph = rth::earth(aspect_ratio, img);
        else if (n == "simple_light") s->graph = rth::simple_light(aspect_ratio, seed);
        else if (n == "cornell_box") s->graph = rth::cornell_box(aspect_ratio);
        else if (n == "cornell_smoke") s->graph = rth::cornell_smoke(aspect_ratio);
        else if (n == "final_scene") s->graph = rth::final_scene(aspect_ratio, seed, img);
        else throw std::runtime_error("unknown scene: " + n);
        rth::flatten(s->graph, s->flat);
    } catch (const std::exception& e) {
        g_error = e.what();
        delete s;
        return nullptr;
    }
    return s;
}

void rth_scene_free(RthScene* scene) { delete scene; }

const RtSceneDesc* rth_scene_desc(const RthScene* scene) { return scene ? &scene->flat.desc : nullptr; }

uint32_t rth_scene_num_world_prims(const RthScene* scene) { return scene ? scene->flat.n_world_prims : 0; }

int rth_scene_roundtrip_check(const RthScene* scene) {
    if (!scene) { g_error = "scene is NULL"; return -1; }
    try {
        rth::Scene back = rth::unflatten(scene->flat.desc);
        std::string first;
        int diffs = rth::compare_scenes(scene->graph, back, first);
        if (diffs) g_error = "first difference: " + first;
        // and the flat form of the rebuilt graph must be byte-identical to the original flat form
        rth::FlatScene again;
        rth::flatten(back, again);
        const rth::FlatScene& f = scene->flat;
        auto same = [](const auto& a, const auto& b) { return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(a[0])) == 0); };
        if (!same(f.prims, again.prims) || !same(f.materials, again.materials) || !same(f.xforms, again.xforms) || !same(f.world, again.world) ||
            !same(f.lights, again.lights) || !same(f.min_x, again.min_x) || !same(f.min_y, again.min_y) || !same(f.min_z, again.min_z) ||
            !same(f.max_x, again.max_x) || !same(f.max_y, again.max_y) || !same(f.max_z, again.max_z) || !same(f.left, again.left) ||
            !same(f.right, again.right) || !same(f.perlins, again.perlins) || f.textures.size() != again.textures.size()) {
            if (!diffs) g_error = "re-flattened arrays differ from the original flat form";
            diffs++;
        }
        return diffs;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}

// Decode an image file the way ImageTexture::new does (texture.rs:68-74): baseline JPEG or binary PPM -> RGB8.
int rth_image_load(const char* path, uint8_t** rgb, uint32_t* width, uint32_t* height) {
    if (!path || !rgb || !width || !height) { g_error = "NULL argument"; return -1; }
    try {
        std::shared_ptr<const rth::RgbImage> im = rth::load_image(path);
        *rgb = (uint8_t*)std::malloc(im->rgb.size() ? im->rgb.size() : 1);
        if (!*rgb) { g_error = "out of memory"; return -1; }
        std::memcpy(*rgb, im->rgb.data(), im->rgb.size());
        *width = im->width; *height = im->height;
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}
void rth_image_free(uint8_t* rgb) { std::free(rgb); }

}  // extern "C"
