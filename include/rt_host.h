/*
 * rt_host.h — C entry points of the HOST side: the reference's scene vocabulary (Object /
 * Material / Texture classes mirroring src/objects.rs, src/materials.rs, src/texture.rs,
 * src/transforms.rs, src/bvh.rs), its scene constructors (src/scenes.rs) and the flattener that
 * turns a scene graph into the pointer-free RtSceneDesc of rt_hip.h.
 *
 * In the north-star layout this layer is the Rust host crate; no Rust toolchain exists in this
 * image, so it is C++ (ray-tracing_amd/host/) behind this small C API, which is what the Python
 * launcher, bench and tests drive.  It never traces a ray: there is no CPU render path here.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include <stdint.h>
#include "rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct RthScene RthScene; /* owns the scene graph and the flattened arrays */

const char* rth_last_error(void);

/* Build one of the reference's scenes (src/scenes.rs): "random_scene" :7-110,
 * "random_scene_light" :171-333, "two_checker_spheres" :399-444, "two_perlin_spheres" :445-482,
 * "earth" :483-507, "simple_light" :508-558, "cornell_box" :559-702, "cornell_smoke" :703-840,
 * "final_scene" :841-1014.  The reference seeds from entropy; here every random draw (sphere
 * placement, BVH split axes, Perlin tables) comes from `seed`.  image_path: the earth map (baseline
 * JPEG or binary PPM) for the two scenes that use ImageTexture; may be NULL otherwise.
 * Returns NULL on error (rth_last_error). */
RthScene* rth_scene_build(const char* name, double aspect_ratio, uint64_t seed, const char* image_path);
void rth_scene_free(RthScene* scene);

/* The flattened scene, valid until rth_scene_free. Pass to rt_scene_create(). */
const RtSceneDesc* rth_scene_desc(const RthScene* scene);

/* Number of leaf primitives of the WORLD graph (pre-order ids 0..n-1; light-list primitives are
 * stored after them). */
uint32_t rth_scene_num_world_prims(const RthScene* scene);

/* Flattening round trip: rebuild a scene graph from the flat description and compare it with the
 * original graph bit-for-bit (every f64 field, topology, order). 0 = identical; otherwise the
 * number of differences (rth_last_error describes the first). */
int rth_scene_roundtrip_check(const RthScene* scene);

/* What ImageTexture::new does with its file (src/texture.rs:68-74: image::open(filename).to_rgb()): decode a
 * baseline JPEG (the reference's images/earthmap.jpg) or an 8-bit binary PPM into RGB8, row 0 on top.
 * *rgb is malloc'ed (width*height*3 bytes); release it with rth_image_free.  0 on success, -1 on error. */
int rth_image_load(const char* path, uint8_t** rgb, uint32_t* width, uint32_t* height);
void rth_image_free(uint8_t* rgb);

/* Names of the scenes rth_scene_build knows, NULL-terminated. */
const char* const* rth_scene_names(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
