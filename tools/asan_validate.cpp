// TEST TOOL (host code only): memory-safety fuzz of the boundary's validation + packing + the CPU build of the device templates.
// Named scenes are built by the host library, a deep copy of the flat description is corrupted at random (indices, kinds, counts,
// BVH children, values) and pushed through tests/hostsim's entry points — under AddressSanitizer + UBSan, which the Python-driven
// fuzz (tools/fuzz_validate.py) cannot use.  Build and run:  bash tools/asan_validate.sh [cases] [seed]
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../include/rt_hip.h"
#include "../include/rt_host.h"

extern "C" {
const char* sim_last_error();
void* sim_scene_create(const RtSceneDesc* d, int* rc_out);
void sim_scene_destroy(void* s);
void sim_trace_primary(void* h, uint32_t W, uint32_t H, double time, double xi, uint32_t precision, int force_generic, int32_t* ids, double* ts);
void sim_render(void* h, const RtRenderParams* p, int force_generic, double* accum, uint64_t* casts);
}

namespace {
struct Copy {  // a deep, mutable copy of a description
    RtSceneDesc d;
    std::vector<RtPrim> prims; std::vector<RtMaterial> mats; std::vector<RtTexture> texs; std::vector<RtPerlin> perl;
    std::vector<RtImage> imgs; std::vector<RtXform> xf; std::vector<RtInstance> world, lights;
    std::vector<double> box[6]; std::vector<int32_t> left, right;
    explicit Copy(const RtSceneDesc& s) : d(s) {
        prims.assign(s.prims, s.prims + s.n_prims); mats.assign(s.materials, s.materials + s.n_materials);
        texs.assign(s.textures, s.textures + s.n_textures); perl.assign(s.perlins, s.perlins + s.n_perlins);
        imgs.assign(s.images, s.images + s.n_images); xf.assign(s.xforms, s.xforms + s.n_xforms);
        world.assign(s.world, s.world + s.n_world); lights.assign(s.lights, s.lights + s.n_lights);
        const double* src[6] = {s.bvh.min_x, s.bvh.min_y, s.bvh.min_z, s.bvh.max_x, s.bvh.max_y, s.bvh.max_z};
        for (int k = 0; k < 6; k++) box[k].assign(src[k], src[k] + s.bvh.n_nodes);
        left.assign(s.bvh.left, s.bvh.left + s.bvh.n_nodes); right.assign(s.bvh.right, s.bvh.right + s.bvh.n_nodes);
        relink();
    }
    void relink() {
        d.prims = prims.data(); d.materials = mats.data(); d.textures = texs.data(); d.perlins = perl.data(); d.images = imgs.data();
        d.xforms = xf.data(); d.world = world.data(); d.lights = lights.data();
        d.bvh.min_x = box[0].data(); d.bvh.min_y = box[1].data(); d.bvh.min_z = box[2].data();
        d.bvh.max_x = box[3].data(); d.bvh.max_y = box[4].data(); d.bvh.max_z = box[5].data();
        d.bvh.left = left.data(); d.bvh.right = right.data();
    }
};

std::mt19937_64 rng;
uint32_t R(uint32_t n) { return n ? (uint32_t)(rng() % n) : 0u; }
double bad_float() {
    static const double v[] = {NAN, INFINITY, -INFINITY, 0.0, -0.0, -1.0, 1e308, -1e308, 5e-324, 1e-300, 1e300, 3.5};
    return v[R(sizeof v / sizeof *v)];
}
uint32_t weird(uint32_t n) {
    const uint32_t v[] = {n, n + 1, 2 * n + 7, 0xFFFFFFFFu, 0x7FFFFFFFu, 0x80000000u, R(n ? n : 1), 1000000u};
    return v[R(8)];
}
int32_t weird_child(uint32_t n_nodes, uint32_t n_prims) {
    const int32_t v[] = {0, (int32_t)R(n_nodes ? n_nodes : 1), (int32_t)n_nodes, (int32_t)n_nodes + 5, -(int32_t)(n_prims + 1), -(int32_t)(n_prims + 9),
                         INT32_MIN, INT32_MAX, ~(int32_t)R(n_prims ? n_prims : 1)};
    return v[R(9)];
}

void corrupt(Copy& c) {
    const int n = 1 + (int)R(3);
    for (int it = 0; it < n; it++) {
        switch (R(16)) {
            case 0: if (!c.prims.empty()) c.prims[R(c.prims.size())].kind = R(2) ? R(8) : 0xFFFFFFFFu; break;
            case 1: if (!c.prims.empty()) c.prims[R(c.prims.size())].material = weird((uint32_t)c.mats.size()); break;
            case 2: if (!c.prims.empty()) c.prims[R(c.prims.size())].d[R(10)] = bad_float(); break;
            case 3: if (!c.mats.empty()) { RtMaterial& m = c.mats[R(c.mats.size())]; m.kind = R(2) ? R(8) : 0xFFFFFFFFu; m.texture = weird((uint32_t)c.texs.size()); m.fuzz = bad_float(); } break;
            case 4: if (!c.texs.empty()) { RtTexture& t = c.texs[R(c.texs.size())]; t.kind = R(7); t.odd = weird((uint32_t)c.texs.size()); t.even = weird((uint32_t)c.texs.size());
                                           t.table = weird((uint32_t)(c.perl.size() + c.imgs.size())); t.scale = bad_float(); } break;
            case 5: case 6: { std::vector<RtInstance>& l = R(2) ? c.world : c.lights; if (l.empty()) break; RtInstance& in = l[R(l.size())];
                switch (R(7)) { case 0: in.geom_kind = R(4); break; case 1: in.geom_index = weird((uint32_t)(c.prims.size() + c.left.size())); break;
                                case 2: in.xform_begin = weird((uint32_t)c.xf.size()); break; case 3: in.xform_count = weird((uint32_t)c.xf.size()); break;
                                case 4: in.is_medium = R(3); break; case 5: in.phase_material = weird((uint32_t)c.mats.size()); break; default: in.density = bad_float(); } } break;
            case 7: case 8: case 9: if (!c.left.empty()) { (R(2) ? c.left : c.right)[R(c.left.size())] = weird_child((uint32_t)c.left.size(), (uint32_t)c.prims.size()); } break;
            case 10: if (!c.left.empty()) c.box[R(6)][R(c.left.size())] = bad_float(); break;
            case 11: if (!c.xf.empty()) { RtXform& x = c.xf[R(c.xf.size())]; x.kind = R(4); x.v[R(3)] = bad_float(); } break;
            case 12: { double* cam = c.d.camera.origin; cam[R(24)] = bad_float(); } break;
            case 13: if (!c.perl.empty()) c.perl[R(c.perl.size())].perm_x[R(256)] = (int32_t)weird(256); break;
            case 14: {  // a count larger than its array would be the caller's lie (undetectable); a SMALLER one must be coped with
                switch (R(5)) { case 0: c.d.n_prims = R(c.d.n_prims + 1); break; case 1: c.d.n_materials = R(c.d.n_materials + 1); break;
                                case 2: c.d.n_textures = R(c.d.n_textures + 1); break; case 3: c.d.n_xforms = R(c.d.n_xforms + 1); break;
                                default: c.d.bvh.n_nodes = R(c.d.bvh.n_nodes + 1); } } break;
            default: if (!c.imgs.empty()) { RtImage& im = c.imgs[R(c.imgs.size())]; if (R(2)) im.width = R(2) ? 0 : im.width / 2; else im.height = R(2) ? 0 : im.height / 2; } break;
        }
    }
}
}  // namespace

int main(int argc, char** argv) {
    const int cases = argc > 1 ? std::atoi(argv[1]) : 500;
    rng.seed(argc > 2 ? std::strtoull(argv[2], nullptr, 0) : 1);
    const char* image = argc > 3 ? argv[3] : nullptr;
    std::vector<RthScene*> scenes;
    for (const char* const* n = rth_scene_names(); *n; n++) {
        RthScene* s = rth_scene_build(*n, 1.0, 7, image);
        if (s) scenes.push_back(s);  // (the two image scenes need the earth map)
    }
    if (scenes.empty()) { std::fprintf(stderr, "no scene\n"); return 2; }
    int rejected = 0, accepted = 0;
    for (int i = 0; i < cases; i++) {
        Copy c(*rth_scene_desc(scenes[R(scenes.size())]));
        corrupt(c);
        int rc = 0;
        void* h = sim_scene_create(&c.d, &rc);
        if (!h) {
            if (rc == 0 || !sim_last_error() || !*sim_last_error()) { std::fprintf(stderr, "case %d: refused without code / message\n", i); return 1; }
            rejected++;
            continue;
        }
        accepted++;
        const uint32_t W = 5, H = 4;
        int32_t ids[W * H]; double ts[W * H], acc[W * H * 3] = {0};
        for (int generic = 0; generic < 2; generic++) sim_trace_primary(h, W, H, 0.3, 0.5, RT_F64, generic, ids, ts);
        for (uint32_t precision : {(uint32_t)RT_F64, (uint32_t)RT_F32}) {
            RtRenderParams p;
            std::memset(&p, 0, sizeof p);
            p.width = W; p.height = H; p.spp_count = 1; p.max_depth = 5; p.max_nan_retries = 2; p.seed = 99; p.precision = precision;
            uint64_t casts = 0;
            sim_render(h, &p, 0, acc, &casts);
        }
        sim_scene_destroy(h);
    }
    for (RthScene* s : scenes) rth_scene_free(s);
    std::printf("%d corrupted descriptions: %d refused with a message, %d accepted and rendered; no sanitizer report\n", cases, rejected, accepted);
    return 0;
}
