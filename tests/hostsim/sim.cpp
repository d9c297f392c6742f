// TEST INFRASTRUCTURE ONLY — host simulation of the DEVICE code.  The device headers are compiled
// for the CPU (RT_HOST_SIM, -ffp-contract=off) and driven one pixel / one sample at a time, with
// the scene packed by the same rt_pack.hpp that rt_hip.hip uploads from.  tests/test_hostsim.py
// compares the results with the oracle and the golden fixtures: T = double must agree bit for bit
// (same arithmetic, same RNG, same libm).  What this does NOT cover is the HIP kernels' own
// lane/wave logic (in-place regeneration, accumulation, stats): that is what the -m gpu tests do.
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "rt_pack.hpp"
#include "rt_path.hpp"
#include "rt_variants.hpp"

namespace {
template <class T> struct Side {
    rt::PackedScene<T> p;
    std::vector<rt::DBvhLite> lds_copy;
    void seal(const rt::DImage* images) {
        rt::DScene<T>& v = p.view;
        v.prims = p.prims.data(); v.materials = p.materials.data(); v.textures = p.textures.data(); v.perlins = p.perlins.data();
        v.xforms = p.xforms.data(); v.world = p.world.data(); v.lights = p.lights.data(); v.bvh = p.bvh.data();
        v.bvh_root_entry = p.bvh_root_entry.data(); v.images = images;
        v.bvh_lite = p.bvh_lite.data();
        v.fast_lite = p.fast_lite.data(); v.fast_root_entry = p.fast_root_entry.data(); v.world_order = p.world_order.data();
        // exercise both sources of node records: the first half "staged" (a second copy), the rest "global"
        lds_copy.assign(p.fast_lite.begin(), p.fast_lite.begin() + p.fast_lite.size() / 2);
        v.bvh_lds = lds_copy.data(); v.n_bvh_lds = (uint32_t)lds_copy.size();
    }
};
struct SimScene {
    uint32_t features = 0;
    std::vector<std::vector<uint8_t>> texels;
    std::vector<rt::DImage> images;
    Side<double> s64;
    Side<float> s32;
};
rt::RenderArgs args_of(const RtRenderParams* p) {
    rt::RenderArgs a;
    a.width = p->width; a.height = p->height; a.spp_begin = p->spp_begin; a.spp_count = p->spp_count;
    a.max_depth = p->max_depth; a.max_nan_retries = p->max_nan_retries ? p->max_nan_retries : 64u; a.seed = p->seed;
    return a;
}
template <class T, uint32_t F>
void primary(const rt::DScene<T>& sc, uint32_t W, uint32_t H, double time, double xi, int32_t* ids, double* ts) {
    for (uint32_t y = 0; y < H; y++)
        for (uint32_t x = 0; x < W; x++) rt::probe_primary<T, F>(sc, W, H, x, y, (T)time, xi, ids[y * W + x], ts[y * W + x]);
}
template <class T, uint32_t F>
void samples(const rt::DScene<T>& sc, const rt::RenderArgs& a, uint32_t x, uint32_t y, double* out, uint64_t* casts) {
    for (uint32_t i = 0; i < a.spp_count; i++) {
        uint32_t n = 0;
        rt::V3<T> r = rt::radiance_of_sample<T, F>(sc, a, x, y, a.spp_begin + i, n);
        out[3 * i] = (double)r.x; out[3 * i + 1] = (double)r.y; out[3 * i + 2] = (double)r.z;
        if (casts) *casts += n;
    }
}
template <class T, uint32_t F>
void frame(const rt::DScene<T>& sc, const rt::RenderArgs& a, double* accum, uint64_t* casts) {
    for (uint32_t y = 0; y < a.height; y++)
        for (uint32_t x = 0; x < a.width; x++) {
            double sx = 0, sy = 0, sz = 0;
            for (uint32_t s = a.spp_begin; s < a.spp_begin + a.spp_count; s++) {
                uint32_t n = 0;
                rt::V3<T> r = rt::radiance_of_sample<T, F>(sc, a, x, y, s, n);
                sx += (double)r.x; sy += (double)r.y; sz += (double)r.z;
                if (casts) *casts += n;
            }
            double* px = accum + ((size_t)y * a.width + x) * 3;
            px[0] += sx; px[1] += sy; px[2] += sz;
        }
}
}  // namespace

#ifdef RT_SIM_COUNTERS
extern "C" { unsigned long long g_sim_counters[8] = {0}; unsigned long long g_sim_walk_hist[64] = {0}; }
#endif
extern "C" {
const char* sim_last_error() { return rt::pack_error().c_str(); }
uint32_t sim_scene_features(void* s) { return ((SimScene*)s)->features; }
// the packer's ray-time intervals (f64 side): shutter hull, fast-walk interval, node-cull interval
void sim_scene_intervals(void* s, double out[6]) {
    const rt::DScene<double>& v = ((SimScene*)s)->s64.p.view;
    out[0] = v.time_lo; out[1] = v.time_hi; out[2] = v.fast_lo; out[3] = v.fast_hi; out[4] = v.node_cull_lo; out[5] = v.node_cull_hi;
}
// force_generic != 0: always run the F_ALL instantiation (checks that every variant agrees with it)
void* sim_scene_create(const RtSceneDesc* d, int* rc_out) {
    int rc = rt::validate(d);
    if (rc_out) *rc_out = rc;
    if (rc) return nullptr;
    SimScene* s = new SimScene();
    s->features = rt::scene_features(d);
    for (uint32_t i = 0; i < d->n_images; i++) {
        const RtImage& im = d->images[i];
        s->texels.emplace_back(im.rgb, im.rgb + (size_t)im.width * im.height * 3);
    }
    for (uint32_t i = 0; i < d->n_images; i++) s->images.push_back(rt::DImage{s->texels[i].data(), d->images[i].width, d->images[i].height});
    rt::pack_scene<double>(d, s->s64.p);
    rt::pack_scene<float>(d, s->s32.p);
    s->s64.seal(s->images.data());
    s->s32.seal(s->images.data());
    return s;
}
void sim_scene_destroy(void* s) { delete (SimScene*)s; }
// rt_device.hpp: checker_is_odd (the sign-only decision) beside the product of the three sines it stands for, and whether it was sure
void sim_checker(const double* xyz, uint32_t n, uint8_t* fast, uint8_t* plain, uint8_t* sure_out) {
    for (uint32_t i = 0; i < n; i++) {
        const double ax = xyz[3 * i], ay = xyz[3 * i + 1], az = xyz[3 * i + 2];
        fast[i] = rt::checker_is_odd_body(ax, ay, az) ? 1 : 0;
        plain[i] = (std::sin(ax) * std::sin(ay) * std::sin(az) < 0.0) ? 1 : 0;
        bool sure = true;
        (void)(rt::sin_sign(ax, sure) * rt::sin_sign(ay, sure) * rt::sin_sign(az, sure));
        sure_out[i] = sure ? 1 : 0;
    }
}

void sim_trace_primary(void* h, uint32_t W, uint32_t H, double time, double xi, uint32_t precision, int force_generic, int32_t* ids, double* ts) {
    SimScene* s = (SimScene*)h;
    uint32_t f = force_generic ? rt::F_ALL : s->features;
    if (precision == RT_F64) {
#define CALL(V) primary<double, V>(s->s64.p.view, W, H, time, xi, ids, ts)
        RT_FOR_VARIANT(f, CALL);
#undef CALL
    } else {
#define CALL(V) primary<float, V>(s->s32.p.view, W, H, time, xi, ids, ts)
        RT_FOR_VARIANT(f, CALL);
#undef CALL
    }
}
void sim_sample_pixel(void* h, const RtRenderParams* p, uint32_t x, uint32_t y, int force_generic, double* out, uint64_t* casts) {
    SimScene* s = (SimScene*)h;
    rt::RenderArgs a = args_of(p);
    uint32_t f = force_generic ? rt::F_ALL : s->features;
    if (p->precision == RT_F64) {
#define CALL(V) samples<double, V>(s->s64.p.view, a, x, y, out, casts)
        RT_FOR_VARIANT(f, CALL);
#undef CALL
    } else {
#define CALL(V) samples<float, V>(s->s32.p.view, a, x, y, out, casts)
        RT_FOR_VARIANT(f, CALL);
#undef CALL
    }
}
void sim_render(void* h, const RtRenderParams* p, int force_generic, double* accum, uint64_t* casts) {
    SimScene* s = (SimScene*)h;
    rt::RenderArgs a = args_of(p);
    uint32_t f = force_generic ? rt::F_ALL : s->features;
    if (p->precision == RT_F64) {
#define CALL(V) frame<double, V>(s->s64.p.view, a, accum, casts)
        RT_FOR_VARIANT(f, CALL);
#undef CALL
    } else {
#define CALL(V) frame<float, V>(s->s32.p.view, a, accum, casts)
        RT_FOR_VARIANT(f, CALL);
#undef CALL
    }
}
}  // extern "C"
