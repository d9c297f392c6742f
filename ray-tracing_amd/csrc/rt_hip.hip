// rt_hip.hip — C-ABI implementation (include/rt_hip.h): scene validation + upload, launches.
// No CPU fallback exists: every entry point either runs the HIP kernels or returns an error.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_launch.hpp"

namespace {

thread_local std::string g_err;
thread_local std::string g_desc;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct DeviceBlock {
    void* ptr = nullptr;
};

template <class T> struct HostScene {
    std::vector<rt::DPrim<T>> prims;
    std::vector<rt::DMaterial<T>> materials;
    std::vector<rt::DTexture<T>> textures;
    std::vector<rt::DPerlin<T>> perlins;
    std::vector<rt::DXform<T>> xforms;
    std::vector<rt::DInstance<T>> world, lights;
    std::vector<rt::DBvhEntry<T>> bvh;
};

}  // namespace

struct RtScene {
    int device = 0;
    rt::DScene<float> s32;
    rt::DScene<double> s64;
    std::vector<void*> allocs;
    uint32_t n_prims = 0;
};

namespace {

template <class T> int upload(RtScene* sc, const std::vector<T>& v, const T** out) {
    *out = nullptr;
    if (v.empty()) return RT_OK;
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, v.size() * sizeof(T)));
    sc->allocs.push_back(d);
    HIP_TRY(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)d;
    return RT_OK;
}

bool texture_needs_uv(const RtSceneDesc* d, uint32_t tex, int depth) {
    if (depth > 16) return false;
    const RtTexture& t = d->textures[tex];
    switch (t.kind) {
        case RT_TEX_CHECKER_UV: case RT_TEX_IMAGE: return true;
        case RT_TEX_CHECKER_3D: return texture_needs_uv(d, t.odd, depth + 1) || texture_needs_uv(d, t.even, depth + 1);
        default: return false;
    }
}
int texture_depth(const RtSceneDesc* d, uint32_t tex, int depth) {
    if (depth > 16) return depth;
    const RtTexture& t = d->textures[tex];
    if (t.kind == RT_TEX_CHECKER_3D || t.kind == RT_TEX_CHECKER_UV) {
        if (t.odd >= d->n_textures || t.even >= d->n_textures) return 99;
        int a = texture_depth(d, t.odd, depth + 1), b = texture_depth(d, t.even, depth + 1);
        return a > b ? a : b;
    }
    return depth;
}

int validate_instances(const RtSceneDesc* d, const RtInstance* ins, uint32_t n, const char* what) {
    for (uint32_t i = 0; i < n; i++) {
        const RtInstance& in = ins[i];
        std::string at = std::string(what) + "[" + std::to_string(i) + "]";
        if (in.geom_kind == RT_GEOM_PRIM) {
            if (in.geom_index >= d->n_prims) return fail(RT_ERR_INVALID, at + ": primitive index out of range");
        } else if (in.geom_kind == RT_GEOM_BVH) {
            if (in.geom_index >= d->bvh.n_nodes) return fail(RT_ERR_INVALID, at + ": BVH root out of range");
        } else {
            return fail(RT_ERR_INVALID, at + ": unknown geom_kind");
        }
        if ((uint64_t)in.xform_begin + in.xform_count > d->n_xforms) return fail(RT_ERR_INVALID, at + ": xform range out of range");
        if (in.is_medium) {
            if (in.phase_material >= d->n_materials) return fail(RT_ERR_INVALID, at + ": phase material out of range");
            if (!(in.density > 0.0)) return fail(RT_ERR_INVALID, at + ": medium density must be > 0");
        }
    }
    return RT_OK;
}

int validate(const RtSceneDesc* d) {
    if (!d) return fail(RT_ERR_INVALID, "scene description is NULL");
    if (d->abi_version != RT_ABI_VERSION) return fail(RT_ERR_INVALID, "abi_version mismatch");
    if (d->n_world == 0) return fail(RT_ERR_INVALID, "world is empty");
    if ((d->n_prims && !d->prims) || (d->n_materials && !d->materials) || (d->n_textures && !d->textures) ||
        (d->n_perlins && !d->perlins) || (d->n_images && !d->images) || (d->n_xforms && !d->xforms) || !d->world ||
        (d->n_lights && !d->lights))
        return fail(RT_ERR_INVALID, "NULL array with non-zero count");
    for (uint32_t i = 0; i < d->n_textures; i++) {
        const RtTexture& t = d->textures[i];
        if (t.kind > RT_TEX_IMAGE) return fail(RT_ERR_INVALID, "texture kind out of range");
        if (t.kind == RT_TEX_NOISE && t.table >= d->n_perlins) return fail(RT_ERR_INVALID, "perlin index out of range");
        if (t.kind == RT_TEX_IMAGE && t.table >= d->n_images) return fail(RT_ERR_INVALID, "image index out of range");
        if (texture_depth(d, i, 0) > 16) return fail(RT_ERR_UNSUPPORTED, "checker textures nested deeper than 16 (or cyclic / out of range)");
    }
    for (uint32_t i = 0; i < d->n_images; i++)
        if (!d->images[i].rgb || d->images[i].width == 0 || d->images[i].height == 0) return fail(RT_ERR_INVALID, "empty image");
    for (uint32_t i = 0; i < d->n_perlins; i++)
        for (int k = 0; k < 256; k++)
            if ((uint32_t)d->perlins[i].perm_x[k] > 255u || (uint32_t)d->perlins[i].perm_y[k] > 255u || (uint32_t)d->perlins[i].perm_z[k] > 255u)
                return fail(RT_ERR_INVALID, "perlin permutation entry outside 0..255");
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const RtMaterial& m = d->materials[i];
        if (m.kind > RT_MAT_FROSTED_DIELECTRIC) return fail(RT_ERR_INVALID, "material kind out of range");
        bool textured = m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_DIFFUSE_LIGHT || m.kind == RT_MAT_ISOTROPIC;
        if (textured && m.texture >= d->n_textures) return fail(RT_ERR_INVALID, "material texture index out of range");
    }
    for (uint32_t i = 0; i < d->n_prims; i++) {
        if (d->prims[i].kind > RT_PRIM_CUBOID) return fail(RT_ERR_INVALID, "primitive kind out of range");
        if (d->prims[i].material >= d->n_materials) return fail(RT_ERR_INVALID, "primitive material index out of range");
    }
    for (uint32_t i = 0; i < d->n_xforms; i++)
        if (d->xforms[i].kind > RT_XFORM_ROTATE_Y) return fail(RT_ERR_INVALID, "xform kind out of range");
    const RtBvh& b = d->bvh;
    if (b.n_nodes) {
        if (!b.min_x || !b.min_y || !b.min_z || !b.max_x || !b.max_y || !b.max_z || !b.left || !b.right)
            return fail(RT_ERR_INVALID, "NULL BVH array");
        for (uint32_t i = 0; i < b.n_nodes; i++) {
            const int32_t kids[2] = {b.left[i], b.right[i]};
            for (int k = 0; k < 2; k++) {
                if (kids[k] >= 0) {  // pre-order: a child node always follows its parent -> acyclic
                    if ((uint32_t)kids[k] <= i || (uint32_t)kids[k] >= b.n_nodes) return fail(RT_ERR_INVALID, "BVH child node index not in pre-order range");
                } else if ((uint32_t)(~kids[k]) >= d->n_prims) {
                    return fail(RT_ERR_INVALID, "BVH leaf primitive index out of range");
                }
            }
        }
    }
    int rc = validate_instances(d, d->world, d->n_world, "world");
    if (rc) return rc;
    return validate_instances(d, d->lights, d->n_lights, "lights");
}

// SoA nodes -> threaded pre-order entries (internal + leaf), see rt_device.hpp DBvhEntry.
template <class T>
void thread_bvh(const RtBvh& b, int32_t node, std::vector<rt::DBvhEntry<T>>& out, std::vector<int32_t>& root_entry) {
    // explicit stack: (node, state) — state 0 = emit node, 1 = after left, 2 = after right
    struct Frame { int32_t node; int state; size_t entry; };
    std::vector<Frame> st;
    st.push_back({node, 0, 0});
    root_entry[node] = (int32_t)out.size();
    auto emit_leaf = [&](int32_t child) {
        rt::DBvhEntry<T> e;
        for (int k = 0; k < 3; k++) { e.mn[k] = T(0); e.mx[k] = T(0); }
        e.prim = ~child;
        e.skip = (int32_t)out.size() + 1;
        out.push_back(e);
    };
    while (!st.empty()) {
        Frame& f = st.back();
        if (f.state == 0) {
            rt::DBvhEntry<T> e;
            e.mn[0] = (T)b.min_x[f.node]; e.mn[1] = (T)b.min_y[f.node]; e.mn[2] = (T)b.min_z[f.node];
            e.mx[0] = (T)b.max_x[f.node]; e.mx[1] = (T)b.max_y[f.node]; e.mx[2] = (T)b.max_z[f.node];
            e.prim = -1; e.skip = 0;
            f.entry = out.size();
            out.push_back(e);
            f.state = 1;
            int32_t l = b.left[f.node];
            if (l >= 0) { st.push_back({l, 0, 0}); } else emit_leaf(l);
        } else if (f.state == 1) {
            f.state = 2;
            int32_t r = b.right[f.node];
            if (r >= 0) { st.push_back({r, 0, 0}); } else emit_leaf(r);
        } else {
            out[f.entry].skip = (int32_t)out.size();
            st.pop_back();
        }
    }
}

template <class T> void convert_instances(const RtInstance* ins, uint32_t n, std::vector<rt::DInstance<T>>& out) {
    for (uint32_t i = 0; i < n; i++) {
        rt::DInstance<T> o;
        o.density = (T)ins[i].density;
        o.geom_kind = ins[i].geom_kind; o.geom_index = ins[i].geom_index;
        o.xform_begin = ins[i].xform_begin; o.xform_count = ins[i].xform_count;
        o.is_medium = ins[i].is_medium ? 1u : 0u; o.phase_material = ins[i].phase_material;
        out.push_back(o);
    }
}

template <class T>
int build_device_scene(RtScene* sc, const RtSceneDesc* d, const rt::DImage* d_images, rt::DScene<T>& out) {
    HostScene<T> h;
    for (uint32_t i = 0; i < d->n_prims; i++) {
        rt::DPrim<T> p;
        for (int k = 0; k < 10; k++) p.d[k] = (T)d->prims[i].d[k];
        p.kind = d->prims[i].kind; p.material = d->prims[i].material;
        h.prims.push_back(p);
    }
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const RtMaterial& m = d->materials[i];
        rt::DMaterial<T> o;
        for (int k = 0; k < 3; k++) o.albedo[k] = (T)m.albedo[k];
        o.fuzz = (T)m.fuzz; o.ref_idx = (T)m.ref_idx; o.texture = m.texture;
        bool textured = m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_DIFFUSE_LIGHT || m.kind == RT_MAT_ISOTROPIC;
        o.kind = m.kind | ((textured && texture_needs_uv(d, m.texture, 0)) ? 0x100u : 0u);  // bit 8: shading needs (u,v)
        h.materials.push_back(o);
    }
    for (uint32_t i = 0; i < d->n_textures; i++) {
        const RtTexture& t = d->textures[i];
        rt::DTexture<T> o;
        for (int k = 0; k < 3; k++) o.color[k] = (T)t.color[k];
        o.scale = (T)t.scale; o.kind = t.kind; o.odd = t.odd; o.even = t.even; o.table = t.table;
        h.textures.push_back(o);
    }
    for (uint32_t i = 0; i < d->n_perlins; i++) {
        const RtPerlin& p = d->perlins[i];
        rt::DPerlin<T> o;
        for (int k = 0; k < 256; k++) {
            o.perm_x[k] = (uint8_t)p.perm_x[k]; o.perm_y[k] = (uint8_t)p.perm_y[k]; o.perm_z[k] = (uint8_t)p.perm_z[k];
            for (int c = 0; c < 3; c++) o.ranvec[k][c] = (T)p.ranvec[k][c];
        }
        h.perlins.push_back(o);
    }
    for (uint32_t i = 0; i < d->n_xforms; i++) {
        rt::DXform<T> o;
        for (int k = 0; k < 3; k++) o.v[k] = (T)d->xforms[i].v[k];
        o.kind = d->xforms[i].kind;
        h.xforms.push_back(o);
    }
    convert_instances<T>(d->world, d->n_world, h.world);
    convert_instances<T>(d->lights, d->n_lights, h.lights);
    std::vector<int32_t> root_entry(d->bvh.n_nodes, -1);
    auto add_roots = [&](const RtInstance* ins, uint32_t n) {
        for (uint32_t i = 0; i < n; i++)
            if (ins[i].geom_kind == RT_GEOM_BVH && root_entry[ins[i].geom_index] < 0) thread_bvh<T>(d->bvh, (int32_t)ins[i].geom_index, h.bvh, root_entry);
    };
    add_roots(d->world, d->n_world);
    add_roots(d->lights, d->n_lights);

    int rc;
    if ((rc = upload(sc, h.prims, &out.prims))) return rc;
    if ((rc = upload(sc, h.materials, &out.materials))) return rc;
    if ((rc = upload(sc, h.textures, &out.textures))) return rc;
    if ((rc = upload(sc, h.perlins, &out.perlins))) return rc;
    if ((rc = upload(sc, h.xforms, &out.xforms))) return rc;
    if ((rc = upload(sc, h.world, &out.world))) return rc;
    if ((rc = upload(sc, h.lights, &out.lights))) return rc;
    if ((rc = upload(sc, h.bvh, &out.bvh))) return rc;
    if ((rc = upload(sc, root_entry, &out.bvh_root_entry))) return rc;
    out.images = d_images;
    out.n_world = d->n_world;
    out.n_lights = d->n_lights;
    for (int k = 0; k < 3; k++) out.background[k] = (T)d->background[k];
    const RtCamera& c = d->camera;
    for (int k = 0; k < 3; k++) {
        out.cam.origin[k] = (T)c.origin[k]; out.cam.llc[k] = (T)c.lower_left_corner[k];
        out.cam.horizontal[k] = (T)c.horizontal[k]; out.cam.vertical[k] = (T)c.vertical[k];
        out.cam.u[k] = (T)c.u[k]; out.cam.v[k] = (T)c.v[k]; out.cam.w[k] = (T)c.w[k];
    }
    out.cam.lens_radius = (T)c.lens_radius; out.cam.t1 = (T)c.t1; out.cam.t2 = (T)c.t2;
    return RT_OK;
}

int check_scene(RtScene* scene) {
    if (!scene) return fail(RT_ERR_INVALID, "scene handle is NULL");
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != scene->device) return fail(RT_ERR_INVALID, "scene belongs to device " + std::to_string(scene->device) + " but device " + std::to_string(dev) + " is current");
    return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return g_err.c_str(); }

int rt_init(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return fail(RT_ERR_NO_DEVICE, std::string("no HIP device: ") + hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(RT_ERR_INVALID, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RT_ERR_NO_DEVICE, std::string("this library contains gfx950 code only; device is ") + prop.gcnArchName);
    g_desc = "rt_hip abi=" + std::to_string(RT_ABI_VERSION) + " " + prop.gcnArchName + " cu=" + std::to_string(prop.multiProcessorCount) +
             " clock_mhz=" + std::to_string(prop.clockRate / 1000);
    return RT_OK;
}

const char* rt_describe(void) { return g_desc.c_str(); }

int rt_scene_create(const RtSceneDesc* desc, RtScene** out) {
    if (!out) return fail(RT_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int rc = validate(desc);
    if (rc) return rc;
    RtScene* sc = new RtScene();
    if (hipGetDevice(&sc->device) != hipSuccess) { delete sc; return fail(RT_ERR_NO_DEVICE, "no current HIP device (call rt_init)"); }
    sc->n_prims = desc->n_prims;
    // images: texel rasters + descriptor table
    std::vector<rt::DImage> imgs;
    for (uint32_t i = 0; i < desc->n_images && rc == RT_OK; i++) {
        const RtImage& im = desc->images[i];
        size_t bytes = (size_t)im.width * im.height * 3;
        void* d = nullptr;
        if (hipMalloc(&d, bytes) != hipSuccess) { rc = fail(RT_ERR_HIP, "hipMalloc(image) failed"); break; }
        sc->allocs.push_back(d);
        if (hipMemcpy(d, im.rgb, bytes, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(RT_ERR_HIP, "hipMemcpy(image) failed"); break; }
        imgs.push_back(rt::DImage{(const uint8_t*)d, im.width, im.height});
    }
    const rt::DImage* d_images = nullptr;
    if (rc == RT_OK) rc = upload(sc, imgs, &d_images);
    if (rc == RT_OK) rc = build_device_scene<float>(sc, desc, d_images, sc->s32);
    if (rc == RT_OK) rc = build_device_scene<double>(sc, desc, d_images, sc->s64);
    if (rc != RT_OK) { rt_scene_destroy(sc); return rc; }
    *out = sc;
    return RT_OK;
}

int rt_scene_destroy(RtScene* scene) {
    if (!scene) return RT_OK;
    for (void* p : scene->allocs) (void)hipFree(p);
    delete scene;
    return RT_OK;
}

int rt_trace_primary(RtScene* scene, uint32_t width, uint32_t height, double time, double medium_xi, uint32_t precision,
                     int32_t* d_ids, double* d_t, void* stream) {
    int rc = check_scene(scene);
    if (rc) return rc;
    if (width < 2 || height < 2 || (uint64_t)width * height > 0x7fffffffull) return fail(RT_ERR_INVALID, "width/height must be >= 2 and width*height < 2^31");
    if (!d_ids || !d_t) return fail(RT_ERR_INVALID, "output pointer is NULL");
    if (!(medium_xi > 0.0 && medium_xi < 1.0)) return fail(RT_ERR_INVALID, "medium_xi must lie in (0,1)");
    if (precision == RT_F64) HIP_TRY(rt::launch_trace_primary<double>(scene->s64, width, height, time, medium_xi, d_ids, d_t, (hipStream_t)stream));
    else if (precision == RT_F32) HIP_TRY(rt::launch_trace_primary<float>(scene->s32, width, height, time, medium_xi, d_ids, d_t, (hipStream_t)stream));
    else return fail(RT_ERR_INVALID, "unknown precision");
    return RT_OK;
}

static int make_args(const RtRenderParams* p, rt::RenderArgs& a) {
    if (!p) return fail(RT_ERR_INVALID, "params is NULL");
    if (p->width < 2 || p->height < 2 || (uint64_t)p->width * p->height > 0x7fffffffull) return fail(RT_ERR_INVALID, "width/height must be >= 2 and width*height < 2^31");
    if (p->max_depth < 0) return fail(RT_ERR_INVALID, "max_depth must be >= 0");
    if ((uint64_t)p->spp_begin + p->spp_count > 0xffffffffull) return fail(RT_ERR_INVALID, "spp range overflows 32 bits");
    a.width = p->width; a.height = p->height; a.spp_begin = p->spp_begin; a.spp_count = p->spp_count;
    a.max_depth = p->max_depth; a.max_nan_retries = p->max_nan_retries ? p->max_nan_retries : 64u; a.seed = p->seed;
    return RT_OK;
}

int rt_render(RtScene* scene, const RtRenderParams* params, float* d_accum, uint64_t* d_stats, void* stream) {
    int rc = check_scene(scene);
    if (rc) return rc;
    rt::RenderArgs a;
    if ((rc = make_args(params, a))) return rc;
    if (!d_accum) return fail(RT_ERR_INVALID, "d_accum is NULL");
    if (a.spp_count == 0) return RT_OK;
    if (params->precision == RT_F64) HIP_TRY(rt::launch_render<double>(scene->s64, a, d_accum, (unsigned long long*)d_stats, (hipStream_t)stream));
    else if (params->precision == RT_F32) HIP_TRY(rt::launch_render<float>(scene->s32, a, d_accum, (unsigned long long*)d_stats, (hipStream_t)stream));
    else return fail(RT_ERR_INVALID, "unknown precision");
    return RT_OK;
}

/* Test hook (not part of the drop-in surface): radiance of samples [spp_begin, spp_begin+spp_count)
 * of ONE pixel, one value per sample, for sample-for-sample comparison with the oracle. */
int rt_debug_sample_pixel(RtScene* scene, const RtRenderParams* params, uint32_t x, uint32_t y, double* d_out, void* stream) {
    int rc = check_scene(scene);
    if (rc) return rc;
    rt::RenderArgs a;
    if ((rc = make_args(params, a))) return rc;
    if (x >= a.width || y >= a.height || !d_out) return fail(RT_ERR_INVALID, "bad pixel / output");
    if (a.spp_count == 0) return RT_OK;
    if (params->precision == RT_F64) HIP_TRY(rt::launch_sample_pixel<double>(scene->s64, a, x, y, a.spp_count, d_out, (hipStream_t)stream));
    else if (params->precision == RT_F32) HIP_TRY(rt::launch_sample_pixel<float>(scene->s32, a, x, y, a.spp_count, d_out, (hipStream_t)stream));
    else return fail(RT_ERR_INVALID, "unknown precision");
    return RT_OK;
}

int rt_resolve(const float* d_accum, uint32_t width, uint32_t height, uint32_t spp, uint8_t* d_rgb, void* stream) {
    if (!d_accum || !d_rgb) return fail(RT_ERR_INVALID, "NULL buffer");
    if (width == 0 || height == 0 || spp == 0 || (uint64_t)width * height * 3 > 0xffffffffull) return fail(RT_ERR_INVALID, "bad width/height/spp");
    HIP_TRY(rt::launch_resolve(d_accum, width, height, spp, d_rgb, (hipStream_t)stream));
    return RT_OK;
}

int rt_device_alloc(void** d_ptr, uint64_t bytes) {
    if (!d_ptr) return fail(RT_ERR_INVALID, "d_ptr is NULL");
    HIP_TRY(hipMalloc(d_ptr, bytes));
    return RT_OK;
}
int rt_device_free(void* d_ptr) {
    HIP_TRY(hipFree(d_ptr));
    return RT_OK;
}
int rt_device_zero(void* d_ptr, uint64_t bytes, void* stream) {
    HIP_TRY(hipMemsetAsync(d_ptr, 0, bytes, (hipStream_t)stream));
    return RT_OK;
}
int rt_copy_to_host(void* h_dst, const void* d_src, uint64_t bytes, void* stream) {
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return RT_OK;
}
int rt_stream_synchronize(void* stream) {
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return RT_OK;
}

}  // extern "C"
