// rt_hip.hip — C-ABI implementation (include/rt_hip.h): scene validation + upload, launches.
// No CPU fallback exists: every entry point either runs the HIP kernels or returns an error.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_launch.hpp"
#include "rt_pack.hpp"

#ifndef RT_BUILD_ID
#define RT_BUILD_ID "unknown"  // _build.py: hash of the device sources and flags this library was built from
#endif

namespace {

thread_local std::string g_desc;

using rt::fail;
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Static device scratch of rt_render_rgb8_noalloc: [0,256) job counter, then RT_NOALLOC_MAX_PIXELS x 3 float sums.
// In host code the array's NAME is only a stand-in: rt_init() resolves its device address once per device with
// hipGetSymbolAddress and everything below uses that pointer.
constexpr int kMaxDevices = 64;
constexpr size_t kScratchHeader = 256;
unsigned char* g_scratch_ptr[kMaxDevices] = {};

}  // namespace

static __device__ __align__(256) unsigned char scratch[kScratchHeader + (size_t)RT_NOALLOC_MAX_PIXELS * 3u * sizeof(float)];

struct RtScene {
    int device = 0;
    uint32_t features = 0;  // rt_device.hpp feature mask: selects the kernel instantiation
    rt::DScene<float> s32;
    rt::DScene<double> s64;
    std::vector<void*> allocs;
    uint32_t n_prims = 0;
    // job counters of the persistent render kernel: a ring, one word per launch, so that launches queued on
    // different streams do not share one.  A slot is reused after kCounters further launches: if its previous
    // launch went to ANOTHER stream, the new launch first waits (on the device, hipStreamWaitEvent) for the event
    // recorded behind that launch; on the same stream, stream order already separates the two.
    static constexpr uint32_t kCounters = 64;
    uint32_t* d_counters = nullptr;
    uint32_t next_counter = 0;
    hipEvent_t slot_event[kCounters] = {};
    hipStream_t slot_stream[kCounters] = {};
    bool slot_recorded[kCounters] = {};
    uint32_t cu_count = 0;
    // rt_render_rgb8: the float accumulation buffer between the render and the resolve kernel (grown on demand)
    float* d_work = nullptr;
    uint64_t work_floats = 0;
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

template <class T>
int build_device_scene(RtScene* sc, const RtSceneDesc* d, const rt::DImage* d_images, rt::DScene<T>& out) {
    rt::PackedScene<T> h;
    rt::pack_scene<T>(d, h);
    out = h.view;
    int rc;
    if ((rc = upload(sc, h.prims, &out.prims))) return rc;
    if ((rc = upload(sc, h.materials, &out.materials))) return rc;
    if ((rc = upload(sc, h.textures, &out.textures))) return rc;
    if ((rc = upload(sc, h.perlins, &out.perlins))) return rc;
    if ((rc = upload(sc, h.xforms, &out.xforms))) return rc;
    if ((rc = upload(sc, h.world, &out.world))) return rc;
    if ((rc = upload(sc, h.lights, &out.lights))) return rc;
    if ((rc = upload(sc, h.bvh, &out.bvh))) return rc;
    if ((rc = upload(sc, h.bvh_root_entry, &out.bvh_root_entry))) return rc;
    if ((rc = upload(sc, h.bvh_lite, &out.bvh_lite))) return rc;
    if ((rc = upload(sc, h.fast_lite, &out.fast_lite))) return rc;
    if ((rc = upload(sc, h.fast_root_entry, &out.fast_root_entry))) return rc;
    if ((rc = upload(sc, h.world_order, &out.world_order))) return rc;
    out.images = d_images;
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

const char* rt_last_error(void) { return rt::pack_error().c_str(); }

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
    if (device < kMaxDevices && !g_scratch_ptr[device]) {  // once per device, with that device current
        void* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        g_scratch_ptr[device] = (unsigned char*)p;
    }
    g_desc = "rt_hip abi=" + std::to_string(RT_ABI_VERSION) + " " + prop.gcnArchName + " cu=" + std::to_string(prop.multiProcessorCount) +
             " clock_mhz=" + std::to_string(prop.clockRate / 1000) + " build=" RT_BUILD_ID;
    return RT_OK;
}

const char* rt_describe(void) { return g_desc.c_str(); }

static int scene_create(const RtSceneDesc* desc, RtScene*& sc, RtScene** out) {
    int rc = rt::validate(desc);
    if (rc) return rc;
    sc = new RtScene();
    if (hipGetDevice(&sc->device) != hipSuccess) { delete sc; sc = nullptr; return fail(RT_ERR_NO_DEVICE, "no current HIP device (call rt_init)"); }
    sc->n_prims = desc->n_prims;
    sc->features = rt::scene_features(desc);
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
    if (rc == RT_OK) {
        hipDeviceProp_t prop;
        void* d = nullptr;
        if (hipGetDeviceProperties(&prop, sc->device) != hipSuccess || hipMalloc(&d, RtScene::kCounters * sizeof(uint32_t)) != hipSuccess) {
            rc = fail(RT_ERR_HIP, "hipMalloc(job counters) failed");
        } else {
            sc->allocs.push_back(d);
            sc->d_counters = (uint32_t*)d;
            sc->cu_count = (uint32_t)prop.multiProcessorCount;
        }
    }
    if (rc != RT_OK) { rt_scene_destroy(sc); sc = nullptr; return rc; }
    *out = sc;
    return RT_OK;
}

int rt_scene_create(const RtSceneDesc* desc, RtScene** out) {
    if (!out) return fail(RT_ERR_INVALID, "out is NULL");
    *out = nullptr;
    RtScene* sc = nullptr;
    try {  // nothing may unwind across the C boundary (host vectors of a huge description can throw bad_alloc)
        return scene_create(desc, sc, out);
    } catch (const std::bad_alloc&) {
        if (sc) rt_scene_destroy(sc);
        return fail(RT_ERR_INVALID, "out of host memory while packing the scene description");
    } catch (const std::exception& e) {
        if (sc) rt_scene_destroy(sc);
        return fail(RT_ERR_INVALID, std::string("scene description rejected: ") + e.what());
    }
}

int rt_scene_destroy(RtScene* scene) {
    if (!scene) return RT_OK;
    for (void* p : scene->allocs) (void)hipFree(p);
    if (scene->d_work) (void)hipFree(scene->d_work);
    for (hipEvent_t e : scene->slot_event)
        if (e) (void)hipEventDestroy(e);
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
    if (precision == RT_F64) HIP_TRY(rt::launch_trace_primary<double>(scene->s64, scene->features, width, height, time, medium_xi, d_ids, d_t, (hipStream_t)stream));
    else if (precision == RT_F32) HIP_TRY(rt::launch_trace_primary<float>(scene->s32, scene->features, width, height, time, medium_xi, d_ids, d_t, (hipStream_t)stream));
    else return fail(RT_ERR_INVALID, "unknown precision");
    return RT_OK;
}

static int make_args(const RtRenderParams* p, rt::RenderArgs& a) {
    if (!p) return fail(RT_ERR_INVALID, "params is NULL");
    if (p->width < 2 || p->height < 2 || (uint64_t)p->width * p->height > 0x7fffffffull) return fail(RT_ERR_INVALID, "width/height must be >= 2 and width*height < 2^31");
    if (p->max_depth < 0) return fail(RT_ERR_INVALID, "max_depth must be >= 0");
    if ((uint64_t)p->spp_begin + p->spp_count > 0xffffffffull) return fail(RT_ERR_INVALID, "spp range overflows 32 bits");
    if (p->spp_count >= (1u << 26)) return fail(RT_ERR_INVALID, "spp_count per launch must be < 2^26 (split the range over launches)");
    a.width = p->width; a.height = p->height; a.spp_begin = p->spp_begin; a.spp_count = p->spp_count;
    a.max_depth = p->max_depth; a.max_nan_retries = p->max_nan_retries ? p->max_nan_retries : 64u; a.seed = p->seed;
    return RT_OK;
}

// One render launch with the job-counter word the caller chose (ring slot of the scene, or static scratch).
static int render_launch(RtScene* scene, const RtRenderParams* params, const rt::RenderArgs& a, uint32_t* d_counter, float* d_accum,
                         uint64_t* d_stats, hipStream_t stream) {
    if (params->flags & ~(uint32_t)(RT_RENDER_NO_SORT | RT_RENDER_DETERMINISTIC)) return fail(RT_ERR_INVALID, "unknown bits in RtRenderParams.flags");
    const rt::LaunchCtx ctx{d_counter, scene->cu_count, (params->flags & RT_RENDER_DETERMINISTIC) != 0};
    const bool no_sort = (params->flags & RT_RENDER_NO_SORT) != 0;
    if (params->precision == RT_F64) HIP_TRY(rt::launch_render<double>(scene->s64, scene->features, a, no_sort, ctx, d_accum, (unsigned long long*)d_stats, stream));
    else if (params->precision == RT_F32) HIP_TRY(rt::launch_render<float>(scene->s32, scene->features, a, no_sort, ctx, d_accum, (unsigned long long*)d_stats, stream));
    else return fail(RT_ERR_INVALID, "unknown precision");
    return RT_OK;
}

// Take the next slot of the scene's job-counter ring for a launch on `stream` (see RtScene).
static int ring_acquire(RtScene* scene, hipStream_t stream, uint32_t& slot, bool& capturing) {
    slot = scene->next_counter++ % RtScene::kCounters;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const hipError_t qe = hipStreamIsCapturing(stream, &cs);
    capturing = qe == hipSuccess && cs != hipStreamCaptureStatusNone;
    if (qe != hipSuccess) (void)hipGetLastError();  // (a legacy-stream query during another stream's capture fails and leaves its code
                                                    // behind: that one is dropped; an error pending from before stays for its owner)
    if (scene->slot_recorded[slot] && scene->slot_stream[slot] != stream && !capturing)
        HIP_TRY(hipStreamWaitEvent(stream, scene->slot_event[slot], 0));
    return RT_OK;
}
static int ring_release(RtScene* scene, hipStream_t stream, uint32_t slot, bool capturing) {
    if (capturing) {  // events recorded inside a capture cannot guard launches outside it: the slot is left unguarded
        scene->slot_recorded[slot] = false;
        return RT_OK;
    }
    if (!scene->slot_event[slot]) HIP_TRY(hipEventCreateWithFlags(&scene->slot_event[slot], hipEventDisableTiming));
    HIP_TRY(hipEventRecord(scene->slot_event[slot], stream));
    scene->slot_stream[slot] = stream;
    scene->slot_recorded[slot] = true;
    return RT_OK;
}

int rt_render(RtScene* scene, const RtRenderParams* params, float* d_accum, uint64_t* d_stats, void* stream) {
    int rc = check_scene(scene);
    if (rc) return rc;
    rt::RenderArgs a;
    if ((rc = make_args(params, a))) return rc;
    if (!d_accum) return fail(RT_ERR_INVALID, "d_accum is NULL");
    if (a.spp_count == 0) return RT_OK;
    uint32_t slot; bool capturing;
    if ((rc = ring_acquire(scene, (hipStream_t)stream, slot, capturing))) return rc;
    if ((rc = render_launch(scene, params, a, scene->d_counters + slot, d_accum, d_stats, (hipStream_t)stream))) return rc;
    return ring_release(scene, (hipStream_t)stream, slot, capturing);
}

// ray_tracing.rs:95-118 as ONE call: sample loop + tonemap, 8-bit image out.  The float sums between the two kernels
// live in a work buffer of the scene handle, (re)allocated here when the frame outgrows it.
int rt_render_rgb8(RtScene* scene, const RtRenderParams* params, uint8_t* d_rgb, uint64_t* d_stats, void* stream) {
    int rc = check_scene(scene);
    if (rc) return rc;
    rt::RenderArgs a;
    if ((rc = make_args(params, a))) return rc;
    if (!d_rgb) return fail(RT_ERR_INVALID, "d_rgb is NULL");
    if (a.spp_count == 0) return fail(RT_ERR_INVALID, "spp_count must be > 0 (the image is the mean of the samples)");
    const uint64_t floats = (uint64_t)a.width * a.height * 3u;
    if (floats > 0xffffffffull) return fail(RT_ERR_INVALID, "frame too large for the resolve kernel (width*height*3 must fit 32 bits)");
    if (scene->work_floats < floats) {
        if (scene->d_work) HIP_TRY(hipFree(scene->d_work));  // hipFree waits for the device: no launch still reads it
        scene->d_work = nullptr; scene->work_floats = 0;
        HIP_TRY(hipMalloc((void**)&scene->d_work, floats * sizeof(float)));
        scene->work_floats = floats;
    }
    HIP_TRY(hipMemsetAsync(scene->d_work, 0, floats * sizeof(float), (hipStream_t)stream));
    uint32_t slot; bool capturing;
    if ((rc = ring_acquire(scene, (hipStream_t)stream, slot, capturing))) return rc;
    if ((rc = render_launch(scene, params, a, scene->d_counters + slot, scene->d_work, d_stats, (hipStream_t)stream))) return rc;
    if ((rc = ring_release(scene, (hipStream_t)stream, slot, capturing))) return rc;
    HIP_TRY(rt::launch_resolve(scene->d_work, a.width, a.height, a.spp_count, d_rgb, (hipStream_t)stream));
    return RT_OK;
}

// rt_render_rgb8 for callers that cannot allocate (HIP graph capture, serving loops): the same two kernels, with the
// float sums and the job counter in the library's static device array `scratch`.  Calls no hipMalloc / hipFree /
// event creation.  Every call on a GPU shares that GPU's array: calls that overlap on one GPU race.
int rt_render_rgb8_noalloc(RtScene* scene, const RtRenderParams* params, uint8_t* d_rgb, uint64_t* d_stats, void* stream) {
    int rc = check_scene(scene);
    if (rc) return rc;
    rt::RenderArgs a;
    if ((rc = make_args(params, a))) return rc;
    if (!d_rgb) return fail(RT_ERR_INVALID, "d_rgb is NULL");
    if (a.spp_count == 0) return fail(RT_ERR_INVALID, "spp_count must be > 0 (the image is the mean of the samples)");
    if ((uint64_t)a.width * a.height > RT_NOALLOC_MAX_PIXELS)
        return fail(RT_ERR_INVALID, "rt_render_rgb8_noalloc: frame of " + std::to_string((uint64_t)a.width * a.height) + " pixels exceeds the static scratch (RT_NOALLOC_MAX_PIXELS = " +
                                        std::to_string((uint64_t)RT_NOALLOC_MAX_PIXELS) + "); use rt_render_rgb8 or rt_render");
    unsigned char* base = scene->device >= 0 && scene->device < kMaxDevices ? g_scratch_ptr[scene->device] : nullptr;
    if (!base) return fail(RT_ERR_INVALID, "rt_render_rgb8_noalloc: call rt_init(device) on this device first (it resolves the scratch array's address)");
    uint32_t* d_counter = (uint32_t*)base;                  // first 256 bytes: job counter of the persistent kernel
    float* d_sums = (float*)(base + kScratchHeader);        // then W*H*3 float sums
    const uint64_t floats = (uint64_t)a.width * a.height * 3u;
    HIP_TRY(hipMemsetAsync(d_sums, 0, floats * sizeof(float), (hipStream_t)stream));  // the array keeps the last call's contents
    if ((rc = render_launch(scene, params, a, d_counter, d_sums, d_stats, (hipStream_t)stream))) return rc;  // (the launcher zeroes the counter)
    HIP_TRY(rt::launch_resolve(d_sums, a.width, a.height, a.spp_count, d_rgb, (hipStream_t)stream));
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
    if (params->precision == RT_F64) HIP_TRY(rt::launch_sample_pixel<double>(scene->s64, scene->features, a, x, y, a.spp_count, d_out, (hipStream_t)stream));
    else if (params->precision == RT_F32) HIP_TRY(rt::launch_sample_pixel<float>(scene->s32, scene->features, a, x, y, a.spp_count, d_out, (hipStream_t)stream));
    else return fail(RT_ERR_INVALID, "unknown precision");
    return RT_OK;
}

/* Test hook: d_shared[i] = n[i] / d[i] by the shared-denominator route of rt_device.hpp (Recip), d_plain[i] by `/`. */
int rt_debug_divide(const double* d_n, const double* d_d, uint64_t count, double* d_shared, double* d_plain, void* stream) {
    if (!d_n || !d_d || !d_shared || !d_plain) return fail(RT_ERR_INVALID, "NULL buffer");
    if (count == 0) return RT_OK;
    if (count > 0xffffffffull * 256ull) return fail(RT_ERR_INVALID, "count too large");
    HIP_TRY(rt::launch_divide_check(d_n, d_d, count, d_shared, d_plain, (hipStream_t)stream));
    return RT_OK;
}

/* Test hook: d_out[i] = n[i] / d[i] by the unguarded shared-denominator routine (rt_device.hpp: RecipCore), d_flag[i] = 1 where
 * the denominator lies outside the range for which the routine is IEEE division. */
int rt_debug_divide_fast(const double* d_n, const double* d_d, uint64_t count, double* d_out, uint32_t* d_flag, void* stream) {
    if (!d_n || !d_d || !d_out || !d_flag) return fail(RT_ERR_INVALID, "NULL buffer");
    if (count == 0) return RT_OK;
    if (count > 0xffffffffull * 256ull) return fail(RT_ERR_INVALID, "count too large");
    HIP_TRY(rt::launch_divide_fast(d_n, d_d, count, d_out, d_flag, (hipStream_t)stream));
    return RT_OK;
}

/* Test hook: per input x, eight int32: as_i32(x), as_u32(x), as_i32((float)x), as_u32((float)x) by the device code's conversion, then
 * the same four by the spelled-out form (rt_device.hpp). */
int rt_debug_convert(const double* d_x, uint64_t count, int32_t* d_out, void* stream) {
    if (!d_x || !d_out) return fail(RT_ERR_INVALID, "NULL buffer");
    if (count == 0) return RT_OK;
    if (count > 0xffffffffull * 256ull) return fail(RT_ERR_INVALID, "count too large");
    HIP_TRY(rt::launch_convert_check(d_x, count, d_out, (hipStream_t)stream));
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
