/*
 * rt_hip.h — C-ABI of the MI355X (gfx950) path-tracing hot path.
 *
 * This is the device boundary a Rust `-sys` crate (or any FFI) binds.  The reference
 * (stneng/Ray-Tracing) has NO plugin / FFI interface; the seam this ABI replaces is
 *
 *   pub fn ray_color<T: Object>(ray, world:&ObjectList, background:Vec3, lights:&Option<T>,
 *                               depth:i32, rng:&mut SmallRng) -> Vec3      src/ray_tracing.rs:11-18
 *   the per-pixel sample loop around it                                    src/ray_tracing.rs:95-118
 *   fn <scene>(aspect) -> (Arc<ObjectList>, Vec3, Arc<Camera>, Arc<Option<ObjectList>>)
 *                                                                          src/scenes.rs:7-9,559-561,841-843
 *
 * The host keeps the reference's Object / Material / Texture scene vocabulary and hands this
 * library a FLATTENED, pointer-free scene (RtSceneDesc): BvhNode trees linearised into
 * structure-of-arrays node/AABB buffers, primitives / materials / textures packed as tagged
 * records, Translate/RotateY wrappers as per-instance transform chains.  Everything is `f64`, the
 * reference's only float type (src/vec3.rs:4-9); the library converts to its device layouts.
 *
 * Conventions: plain C, opaque handle, `int` status codes + rt_last_error(), no exceptions or
 * aborts across the boundary.  The caller owns every buffer it passes; `d_*` arguments are DEVICE
 * pointers (hipMalloc / torch tensor data_ptr) on the device selected with rt_init().  `stream`
 * is a hipStream_t passed as void* (NULL = the default stream); calls are asynchronous on it.
 * One scene handle belongs to one device and must be used from one host thread at a time.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 1u /* layout of RtSceneDesc / RtRenderParams; entry points are only ever added */

enum RtStatus {
    RT_OK = 0,
    RT_ERR_INVALID = 1,     /* bad argument / malformed scene description */
    RT_ERR_HIP = 2,         /* a HIP runtime call failed (message in rt_last_error) */
    RT_ERR_UNSUPPORTED = 3, /* scene graph shape the flat form cannot express */
    RT_ERR_NO_DEVICE = 4
};

/* Arithmetic type the device path computes in. RT_F64 restates the reference's f64 arithmetic
 * operation-for-operation (primitive ids and t bit-exact against the oracle, frames equal to the oracle's
 * sample for sample: this is the mode every parity claim is made for; one stated exception: the vector divisions
 * unit(v) and v / s go through a shared-reciprocal routine that is IEEE division for denominators |d| in
 * [2^-400, 2^400) and numerators |n| in [2^-600, 2^600] or 0 — a scene with a radius or a direction length outside
 * that range may differ from the reference in the last bit of a normal).  RT_F32 is a throughput mode outside
 * the parity bar: the tests hold it to >= 99.5 % identical ids and t within 5e-4 relative on the probe (f32
 * cancellation in the sphere quadratic at coordinates ~1000), and its images are an independent estimate at the
 * Monte-Carlo noise floor — for scenes within f32's reach: once coordinates exceed ~10^4 times the 0.001 ray offset
 * (|x| >~ 8000, where one f32 ulp is that offset) bounces re-hit their own surface and the estimate is biased; every
 * launch still ends and every sample is counted (tools/gpu_f32_sanity.py). */
enum RtPrecision { RT_F32 = 0, RT_F64 = 1 };

/* ---- primitives: src/objects.rs:81-85,146-153,217-225,271-279,325-333,381-392 ------------- */
enum RtPrimKind {
    RT_PRIM_SPHERE = 0,        /* d = center(3), radius */
    RT_PRIM_MOVING_SPHERE = 1, /* d = center1(3), center2(3), t1, t2, radius */
    RT_PRIM_RECT_XY = 2,       /* d = x1, x2, y1, y2, k, face(+-1) */
    RT_PRIM_RECT_XZ = 3,       /* d = x1, x2, z1, z2, k, face */
    RT_PRIM_RECT_YZ = 4,       /* d = y1, y2, z1, z2, k, face */
    RT_PRIM_CUBOID = 5         /* d = box_min(3), box_max(3); six sides -z,+z,-y,+y,-x,+x as objects.rs:399-452 */
};
typedef struct RtPrim {
    uint32_t kind;     /* RtPrimKind */
    uint32_t material; /* index into materials */
    double d[10];
} RtPrim;

/* ---- materials: src/materials.rs:31-173 --------------------------------------------------- */
enum RtMaterialKind {
    RT_MAT_LAMBERTIAN = 0,        /* texture = albedo */
    RT_MAT_METAL = 1,             /* albedo[3], fuzz */
    RT_MAT_DIELECTRIC = 2,        /* ref_idx */
    RT_MAT_DIFFUSE_LIGHT = 3,     /* texture = emit */
    RT_MAT_ISOTROPIC = 4,         /* texture = albedo (ConstantMedium phase function) */
    RT_MAT_FROSTED_DIELECTRIC = 5 /* ref_idx, fuzz */
};
typedef struct RtMaterial {
    uint32_t kind;    /* RtMaterialKind */
    uint32_t texture; /* index into textures (Lambertian / DiffuseLight / Isotropic) */
    double albedo[3];
    double fuzz;
    double ref_idx;
} RtMaterial;

/* ---- textures: src/texture.rs:11-98 ------------------------------------------------------- */
enum RtTextureKind {
    RT_TEX_SOLID = 0,      /* color */
    RT_TEX_CHECKER_3D = 1, /* odd, even: texture indices (CheckerTexture, sines) */
    RT_TEX_CHECKER_UV = 2, /* odd, even (CheckerTextureUV) */
    RT_TEX_NOISE = 3,      /* table = perlin index, scale */
    RT_TEX_IMAGE = 4       /* table = image index */
};
typedef struct RtTexture {
    uint32_t kind; /* RtTextureKind */
    uint32_t odd;
    uint32_t even;
    uint32_t table;
    double color[3];
    double scale;
} RtTexture;

/* src/texture.rs:100-106 */
typedef struct RtPerlin {
    int32_t perm_x[256];
    int32_t perm_y[256];
    int32_t perm_z[256];
    double ranvec[256][3];
} RtPerlin;

/* src/texture.rs:63-66: decoded RGB8 raster, row 0 = top of the file; HOST pointer, copied. */
typedef struct RtImage {
    uint32_t width;
    uint32_t height;
    const uint8_t* rgb;
} RtImage;

/* ---- instancing: src/transforms.rs:5-8,42-47 ---------------------------------------------- */
enum RtXformKind {
    RT_XFORM_TRANSLATE = 0, /* v = offset */
    RT_XFORM_ROTATE_Y = 1   /* v[0] = sin_theta, v[1] = cos_theta */
};
typedef struct RtXform {
    uint32_t kind;
    uint32_t reserved;
    double v[3];
} RtXform;

enum RtGeomKind {
    RT_GEOM_PRIM = 0, /* geom_index = primitive index */
    RT_GEOM_BVH = 1   /* geom_index = root node index in the BVH buffers */
};
/* One entry of an ObjectList (src/objects.rs:29-31): a primitive or a BVH, wrapped outermost-first
 * by xforms[xform_begin .. xform_begin+xform_count) (Translate(RotateY(x)) = {TRANSLATE, ROTATE_Y}),
 * optionally the boundary of a ConstantMedium (src/objects.rs:514-527). */
typedef struct RtInstance {
    uint32_t geom_kind; /* RtGeomKind */
    uint32_t geom_index;
    uint32_t xform_begin;
    uint32_t xform_count;
    uint32_t is_medium;      /* 1: ConstantMedium whose boundary is this geometry */
    uint32_t phase_material; /* Isotropic material index when is_medium */
    double density;
} RtInstance;

/* ---- BVH: src/bvh.rs:79-83 linearised, structure-of-arrays -------------------------------- */
/* Internal nodes only (a leaf IS the primitive, bvh.rs:97-99).  Node i: box min/max, children.
 * child >= 0: internal node index; child < 0: leaf, primitive index = ~child.
 * Nodes of one tree are stored in pre-order (node, left subtree, right subtree). */
typedef struct RtBvh {
    uint32_t n_nodes;
    const double* min_x; const double* min_y; const double* min_z;
    const double* max_x; const double* max_y; const double* max_z;
    const int32_t* left;
    const int32_t* right;
} RtBvh;

/* src/camera.rs:7-18 */
typedef struct RtCamera {
    double origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3], w[3];
    double lens_radius, t1, t2;   /* shutter: t1 < t2, both finite, also after rounding to f32 (gen_range, src/camera.rs:63) */
} RtCamera;

/* The flattened scene = the 4-tuple returned by the scenes.rs constructors. */
typedef struct RtSceneDesc {
    uint32_t abi_version; /* RT_ABI_VERSION */
    uint32_t n_prims;     const RtPrim* prims;
    uint32_t n_materials; const RtMaterial* materials;
    uint32_t n_textures;  const RtTexture* textures;
    uint32_t n_perlins;   const RtPerlin* perlins;
    uint32_t n_images;    const RtImage* images;
    uint32_t n_xforms;    const RtXform* xforms;
    uint32_t n_world;     const RtInstance* world;  /* ObjectList order is semantically relevant; 0 objects is a valid (all-background) world */
    uint32_t n_lights;    const RtInstance* lights; /* 0 <=> lights = None.  Rectangle / cuboid lights need low < high on every sampled
                                                       axis, finite, also after rounding to f32 (the reference's gen_range panics otherwise) */
    RtBvh bvh;
    double background[3];
    RtCamera camera;
} RtSceneDesc;

typedef struct RtScene RtScene; /* opaque; owns the device copies */

/* ray_tracing.rs:95-111 parameters of one render launch. Samples [spp_begin, spp_begin+spp_count)
 * of every pixel; the RNG stream of a sample is keyed by (seed, pixel, sample index), so the
 * union over launches / GPUs does not depend on how the spp range was sharded. */
typedef struct RtRenderParams {
    uint32_t width, height;
    uint32_t spp_begin, spp_count;
    int32_t max_depth;        /* 50 in the reference (ray_tracing.rs:105) */
    uint32_t max_nan_retries; /* cap on the NaN retry loop of ray_tracing.rs:102-110 (0 -> 64) */
    uint64_t seed;
    uint32_t precision;       /* RtPrecision */
    uint32_t flags;           /* RtRenderFlags, 0 = default */
} RtRenderParams;
/* RT_RENDER_NO_SORT: use the wave-level kernel even where the block-sorted one applies (same samples, same
 * sums up to the order of the additions; for A/B measurements and tests).
 *
 * RT_RENDER_DETERMINISTIC: a reproducible accumulation mode.  The reference sums a pixel's samples into one f64 in
 * sample order (src/ray_tracing.rs:98-112).  By DEFAULT this library does not: the block-sorted kernel adds a tile's
 * samples to per-pixel f64 sums in LDS in the order its four waves finish them, and every launch with fewer than 64
 * tiles per resident wave / block (every frame below ~12 Mpixel) is cut into sample chunks whose partial sums reach
 * d_accum through f32 atomics in arrival order — two identical calls can differ in the last bit of a sum, and a grey
 * level of the resolved image can flip.  With this flag ONE lane renders all of a pixel's samples of the launch, in
 * index order, sums them in f64 in that order and adds the rounded sum to d_accum with one plain store: the same call
 * gives the same bits on every run (and on every gfx950), and d_accum of a zeroed buffer is the reference's sum
 * rounded to f32.  Cost: the wave-level kernel without sample chunks, block regrouping or streaming jobs, and a wave
 * is as slow as the lane with the longest paths; measured on one MI355X, f64, share of the default throughput: earth
 * 0.83, cornell_box 1024^2 @ 1024 spp 0.69 (@ 128 spp 0.64), final_scene 0.39, random_scene 0.33.
 * Implies RT_RENDER_NO_SORT. */
enum RtRenderFlags { RT_RENDER_NO_SORT = 1, RT_RENDER_DETERMINISTIC = 2 };

/* d_stats: 4 x uint64 counters, ADDED to by rt_render:
 * [0] accepted samples, [1] ray casts (world.hit calls), [2] NaN retries, [3] samples dropped at the retry cap */
#define RT_STATS_WORDS 4

/* Select the device for the calling thread (hipSetDevice) and check that it is a gfx950. */
int rt_init(int device);
const char* rt_last_error(void);
/* Human-readable build/device description, e.g. "rt_hip abi=1 gfx950 cu=256 clock_mhz=2400 build=3f9c0a1b2c4d"
 * (build = hash of the device sources and flags the library was compiled from). */
const char* rt_describe(void);

/* Upload a flattened scene. Replaces: handing `&ObjectList`/`&Option<ObjectList>`/`Camera` to
 * ray_color and get_ray (src/ray_tracing.rs:78,90-92,103-106). Validates every index. */
int rt_scene_create(const RtSceneDesc* desc, RtScene** out);
int rt_scene_destroy(RtScene* scene);

/* Deterministic parity probe (closest hit of world.hit(ray, 0.001, f64::MAX), ray_tracing.rs:22,
 * objects.rs:38-48) for the jitter-free camera grid u = x/(W-1), v = y/(H-1), lens offset 0,
 * ray.time = time; every ConstantMedium uses xi = medium_xi instead of a random draw.
 * Any finite `time` is accepted.  Inside [min(0, camera.t1), max(1, camera.t2)] BVH leaves of moving spheres are
 * culled by their time-hull boxes; outside that range the leaf culling is switched off (every reached leaf is
 * tested exactly).  At ray times for which the description's node boxes do not contain their moving spheres (the
 * reference builds them for (0, 1), bvh.rs:85-111) the reference's loose per-axis box test still reaches some of
 * those spheres: the walk then takes every node decision with the reference's own test.  Exact either way, only slower.
 * d_ids[y*W+x] = prim*8 + side (side: cuboid face 0..5, 7 = medium event) or -1 on a miss;
 * d_t[y*W+x] = hit distance (f64 even in RT_F32 mode) or -1. */
int rt_trace_primary(RtScene* scene, uint32_t width, uint32_t height, double time, double medium_xi,
                     uint32_t precision, int32_t* d_ids, double* d_t, void* stream);

/* The hot path: ray_tracing.rs:95-111 (sample loop, get_ray, ray_color depth 50, NaN retry).
 * d_accum[(y*W+x)*3+c] += sum of accepted radiance over the spp range (linear, un-normalised f32;
 * y counts upward as in ray_tracing.rs:97).  d_stats may be NULL. */
int rt_render(RtScene* scene, const RtRenderParams* params, float* d_accum, uint64_t* d_stats, void* stream);

/* ray_tracing.rs:112-117,126-128: /spp, sqrt, clamp [0,0.99999], *256 as u8, written to image
 * row H-1-y.  d_rgb: W*H*3 bytes, row-major, top row first. */
int rt_resolve(const float* d_accum, uint32_t width, uint32_t height, uint32_t spp, uint8_t* d_rgb, void* stream);

/* The whole per-pixel block src/ray_tracing.rs:95-118 in one call (sample loop AND tonemap: the reference's jobs
 * return `Vec<[u8;3]>`): d_rgb = 8-bit image (top row first) of the mean of samples [spp_begin, spp_begin+spp_count).
 * The float sums between the two kernels live in a work buffer owned by the scene handle; this call hipMallocs it
 * when the frame outgrows it (so it must not be used under stream capture: see rt_render_rgb8_noalloc). */
int rt_render_rgb8(RtScene* scene, const RtRenderParams* params, uint8_t* d_rgb, uint64_t* d_stats, void* stream);

/* No counterpart in the reference.  rt_render_rgb8 for callers that CANNOT ALLOCATE (HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc): same kernels, same output, but the float sums and the job
 * counter live in a static device array of the library (RT_NOALLOC_MAX_PIXELS x 12 bytes + 256, held on every GPU
 * the library runs on until the process ends).  ("Same output": the same as rt_render_rgb8 with the same flags — bit for bit
 * with RT_RENDER_DETERMINISTIC, and up to the order of the additions without it, see RtRenderFlags.)
 * Calls no hipMalloc / hipFree / event creation.  Frames of more than
 * RT_NOALLOC_MAX_PIXELS pixels are rejected with RT_ERR_INVALID.  rt_init(device) must have run on the device.
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two scenes) race. */
#define RT_NOALLOC_MAX_PIXELS (1024u * 1024u)
int rt_render_rgb8_noalloc(RtScene* scene, const RtRenderParams* params, uint8_t* d_rgb, uint64_t* d_stats, void* stream);

/* ---- multi-GPU exchange step -----------------------------------------------------------------------------
 * Replaces the one gather the reference has: worker jobs -> mpsc channel -> the single image
 * (src/ray_tracing.rs:84,119-131).  Here every GPU renders a disjoint sample range of the FULL frame
 * (RtRenderParams.spp_begin/spp_count) into its own float buffer and the buffers are summed in place by ONE
 * RCCL all-reduce over xGMI; afterwards every rank holds the sums of all samples (resolve on any rank).
 * One process (or host thread) per GPU; RCCL (librccl.so.1) is loaded on first use — RT_ERR_UNSUPPORTED if absent.
 *   rank 0:      rt_comm_unique_id(id)  and ships the 128 bytes to the other ranks by any means (file, socket, MPI)
 *   every rank:  rt_init(local device);  rt_comm_create(id, n_ranks, rank, &comm)   (collective: blocks until all joined)
 *   per render:  rt_render(...);  rt_allreduce_accum(comm, d_accum, W*H*3, stream);  [rt_allreduce_stats]
 * rt_comm_adopt wraps a communicator the caller already owns (an ncclComm_t); rt_comm_destroy does not destroy it. */
#define RT_COMM_ID_BYTES 128
typedef struct RtComm RtComm;
int rt_comm_unique_id(uint8_t id[RT_COMM_ID_BYTES]);
int rt_comm_create(const uint8_t id[RT_COMM_ID_BYTES], int n_ranks, int rank, RtComm** out);
int rt_comm_adopt(void* nccl_comm, RtComm** out);
int rt_comm_destroy(RtComm* comm);
/* d_accum[i] <- sum over ranks of d_accum[i], i < count (f32, in place), asynchronous on `stream`. */
int rt_allreduce_accum(RtComm* comm, float* d_accum, uint64_t count, void* stream);
/* d_stats[0..RT_STATS_WORDS) <- sum over ranks (u64, in place). */
int rt_allreduce_stats(RtComm* comm, uint64_t* d_stats, void* stream);

/* Test hook, no counterpart in the reference: radiance of samples [spp_begin, spp_begin+spp_count)
 * of ONE pixel (x, y counted upward), d_out[3*i..] = sample spp_begin+i, for sample-for-sample
 * comparison of the RT_F64 path with the CPU oracle (same keyed RNG stream). */
int rt_debug_sample_pixel(RtScene* scene, const RtRenderParams* params, uint32_t x, uint32_t y, double* d_out, void* stream);

/* Test hook, no counterpart in the reference: the device path divides through a shared-denominator routine
 * (rt_device.hpp: Recip) that must equal IEEE division bit for bit; d_shared[i] = n[i]/d[i] by that routine,
 * d_plain[i] by the compiler's `/`. */
int rt_debug_divide(const double* d_n, const double* d_d, uint64_t count, double* d_shared, double* d_plain, void* stream);
/* Test hook, no counterpart in the reference: the UNGUARDED shared-denominator quotient routine that unit(v) and v / s use
 * on the device (rt_device.hpp: RecipCore); d_flag[i] = 1 where |d[i]| lies outside [2^-400, 2^400). */
int rt_debug_divide_fast(const double* d_n, const double* d_d, uint64_t count, double* d_out, uint32_t* d_flag, void* stream);
/* Test hook, no counterpart in the reference: the float-to-integer conversions of the textures (Rust `as i32` / `as u32`: texture.rs:29-30,
 * 82-89,143-145).  Per input x eight int32 in d_out[8*i..]: as_i32(x), as_u32(x), as_i32((float)x), as_u32((float)x) by the conversion
 * the device code uses (the hardware's saturating instructions), then the same four by the form that spells Rust's rules out. */
int rt_debug_convert(const double* d_x, uint64_t count, int32_t* d_out, void* stream);

/* Small helpers so a host without another HIP binding can drive the library. */
int rt_device_alloc(void** d_ptr, uint64_t bytes);
int rt_device_free(void* d_ptr);
int rt_device_zero(void* d_ptr, uint64_t bytes, void* stream);
int rt_copy_to_host(void* h_dst, const void* d_src, uint64_t bytes, void* stream); /* synchronises the stream */
int rt_stream_synchronize(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
