// rt_device.hpp — device-side scene layouts, math, RNG, intersection and textures for gfx950,
// templated on the arithmetic type T (double: restates the reference's f64 arithmetic operation
// for operation; float: throughput mode) and on a FEATURE MASK F that removes, at compile time,
// every code path a scene cannot reach (a wave executes the union of its lanes' paths, so on this
// machine code that is never needed is best not present: registers, I-cache).
//
// The reference's recursive, dyn-dispatched ray_color (src/ray_tracing.rs:11-60) becomes an
// iterative per-lane state machine; the Object/Material/Texture/PDF traits become switches over
// tagged records.  Every routine cites the reference lines whose arithmetic (evaluation order,
// strict/inclusive comparisons, NaN behaviour) it keeps.  Where the reference computes values it
// then discards (sphere uv before the range test, the ONB tangents of a CosinePDF that is only
// evaluated) they are not computed: values that are used are bit-identical.
//
// Branch policy (wave64): short conditionals are written as selects so the wave stays converged;
// real branches are kept only where a whole wave commonly skips the work (sphere discriminant,
// material kinds, BVH traversal).
#pragma once
#if defined(RT_HOST_SIM)
#include "rt_hostsim_shim.hpp"  // tests/hostsim: runs these templates on the CPU against the oracle (test-only)
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

// tests/hostsim can count traversal events (node classifications, exact fallbacks, leaf tests); a no-op on the device
#if defined(RT_HOST_SIM) && defined(RT_SIM_COUNTERS)
extern "C" unsigned long long g_sim_counters[8];
extern "C" unsigned long long g_sim_walk_hist[64];  // walks by number of node classifications, bins of 8
#define RT_SIM_COUNT(i) (g_sim_counters[i]++)
#define RT_SIM_WALK(n0) (g_sim_walk_hist[(g_sim_counters[0] - (n0)) / 8 > 63 ? 63 : (g_sim_counters[0] - (n0)) / 8]++)
#else
#define RT_SIM_COUNT(i) ((void)0)
#define RT_SIM_WALK(n0) ((void)0)
#endif

// Profiling build (-DRT_PHASE_CLOCK, tools/build_ab.py): per-wave cycle and active-lane sums per phase of the
// wave-level render kernel, printed for a few blocks.  Slots: [i] cycles, [16+i] cycles x active lanes.
#if defined(RT_PHASE_CLOCK) && defined(__HIP_DEVICE_COMPILE__)
__device__ inline unsigned long long* rt_pc() { __shared__ unsigned long long pc[16][32]; return pc[threadIdx.x >> 6]; }
#define RT_PCB() unsigned long long pc_t_ = __builtin_readcyclecounter()
#define RT_PCE(i)                                                                                        \
    do {                                                                                                 \
        const unsigned long long n_ = __builtin_readcyclecounter(), m_ = __builtin_amdgcn_ballot_w64(true); \
        if ((threadIdx.x & 63u) == (unsigned)__builtin_ctzll(m_)) {                                      \
            rt_pc()[i] += n_ - pc_t_; rt_pc()[16 + (i)] += (n_ - pc_t_) * (unsigned long long)__builtin_popcountll(m_); \
        }                                                                                                \
        pc_t_ = __builtin_readcyclecounter();                                                            \
    } while (0)
#else
#define RT_PCB() ((void)0)
#define RT_PCE(i) ((void)0)
#endif

namespace rt {

// ------------------------------------------------------------------------------------------
// feature mask
// ------------------------------------------------------------------------------------------
enum : uint32_t {
    F_BVH = 1u << 0,         // a world / medium instance is a BvhNode tree
    F_MEDIUM = 1u << 1,      // ConstantMedium (+ Isotropic)
    F_MOVING = 1u << 2,      // MovingSphere
    F_XFORM = 1u << 3,       // Translate / RotateY wrappers
    F_SPHERE = 1u << 4,
    F_RECT = 1u << 5,
    F_CUBOID = 1u << 6,
    F_METAL = 1u << 7,
    F_DIELECTRIC = 1u << 8,  // Dielectric or FrostedDielectric
    F_FROSTED = 1u << 9,
    F_LIGHTS = 1u << 10,     // lights = Some(..): mixture importance sampling
    F_TEXTURED = 1u << 11,   // any non-SolidColor texture (checker / noise / image)
    F_UV = 1u << 12,         // some material needs the hit's (u,v)
    F_NOISE = 1u << 13,      // a NoiseTexture exists: the render kernels stage Perlin tables in LDS
    F_MEDIUM_BVH = 1u << 14, // a ConstantMedium whose boundary is a BvhNode tree (no scene of scenes.rs has one): both boundary searches
                             // are then tree walks, and the search range of a medium stays live across the walk code
    F_ALL = 0xffffffffu
};
template <uint32_t F> __device__ constexpr bool has(uint32_t bit) { return (F & bit) != 0; }

// ------------------------------------------------------------------------------------------
// device layouts (built by rt_pack.hpp from the f64 RtSceneDesc)
// ------------------------------------------------------------------------------------------
template <class T> struct DPrim { T d[10]; uint32_t kind, material; };
template <class T> struct DMaterial { T albedo[3]; T fuzz, ref_idx; uint32_t kind, texture; };
template <class T> struct DTexture { T color[3]; T scale; uint32_t kind, odd, even, table; };
// Perlin table (texture.rs:100-105).  The render kernels stage copies in LDS: f32 vectors are padded to 4
// components (one ds_read_b128 per vector); f64 vectors are not (two tables + the sort buffers must fit 3 blocks per CU)
template <class T> struct alignas(16) DPerlin {
    static constexpr uint32_t kStride = sizeof(T) == 4 ? 4u : 3u;
    T ranvec[256][kStride];
    uint8_t perm[3][256];
};
struct DImage { const uint8_t* rgb; uint32_t width, height; };
template <class T> struct DXform { T v[3]; uint32_t kind; };
// `prim`: a COPY of prims[geom_index] where the geometry is one primitive (geom_kind 0; zeros for a tree).  The scans over the world
// and the light list read an entry and its primitive with one address: as two records the primitive's address came out of the
// instance's, a second scalar-load round trip per list entry of every cast.
template <class T> struct alignas(16) DInstance { T density; uint32_t geom_kind, geom_index, xform_begin, xform_count, is_medium, phase_material; DPrim<T> prim; };
// The whole record in one go (16-byte pieces, which the backend merges into the widest scalar loads): read field by field through a
// reference, each field is loaded where it is first needed — behind the branch on the field before — and a list entry costs four to
// five dependent scalar-load round trips instead of one.
// `p` must be the same address in every lane of the wave (the list scans' loop counters are): readfirstlane hands all lanes lane 0's.
template <class T> __device__ inline DInstance<T> load_record(const DInstance<T>* p) {
#if defined(RT_HOST_SIM) || !defined(__HIP_DEVICE_COMPILE__)
    return *p;
#else
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    constexpr int N = (int)(sizeof(DInstance<T>) / 16);
    static_assert(sizeof(DInstance<T>) % 16 == 0, "DInstance is read in 16-byte pieces");
    union U { DInstance<T> r; u32x4 q[N]; __device__ U() {} } u;
    const u32x4* src = reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int k = 0; k < N; k++) u.q[k] = src[k];
    // every dword passes through readfirstlane — free for a value that sits in a scalar register, and a convergent operation, which
    // the compiler may not make dependent on a branch: none of the loads can be moved below the test of another field
    union W { u32x4 q[N]; int w[N * 4]; __device__ W() {} } x;
#pragma unroll
    for (int k = 0; k < N; k++) x.q[k] = u.q[k];
#pragma unroll
    for (int k = 0; k < N * 4; k++) x.w[k] = __builtin_amdgcn_readfirstlane(x.w[k]);
#pragma unroll
    for (int k = 0; k < N; k++) u.q[k] = x.q[k];
    return u.r;
#endif
}
// A list entry as the scan sees it: the copy read in one go, or (EAGER false) the record in memory, read field by field.  (The copy
// must not be indexed with a run-time subscript: it would stop being registers.)
template <bool EAGER, class T> struct RecordOf;
template <class T> struct RecordOf<true, T> { DInstance<T> r; __device__ explicit RecordOf(const DInstance<T>* p) : r(load_record(p)) {} };
template <class T> struct RecordOf<false, T> { const DInstance<T>& r; __device__ explicit RecordOf(const DInstance<T>* p) : r(*p) {} };
// Threaded BVH: the reference's fixed left-then-right recursion (src/bvh.rs:114-129) needs no
// stack when internal nodes AND leaves are laid out in pre-order and each internal node knows
// where its subtree ends: box hit -> next entry, box miss -> skip.  prim >= 0 marks a leaf.
template <class T> struct DBvhEntry { T mn[3], mx[3]; int32_t skip, prim; };
// Compact copy of a BVH entry for the division-free node classification: the box in f32, rounded
// OUTWARD (so it contains the exact box), same skip/prim links.  32 bytes: this is what is staged in LDS.
struct alignas(16) DBvhLite { float mn[3], mx[3]; int32_t skip, prim; };
template <class T> struct DCamera { T origin[3], llc[3], horizontal[3], vertical[3], u[3], v[3], w[3]; T lens_radius, t1, t2; };

template <class T> struct DScene {
    const DPrim<T>* prims;
    const DMaterial<T>* materials;
    const DTexture<T>* textures;
    const DPerlin<T>* perlins;
    const DImage* images;
    const DXform<T>* xforms;
    const DInstance<T>* world;
    const DInstance<T>* lights;
    const DBvhEntry<T>* bvh;        // the REFERENCE's trees, threaded, exact boxes (walk_run: exact emulation of bvh.rs:114-129)
    const int32_t* bvh_root_entry;  // node index (RtBvh numbering) -> entry index, for instance roots
    const DBvhLite* bvh_lite;       // same indexing as bvh
    const DBvhLite* fast_lite;      // the device's OWN trees over the same leaves (SAH, rt_pack.hpp), threaded (walk_fast)
    const int32_t* fast_root_entry; // node index (RtBvh numbering) -> entry index in fast_lite
    // The world scan's ZONE: the leading run of the list before its first ConstantMedium.  Its entries are TESTED in the order
    // world_order[0..n_zone) — plain primitives first, trees last, so that a walk starts with the closest distance the cheap tests
    // found (final_scene: 35 % fewer node steps, 44 % fewer leaf tests per cast) — and re-scanned in list order by the rays that met
    // two equal distances on the way (world_hit).  n_zone = 0: the list order is already that order.
    const uint32_t* world_order;
    uint32_t n_zone;
    const DBvhLite* bvh_lds;        // entries [0, n_bvh_lds) of fast_lite staged in LDS by the render kernel (else n_bvh_lds = 0)
    uint32_t n_bvh, n_bvh_lds;      // n_bvh: entries of fast_lite (= of bvh: same leaves, same number of internal nodes)
    const DPerlin<T>* perlin_lds;     // perlins [0, n_perlin_lds) staged in LDS by the render kernel (else n_perlin_lds = 0)
    uint32_t n_perlins, n_perlin_lds;
    unsigned long long* tex_cache;    // -DRT_TEX_LDS_CACHE=<entries> builds: LDS texel cache of the render kernel (else unused)
    T time_lo, time_hi;             // ray times for which the leaf boxes of moving spheres are valid
    T fast_lo, fast_hi;             // ray times for which walk_fast may run: time_lo..time_hi, or empty when the description's node boxes
                                    // are not nested (rt_pack.hpp)
    T node_cull_lo, node_cull_hi;   // ray times at which every moving sphere lies inside all its ancestors' boxes in the REFERENCE's
                                    // trees (rt_pack.hpp); outside, walk_run may not skip a node on geometric grounds
    uint32_t n_world, n_lights, n_materials;
    T background[3];
    DCamera<T> cam;
};

// Job list of the persistent render kernel: job j = sample chunk j / n_tiles of 8x8 tile j % n_tiles
struct JobGrid {
    uint32_t* counter;  // zeroed before the launch; waves draw jobs with atomicAdd until it passes n_jobs
    uint32_t tiles_x, n_tiles, n_chunks, chunk_spp, n_jobs;
};

struct RenderArgs {
    uint32_t width, height, spp_begin, spp_count;
    int32_t max_depth;
    uint32_t max_nan_retries;
    uint64_t seed;
};

// ------------------------------------------------------------------------------------------
// math helpers
// ------------------------------------------------------------------------------------------
template <class T> struct Lim;
template <> struct Lim<double> { static __device__ constexpr double max() { return 1.7976931348623157e308; } };
template <> struct Lim<float> { static __device__ constexpr float max() { return 3.402823466e+38f; } };

__device__ inline double sqrt_r(double x) { return ::sqrt(x); }
__device__ inline float sqrt_r(float x) { return ::sqrtf(x); }
// ocml's general f64 routines (sin with its large-argument reduction, log; atan2 / asin in sphere_uv_call) are real CALLS in the f64
// kernels, not inlined code: each wants ~60 registers, and inlined into shading code where the whole path state is live they were
// what the f64 kernels spilled for.  Round 3, sin (checker and noise textures) as a call: noise variant 112 -> 16 B of scratch per
// lane, spheres + BVH 160 -> 80 B; two_perlin_spheres 4800 -> 5640 Msamples/s, final_scene 477 -> 517, random_scene 2675 -> 2755.
// log (ConstantMedium's free-flight distance): generic variant 96 -> 48 B, +0.3 .. 0.6 %.  (-DRT_INLINE_LIBM: the inlined form.)
#if !defined(RT_INLINE_LIBM) && !defined(RT_HOST_SIM)
__device__ __attribute__((noinline)) inline double rsin(double x) { return ::sin(x); }
#else
__device__ inline double rsin(double x) { return ::sin(x); }
#endif
__device__ inline float rsin(float x) { return ::sinf(x); }
__device__ inline void rsincos(double x, double& s, double& c) { ::sincos(x, &s, &c); }
// sin and cos of an angle KNOWN to lie in [0, 2 pi] (the azimuth 2 pi r1 of a sampled direction, vec3.rs:231-248).  ocml's f64 sincos
// is a general routine (argument reduction for any magnitude, ~75 executed instructions, ~60 registers: inlined into the Lambertian
// bounce it was what the block-sorted Cornell kernel spilled for).  With the range known the reduction is one Cody-Waite step
// (fdlibm's __ieee754_rem_pio2 medium path: k <= 4, so k * pio2_1 is exact) followed by fdlibm's __kernel_sin / __kernel_cos
// polynomials (FreeBSD msun's branch-free form of the latter).  Checked against quad precision on 2e7 arguments (uniform and next to
// the multiples of pi/2): absolute error <= 8.7e-17, <= 0.78 ulp wherever |result| > 1e-6 (glibc: 0.56) — the same class of
// "<= 1 ulp libm" difference the device's ocml already had against the oracle's glibc.  The CPU build of these templates
// (tests/hostsim) keeps ::sincos.
struct SinCos { double s, c; };
#if defined(RT_HOST_SIM) || !defined(__HIP_DEVICE_COMPILE__) || defined(RT_OCML_SINCOS)
__device__ inline SinCos sincos_0_2pi(double x) { SinCos r; ::sincos(x, &r.s, &r.c); return r; }
#else
#ifdef RT_SINCOS_INLINE
__device__ inline SinCos sincos_0_2pi(double x) {
#else
__device__ __attribute__((noinline)) inline SinCos sincos_0_2pi(double x) {
#endif
    double s, c;
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
    const int k = (int)(x * two_over_pi + 0.5);
    const double fn = (double)k;
    const double r = x - fn * pio2_1;
    const double w = fn * pio2_1t;
    const double y0 = r - w;
    const double y1 = (r - y0) - w;
    const double z = y0 * y0;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double v = z * y0;
    const double rs = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, S6, S5), S4), S3), S2);
    const double sn = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * S1);
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double w2 = z * z;
    const double rc = z * __builtin_fma(z, __builtin_fma(z, C3, C2), C1) + (w2 * w2) * __builtin_fma(z, __builtin_fma(z, C6, C5), C4);
    const double hz = 0.5 * z;
    const double wc = 1.0 - hz;
    const double cs = wc + (((1.0 - wc) - hz) + (z * rc - y0 * y1));
    const bool odd = (k & 1) != 0;
    const double a = odd ? cs : sn, b = odd ? sn : cs;
    s = (k & 2) ? -a : a;
    c = ((k + 1) & 2) ? -b : b;
    SinCos out; out.s = s; out.c = c;
    return out;
}
#endif
__device__ inline void rsincos_0_2pi(double x, double& s, double& c) { const SinCos r = sincos_0_2pi(x); s = r.s; c = r.c; }
__device__ inline void rsincos_0_2pi(float x, float& s, float& c) { ::sincosf(x, &s, &c); }
__device__ inline void rsincos(float x, float& s, float& c) { ::sincosf(x, &s, &c); }
__device__ inline double ratan2(double y, double x) { return ::atan2(y, x); }
__device__ inline float ratan2(float y, float x) { return ::atan2f(y, x); }
__device__ inline double rasin(double x) { return ::asin(x); }
__device__ inline float rasin(float x) { return ::asinf(x); }
#if !defined(RT_INLINE_LIBM) && !defined(RT_HOST_SIM)
__device__ __attribute__((noinline)) inline double rlog(double x) { return ::log(x); }
#else
__device__ inline double rlog(double x) { return ::log(x); }
#endif
__device__ inline float rlog(float x) { return ::logf(x); }
__device__ inline double rfloor(double x) { return ::floor(x); }
__device__ inline float rfloor(float x) { return ::floorf(x); }
__device__ inline double rabs(double x) { return ::fabs(x); }
__device__ inline float rabs(float x) { return ::fabsf(x); }
__device__ inline double rmin(double a, double b) { return ::fmin(a, b); }  // IEEE minNum == Rust f64::min
__device__ inline float rmin(float a, float b) { return ::fminf(a, b); }
__device__ inline double rmax(double a, double b) { return ::fmax(a, b); }
__device__ inline float rmax(float a, float b) { return ::fmaxf(a, b); }

// ------------------------------------------------------------------------------------------
// Division with a SHARED denominator.
// The reference divides (true f64 `/`).  hipcc expands an f64 division into  div_scale x2, rcp, 4 fma (Newton on the
// reciprocal), mul, fma (residual), div_fmas, div_fixup  — 11 instructions, one of them quarter-rate — and shares nothing
// between divisions by the same value (div_scale ties the scaled denominator to each numerator).
// RecipCore(d) runs the reciprocal part once; quot(n) is then mul + fma + fma + div_fixup: the same operations, in the
// same order, as the compiler's expansion with its scaling switched off.  The result is therefore the same bits as `/`
// whenever div_scale would not have scaled: |d| in [2^-400, 2^400) and |n| in [2^-600, 2^600], or n = 0; and div_fixup
// gives the IEEE result for every zero / inf / NaN operand.  Outside (subnormal or huge operands that are not special
// values) the last bit can differ.  tests/test_gpu_divide.py checks all of this bit for bit on the device.
// Who uses what (profiles/r02_shared_recip.md has the measurements):
//   * Recip<double> — unit(v) and `v / s`, three quotients per denominator — is the UNGUARDED routine: +6.1 % on
//     cornell_box.  Its operands are vector components over a length or radius: a component is 0 or within 2^-200 of the
//     length in every vector a bounded scene produces, so these quotients are IEEE division in practice, but the routine
//     does not check it.  These divisions shape directions and normals (shading); nothing that decides WHICH primitive
//     a ray hits or at what t goes through it.
//   * Intersection distances (rectangle planes, sphere roots, boxes) use plain `/` (PlainDiv): always exact.  A checked
//     variant for the 12 plane distances of a Cornell cast (denominators checked once each, wave-wide vote, plain
//     repeat when in doubt) was exact and measured -1.2 %: not used.
//   * SharedRecip<double> adds exponent guards per quotient (plain `/` outside them): exact everywhere, but the guard's
//     VALU -> SALU -> branch chain loses 11 %; kept for -DRT_SHARED_RECIP builds and as the test's reference point.
// ------------------------------------------------------------------------------------------
template <class T> struct Recip {
    T d;
    __device__ explicit Recip(T d_) : d(d_) {}
    __device__ T div(T n) const { return n / d; }
};
template <class T> struct SharedRecip : Recip<T> { __device__ explicit SharedRecip(T d_) : Recip<T>(d_) {} };
#if !defined(RT_HOST_SIM) && defined(__HIP_DEVICE_COMPILE__)
#define RT_HAVE_RECIP_CORE 1
struct RecipCore {
    double d, r;
    __device__ explicit RecipCore(double d_) : d(d_) {
        double r0 = __builtin_amdgcn_rcp(d_);
        double e = __builtin_fma(-d_, r0, 1.0);
        r0 = __builtin_fma(e, r0, r0);
        e = __builtin_fma(-d_, r0, 1.0);
        r = __builtin_fma(e, r0, r0);
    }
    __device__ double quot(double n) const {
        const double q = n * r;
        const double e = __builtin_fma(-d, q, n);
        return __builtin_amdgcn_div_fixup(__builtin_fma(e, r, q), d, n);
    }
    static __device__ inline bool moderate(double x) {  // biased exponent in [1023-400, 1023+400)
        const uint32_t e = ((uint32_t)__double2hiint(x) >> 20) & 0x7ffu;
        return e - 623u < 800u;
    }
};
template <> struct SharedRecip<double> : RecipCore {
    bool ok;
    __device__ explicit SharedRecip(double d_) : RecipCore(d_), ok(moderate(d_)) {}
    __device__ double div(double n) const {
        // wave-uniform choice (both routes give the same bits inside the guards, so a whole wave may take either):
        // one scalar branch instead of an exec-masked region per division
        // (the plain route is marked unlikely so that it is laid out of line: the common route then has no taken branch)
#ifndef RT_RECIP_NOGUARD  // (A/B measurement only: what the guards cost)
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(ok && moderate(n))) != 0, 0)) return n / d;
#endif
        return quot(n);
    }
};
#if defined(RT_SHARED_RECIP)
template <> struct Recip<double> : SharedRecip<double> { __device__ explicit Recip(double d_) : SharedRecip<double>(d_) {} };
#elif !defined(RT_PLAIN_UNIT)
template <> struct Recip<double> : RecipCore {
    __device__ explicit Recip(double d_) : RecipCore(d_) {}
    __device__ double div(double n) const { return quot(n); }
};
#endif
#endif
// A division whose denominator nothing else shares (same interface, plain `/`)
template <class T> struct PlainDiv {
    T d;
    __device__ explicit PlainDiv(T d_) : d(d_) {}
    __device__ T div(T n) const { return n / d; }
};

template <class T> struct V3 {
    T x, y, z;
    __device__ V3() {}
    __device__ V3(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {}
    __device__ explicit V3(const T* p) : x(p[0]), y(p[1]), z(p[2]) {}
};
// src/vec3.rs:56-206 (operator order kept; `*` between vectors is the dot product there)
template <class T> __device__ inline V3<T> operator+(V3<T> a, V3<T> b) { return V3<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class T> __device__ inline V3<T> operator-(V3<T> a, V3<T> b) { return V3<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <class T> __device__ inline V3<T> operator-(V3<T> a) { return V3<T>(-a.x, -a.y, -a.z); }
template <class T> __device__ inline V3<T> operator*(V3<T> a, T b) { return V3<T>(a.x * b, a.y * b, a.z * b); }
template <class T> __device__ inline V3<T> operator/(V3<T> a, T b) { const Recip<T> r(b); return V3<T>(r.div(a.x), r.div(a.y), r.div(a.z)); }
template <class T> __device__ inline T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T> __device__ inline T sqlen(V3<T> a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
template <class T> __device__ inline T length(V3<T> a) { return sqrt_r(a.x * a.x + a.y * a.y + a.z * a.z); }
template <class T> __device__ inline V3<T> elemul(V3<T> a, V3<T> b) { return V3<T>(a.x * b.x, a.y * b.y, a.z * b.z); }
template <class T> __device__ inline V3<T> cross(V3<T> a, V3<T> b) {
    return V3<T>(a.y * b.z - b.y * a.z, b.x * a.z - a.x * b.z, a.x * b.y - b.x * a.y);
}
// src/vec3.rs:30-36; a zero vector yields NaN (the reference panics) -> the sample is retried
template <class T> __device__ inline V3<T> unit(V3<T> a) { const Recip<T> r(length(a)); return V3<T>(r.div(a.x), r.div(a.y), r.div(a.z)); }
template <class T> __device__ inline V3<T> sel(bool c, V3<T> a, V3<T> b) { return V3<T>(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }

template <class T> struct Ray { V3<T> ori, dir; T time; };
template <class T> __device__ inline V3<T> at(const Ray<T>& r, T t) { return r.ori + r.dir * t; }  // src/ray.rs:12-14

template <class T> struct Consts;
template <> struct Consts<double> { static __device__ constexpr double pi() { return 3.14159265358979323846; } };
template <> struct Consts<float> { static __device__ constexpr float pi() { return 3.14159265358979323846f; } };

// ------------------------------------------------------------------------------------------
// RNG: xoshiro128++ keyed by (seed, pixel, sample) through splitmix64.  Same specification as the
// oracle's rng.hpp (the reference's generators are entropy-seeded: only distributions matter).
// ------------------------------------------------------------------------------------------
struct Rng {
    uint32_t s0, s1, s2, s3;
    static __device__ inline uint64_t splitmix64(uint64_t& z) {
        z += 0x9E3779B97F4A7C15ull;
        uint64_t r = z;
        r = (r ^ (r >> 30)) * 0xBF58476D1CE4E5B9ull;
        r = (r ^ (r >> 27)) * 0x94D049BB133111EBull;
        return r ^ (r >> 31);
    }
    __device__ inline void init(uint64_t seed, uint32_t pixel, uint32_t sample) {
        uint64_t key = ((uint64_t)pixel << 32) | sample;
        uint64_t z = seed + key * 0x9E3779B97F4A7C15ull;
        uint64_t a = splitmix64(z);
        uint64_t b = splitmix64(z);
        s0 = (uint32_t)a; s1 = (uint32_t)(a >> 32); s2 = (uint32_t)b; s3 = (uint32_t)(b >> 32);
        if ((s0 | s1 | s2 | s3) == 0) s0 = 1;
    }
    static __device__ inline uint32_t rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
    __device__ inline uint32_t next_u32() {
        uint32_t result = rotl(s0 + s3, 7) + s0;
        uint32_t t = s1 << 9;
        s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= t;
        s3 = rotl(s3, 11);
        return result;
    }
    __device__ inline uint64_t next_u64() { uint64_t hi = next_u32(); uint64_t lo = next_u32(); return (hi << 32) | lo; }
    // One "raw draw" is what gen<T>() / gen_range(T,T) consume: a u64 in f64 mode, a u32 in f32 mode.
    __device__ inline void raw(uint64_t& r) { r = next_u64(); }
    __device__ inline void raw(uint32_t& r) { r = next_u32(); }
    // rand 0.7.3 gen::<f64>(): 53 high bits * 2^-53;  f32: 24 high bits * 2^-24
    static __device__ inline double unit_of(uint64_t r, double) { return (double)(r >> 11) * (1.0 / 9007199254740992.0); }
    static __device__ inline float unit_of(uint32_t r, float) { return (float)(r >> 8) * (1.0f / 16777216.0f); }
    // rand 0.7.3 gen_range(lo,hi) for floats: 52 (23) fraction bits scaled into [lo,hi)
    static __device__ inline double range_of(uint64_t r, double lo, double hi) { return lo + (hi - lo) * ((double)(r >> 12) * (1.0 / 4503599627370496.0)); }
    static __device__ inline float range_of(uint32_t r, float lo, float hi) { return lo + (hi - lo) * ((float)(r >> 9) * (1.0f / 8388608.0f)); }
    __device__ inline void gen(double& out) { out = unit_of(next_u64(), 0.0); }
    __device__ inline void gen(float& out) { out = unit_of(next_u32(), 0.0f); }
    template <class T> __device__ inline T gen() { T v; gen(v); return v; }
    // gen_range redraws when the scaled value rounds up to hi
    __device__ inline double gen_range(double lo, double hi) {
        for (;;) { double v = range_of(next_u64(), lo, hi); if (v < hi) return v; }
    }
    __device__ inline float gen_range(float lo, float hi) {
        for (;;) { float v = range_of(next_u32(), lo, hi); if (v < hi) return v; }
    }
    // rand 0.7.3 gen_range(0,n) for integers: widening multiply with rejection zone
    __device__ inline uint32_t gen_index(uint32_t n) {
        uint64_t m = (uint64_t)next_u32() * n;
        uint32_t l = (uint32_t)m;
        if (l < n) {
            uint32_t thresh = (0u - n) % n;
            while (l < thresh) { m = (uint64_t)next_u32() * n; l = (uint32_t)m; }
        }
        return (uint32_t)(m >> 32);
    }
};
template <class T> struct RawOf;
template <> struct RawOf<double> { typedef uint64_t type; };
template <> struct RawOf<float> { typedef uint32_t type; };

// src/vec3.rs:208-216
template <class T> __device__ inline V3<T> random_in_unit_sphere(Rng& rng) {
    for (;;) {
        T a = rng.gen<T>(); T b = rng.gen<T>(); T c = rng.gen<T>();
        V3<T> t = V3<T>(a, b, c) * T(2) - V3<T>(T(1), T(1), T(1));
        if (sqlen(t) <= T(1)) return t;
    }
}
// src/vec3.rs:223-230
template <class T> __device__ inline V3<T> random_in_unit_disk(Rng& rng) {
    for (;;) {
        T a = rng.gen<T>(); T b = rng.gen<T>();
        V3<T> t = V3<T>(a, b, T(0)) * T(2) - V3<T>(T(1), T(1), T(0));
        if (sqlen(t) <= T(1)) return t;
    }
}
// src/vec3.rs:249-257
template <class T> __device__ inline V3<T> reflect(V3<T> v, V3<T> n) { return v - n * T(2) * dot(v, n); }
template <class T> __device__ inline V3<T> refract(V3<T> v, V3<T> n, T etai_over_etat) {
    T cos_theta = rmin(dot(-v, n), T(1));
    V3<T> r_out_perp = (v + n * cos_theta) * etai_over_etat;
    V3<T> r_out_parallel = (-n) * sqrt_r(rabs(T(1) - sqlen(r_out_perp)));
    return r_out_perp + r_out_parallel;
}

// src/pdf.rs:11-24: tangents of the basis whose third axis is the UNIT vector w
template <class T> __device__ inline void onb_tangents(V3<T> w, V3<T>& u, V3<T>& v) {
    V3<T> a = rabs(w.x) > T(0.9) ? V3<T>(T(0), T(1), T(0)) : V3<T>(T(1), T(0), T(0));
    v = unit(cross(w, a));
    u = unit(cross(w, v));
}

// ------------------------------------------------------------------------------------------
// camera: src/camera.rs:57-65
// ------------------------------------------------------------------------------------------
template <class T> __device__ inline Ray<T> camera_ray(const DCamera<T>& c, T s, T t, V3<T> rd_unit, T time) {
    V3<T> rd = rd_unit * c.lens_radius;
    V3<T> offset = V3<T>(c.u) * rd.x + V3<T>(c.v) * rd.y;
    Ray<T> r;
    r.ori = V3<T>(c.origin) + offset;
    r.dir = V3<T>(c.llc) + V3<T>(c.horizontal) * s + V3<T>(c.vertical) * t - V3<T>(c.origin) - offset;
    r.time = time;
    return r;
}
template <class T> __device__ inline Ray<T> get_ray(const DCamera<T>& c, T s, T t, Rng& rng) {
    V3<T> rd = random_in_unit_disk<T>(rng);   // drawn even when lens_radius == 0
    T time = rng.gen_range(c.t1, c.t2);
    return camera_ray(c, s, t, rd, time);
}

// ------------------------------------------------------------------------------------------
// primitive intersection (object-space ray).  Only t is produced during the closest-hit search;
// p / normal / uv of the WINNING hit are computed afterwards by finish_hit with the reference's
// formulas (the reference computes them for every candidate; values that are discarded have no
// side effects).
// ------------------------------------------------------------------------------------------
// src/objects.rs:227-244, 281-298, 335-352.  axis 0: XY (k on z), 1: XZ (k on y), 2: YZ (k on x).
// Select form: the in-plane point is computed for every lane; the comparisons are the reference's.
// The reciprocals of a ray's direction components, shared by every rectangle test of that ray in one space
// (world, or an instance's object space): see Recip.
#if defined(RT_SHARED_RECIP)
template <class T> struct RayInv {
    Recip<T> x, y, z;
    __device__ explicit RayInv(V3<T> d) : x(d.x), y(d.y), z(d.z) {}
    __device__ Recip<T> on(int axis, V3<T>) const { return axis == 0 ? z : (axis == 1 ? y : x); }  // XY -> z, XZ -> y, YZ -> x
};
#else
template <class T> struct RayInv {  // plain divisions: nothing to carry, the divider is made from the direction at the test
    __device__ explicit RayInv(V3<T>) {}
    __device__ PlainDiv<T> on(int axis, V3<T> d) const { return PlainDiv<T>(axis == 0 ? d.z : (axis == 1 ? d.y : d.x)); }
};
#endif
template <class T, class Div>
__device__ inline bool rect_hit(int axis, T a1, T a2, T b1, T b2, T k, V3<T> o, V3<T> d, const Div& inv_dk, T t_min, T t_max, T& t_out) {
    const T ok = axis == 0 ? o.z : (axis == 1 ? o.y : o.x);
    const T oa = axis == 2 ? o.y : o.x, da = axis == 2 ? d.y : d.x;
    const T ob = axis == 0 ? o.y : o.z, db = axis == 0 ? d.y : d.z;
    const T t = inv_dk.div(k - ok);
    const T a = oa + t * da;
    const T b = ob + t * db;
    const bool hit = (t >= t_min) & (t <= t_max) & (a >= a1) & (a <= a2) & (b >= b1) & (b <= b2);  // inclusive
    t_out = hit ? t : t_out;
    return hit;
}
// `tie` (prim_hit, sphere_hit): set when a candidate distance EQUALS t_max exactly.  Equal distances are where the
// outcome of a closest-hit search depends on the order of the tests (strict for spheres: the earlier one keeps the hit;
// inclusive for rectangles: the later one takes it); walk_fast, which visits leaves in another order than
// the reference, hands such rays to the reference-order walk.  Over-reporting is harmless.
// src/objects.rs:87-119 / 160-193 (strict comparisons).  Both roots are evaluated; the first in
// range wins, as in the reference.
template <class T>
__device__ inline bool sphere_hit(V3<T> center, T radius, V3<T> o, V3<T> d, T t_min, T t_max, T& t_out, bool& tie) {
    V3<T> oc = o - center;
    T a = dot(d, d);
    T half_b = dot(oc, d);
    T c = dot(oc, oc) - radius * radius;
    T discriminant = half_b * half_b - a * c;
    if (discriminant > T(0)) {
        T sq = sqrt_r(discriminant);
#if defined(RT_SHARED_RECIP)
        const Recip<T> inv_a(a);
#else
        const PlainDiv<T> inv_a(a);  // intersection distances are true divisions (see Recip)
#endif
        T r1 = inv_a.div(-half_b - sq);
        T r2 = inv_a.div(-half_b + sq);
        bool h1 = (r1 > t_min) & (r1 < t_max);
        bool h2 = (r2 > t_min) & (r2 < t_max);
        tie = (r1 == t_max) | (r2 == t_max);
        if (h1 | h2) { t_out = h1 ? r1 : r2; return true; }
    }
    return false;
}
template <class T>
__device__ inline bool sphere_hit(V3<T> center, T radius, V3<T> o, V3<T> d, T t_min, T t_max, T& t_out) {
    bool tie = false;
    return sphere_hit(center, radius, o, d, t_min, t_max, t_out, tie);
}
// src/objects.rs:154-158
template <class T> __device__ inline V3<T> moving_center(const DPrim<T>& p, T time) {
    V3<T> c1(p.d), c2(p.d + 3);
    return c1 + (c2 - c1) * ((time - p.d[6]) / (p.d[7] - p.d[6]));
}
// cuboid side i as a rect: src/objects.rs:399-452
template <class T> __device__ inline void cuboid_side(const DPrim<T>& p, int i, int& axis, T& a1, T& a2, T& b1, T& b2, T& k, T& face) {
    const T* mn = p.d; const T* mx = p.d + 3;
    axis = i >> 1;
    face = (i & 1) ? T(1) : T(-1);
    a1 = axis == 2 ? mn[1] : mn[0]; a2 = axis == 2 ? mx[1] : mx[0];
    b1 = axis == 0 ? mn[1] : mn[2]; b2 = axis == 0 ? mx[1] : mx[2];
    const int kc = 2 - axis;  // XY -> z, XZ -> y, YZ -> x
    k = (i & 1) ? mx[kc] : mn[kc];
}

// Closest hit of ONE primitive in (t_min, t_max); side = cuboid face.
template <class T, uint32_t F>
__device__ inline bool prim_hit(const DPrim<T>& p, V3<T> o, V3<T> d, const RayInv<T>& inv, T time, T t_min, T t_max, T& t_out, int& side) {
    side = 0;
    const uint32_t kind = p.kind;
    if ((has<F>(F_SPHERE) || has<F>(F_MOVING)) && kind <= 1) {  // one sphere test for both kinds: mixed leaves stay converged
        V3<T> center(p.d);
        T radius = p.d[3];
        if (has<F>(F_MOVING) && (!has<F>(F_SPHERE) || kind == 1)) { center = moving_center(p, time); radius = p.d[8]; }
        return sphere_hit(center, radius, o, d, t_min, t_max, t_out);
    }
    if (has<F>(F_RECT) && kind >= 2 && kind <= 4) {
        // one copy per axis behind a wave-uniform branch: with a run-time axis the six component picks are 12 selects
        if (kind == 2) return rect_hit<T>(0, p.d[0], p.d[1], p.d[2], p.d[3], p.d[4], o, d, inv.on(0, d), t_min, t_max, t_out);
        if (kind == 3) return rect_hit<T>(1, p.d[0], p.d[1], p.d[2], p.d[3], p.d[4], o, d, inv.on(1, d), t_min, t_max, t_out);
        return rect_hit<T>(2, p.d[0], p.d[1], p.d[2], p.d[3], p.d[4], o, d, inv.on(2, d), t_min, t_max, t_out);
    }
    if (has<F>(F_CUBOID) && kind == 5) {  // six sides in tuple order, shrinking closest (src/objects.rs:458-485)
        bool any = false;
        T closest = t_max;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            int axis; T a1, a2, b1, b2, k, face;
            cuboid_side(p, i, axis, a1, a2, b1, b2, k, face);
            T t = closest;
            const bool h = rect_hit<T>(axis, a1, a2, b1, b2, k, o, d, inv.on(axis, d), t_min, closest, t);
            any |= h; closest = t; side = h ? i : side;
        }
        t_out = closest;
        return any;
    }
    return false;
}
// The same test for walk_fast, which also needs to know about equal distances (`tie`, see rect_hit's note).  Kept as a
// separate function: folding the flag into prim_hit changed the code generated for the kernels that never use it
// (cornell_box -1.5 %).
template <class T, uint32_t F>
__device__ inline bool prim_hit_tie(const DPrim<T>& p, V3<T> o, V3<T> d, const RayInv<T>& inv, T time, T t_min, T t_max, T& t_out, int& side, bool& tie) {
    tie = false;
    const uint32_t kind = p.kind;
    if ((has<F>(F_SPHERE) || has<F>(F_MOVING)) && kind <= 1) {
        side = 0;
        V3<T> center(p.d);
        T radius = p.d[3];
        if (has<F>(F_MOVING) && (!has<F>(F_SPHERE) || kind == 1)) { center = moving_center(p, time); radius = p.d[8]; }
        return sphere_hit(center, radius, o, d, t_min, t_max, t_out, tie);
    }
    const bool h = prim_hit<T, F>(p, o, d, inv, time, t_min, t_max, t_out, side);
    tie = h & (t_out == t_max);  // inclusive tests: a hit AT t_max is accepted
    return h;
}

// src/bvh.rs:26-58 — loose slab test, true divisions, minNum/maxNum; all three axes evaluated
template <class T> __device__ inline bool aabb_hit(const T* mn, const T* mx, V3<T> o, V3<T> d, T t_min, T t_max) {
    bool ok = true;
    {
        T t1 = (mn[0] - o.x) / d.x, t2 = (mx[0] - o.x) / d.x;
        const bool sw = t1 > t2; const T lo = sw ? t2 : t1, hi = sw ? t1 : t2;
        ok &= !(rmin(t_max, hi) <= rmax(t_min, lo));
    }
    {
        T t1 = (mn[1] - o.y) / d.y, t2 = (mx[1] - o.y) / d.y;
        const bool sw = t1 > t2; const T lo = sw ? t2 : t1, hi = sw ? t1 : t2;
        ok &= !(rmin(t_max, hi) <= rmax(t_min, lo));
    }
    {
        T t1 = (mn[2] - o.z) / d.z, t2 = (mx[2] - o.z) / d.z;
        const bool sw = t1 > t2; const T lo = sw ? t2 : t1, hi = sw ? t1 : t2;
        ok &= !(rmin(t_max, hi) <= rmax(t_min, lo));
    }
    return ok;
}

// Does the reference's loose test of this box certainly PASS for every t_max it can be called with on the way to a
// leaf whose accepted hit lies at t_win?  (walk_fast: the box is the leaf's parent node in the reference's tree.)
// The reference passes axis a iff min(t_max, hi_a) > max(t_min, lo_a), with the slab ends computed as in aabb_hit, and
// its t_max is the closest distance found so far: >= t_win, and > t_win when tmax_above (no equal distance anywhere).
// Sufficient, per axis: hi_a > max(t_min, lo_a), t_win > t_min, and lo_a < t_win — or lo_a == t_win with tmax_above
// (a hit ON the box face, the normal case for a cuboid whose face lies in its parent's box).  A NaN slab end (0/0) is
// left to the exact emulation.
template <class T> __device__ inline bool aabb_vouch(const T* mn, const T* mx, V3<T> o, V3<T> d, T t_min, T t_win, bool tmax_above) {
    bool ok = t_win > t_min;
    {
        T t1 = (mn[0] - o.x) / d.x, t2 = (mx[0] - o.x) / d.x;
        const bool sw = t1 > t2; const T lo = sw ? t2 : t1, hi = sw ? t1 : t2;
        const T m = lo > t_min ? lo : t_min;
        ok &= (lo == lo) & (hi == hi) & (hi > m) & ((lo < t_win) | ((lo == t_win) & tmax_above));
    }
    {
        T t1 = (mn[1] - o.y) / d.y, t2 = (mx[1] - o.y) / d.y;
        const bool sw = t1 > t2; const T lo = sw ? t2 : t1, hi = sw ? t1 : t2;
        const T m = lo > t_min ? lo : t_min;
        ok &= (lo == lo) & (hi == hi) & (hi > m) & ((lo < t_win) | ((lo == t_win) & tmax_above));
    }
    {
        T t1 = (mn[2] - o.z) / d.z, t2 = (mx[2] - o.z) / d.z;
        const bool sw = t1 > t2; const T lo = sw ? t2 : t1, hi = sw ? t1 : t2;
        const T m = lo > t_min ? lo : t_min;
        ok &= (lo == lo) & (hi == hi) & (hi > m) & ((lo < t_win) | ((lo == t_win) & tmax_above));
    }
    return ok;
}

// Cheap classification of a box against the ray, in f32 and WITHOUT divisions: 1 = the reference's
// loose test certainly passes, 0 = nothing inside this box can be hit in (t_min, t_max) (cull),
// 2 = too close to call -> the caller evaluates aabb_hit exactly (in T, true divisions).
//   Inputs: the box rounded outward to f32, the ray origin and reciprocal direction rounded to f32.
//   a = (bound - o) * (1/d) then differs from the reference's (bound - o) / d by less than
//   e_i = 2^-18 (|o_i| + B) |1/d_i|  per axis (B = largest |coordinate| of the tree; the true error
//   is < 5e-7 of that scale, e_i is 7x larger).  Each slab interval [lo_i, hi_i] is therefore known
//   to within e_i, and (t_min, t_max) to within their f32 rounding:
//   inner = min(t_max-, hi_i - e_i) - max(t_min+, lo_i + e_i) > 0: every axis overlaps (t_min,t_max)
//           by more than all rounding -> the per-axis loose test on the exact box passes, exactly
//           as in the reference -> descend;
//   outer = min(t_max+, hi_i + e_i) - max(t_min-, lo_i - e_i) < 0: a hit at t in (t_min,t_max)
//           inside this box would lie inside every slab interval up to the hit point's rounding
//           error (~1e-13 in f64), which e_i exceeds by orders of magnitude -> skipping cannot
//           change the closest hit, whatever the loose test says;
//   otherwise (grazing rays, axis-parallel rays with 1/d = inf, NaN, f32 overflow): the exact test.
// The reference's loose test lets through ~27 % of a 480-node tree per ray; the cull visits the few
// nodes the ray actually crosses and still returns the identical hit (ids and t bit-exact).
// R = float on the f32 box: first level (cheap).  R = double on the EXACT f64 box (f64 mode only): second level
// for the lanes the first could not decide; its only error is f64 rounding (e_i = 2^-46 (|o_i| + B)|1/d_i|), so it
// resolves e.g. rays that START on a box face (t_min = 0.001 is below the f32 resolution at coordinates ~1000)
// without the 6-division exact test, which is left with genuine ties (a hit lying on a plane two boxes share).
template <class R> struct RayR { R ox, oy, oz, ix, iy, iz, ex, ey, ez, tmin_lo, tmin_hi; };
__device__ inline float rminf(float a, float b) { return fminf(a, b); }
__device__ inline float rmaxf(float a, float b) { return fmaxf(a, b); }
__device__ inline double rminf(double a, double b) { return ::fmin(a, b); }
__device__ inline double rmaxf(double a, double b) { return ::fmax(a, b); }
template <class R, class BoxT>
__device__ inline int aabb_classify(const BoxT* mn, const BoxT* mx, const RayR<R>& r, R tmax_lo, R tmax_hi) {
    const R x1 = ((R)mn[0] - r.ox) * r.ix, x2 = ((R)mx[0] - r.ox) * r.ix;
    const R y1 = ((R)mn[1] - r.oy) * r.iy, y2 = ((R)mx[1] - r.oy) * r.iy;
    const R z1 = ((R)mn[2] - r.oz) * r.iz, z2 = ((R)mx[2] - r.oz) * r.iz;
    const R lx = x1 < x2 ? x1 : x2, hx = x1 < x2 ? x2 : x1;
    const R ly = y1 < y2 ? y1 : y2, hy = y1 < y2 ? y2 : y1;
    const R lz = z1 < z2 ? z1 : z2, hz = z1 < z2 ? z2 : z1;
    const R in_lo = rmaxf(rmaxf(lx + r.ex, ly + r.ey), rmaxf(lz + r.ez, r.tmin_hi));
    const R in_hi = rminf(rminf(hx - r.ex, hy - r.ey), rminf(hz - r.ez, tmax_lo));
    const R out_lo = rmaxf(rmaxf(lx - r.ex, ly - r.ey), rmaxf(lz - r.ez, r.tmin_lo));
    const R out_hi = rminf(rminf(hx + r.ex, hy + r.ey), rminf(hz + r.ez, tmax_hi));
    // NaN anywhere (0 * inf on an axis-parallel ray, ...) -> undecided
    const bool nan = (lx != lx) | (ly != ly) | (lz != lz) | (hx != hx) | (hy != hy) | (hz != hz) | (r.ex != r.ex) | (r.ey != r.ey) | (r.ez != r.ez);
    if (nan) return 2;
    return in_hi - in_lo > R(0) ? 1 : (out_hi - out_lo < R(0) ? 0 : 2);
}
// T -> f32, saturating at +-FLT_MAX: the search ranges start at +-f64::MAX (ray_tracing.rs:22, objects.rs:530), which is
// outside f32 (the plain cast is undefined in C++ and gives inf, whose widened ends would be inf - inf = NaN)
__device__ inline float to_f32_sat(double t) { return (float)(t > 3.4028234663852886e38 ? 3.4028234663852886e38 : (t < -3.4028234663852886e38 ? -3.4028234663852886e38 : t)); }
__device__ inline float to_f32_sat(float t) { return t; }
// t widened by one part in 2^20 either way (covers its rounding to f32; harmless in f64).  At +-FLT_MAX the outer end
// overflows to +-inf, which is the right bound.
template <class R> __device__ inline void widen(R t, R& lo, R& hi) {
    const R w = (t < R(0) ? -t : t) * R(9.5367431640625e-07);
    lo = t - w; hi = t + w;
}
template <class R, class T>
__device__ inline void ray_consts(RayR<R>& r, V3<T> o, V3<T> d, T t_min, R B, R rel) {
    r.ox = (R)o.x; r.oy = (R)o.y; r.oz = (R)o.z;
    r.ix = R(1) / (R)d.x; r.iy = R(1) / (R)d.y; r.iz = R(1) / (R)d.z;
    const R tiny = R(1e-30);
    r.ex = ((r.ox < 0 ? -r.ox : r.ox) + B + tiny) * rel * (r.ix < 0 ? -r.ix : r.ix);
    r.ey = ((r.oy < 0 ? -r.oy : r.oy) + B + tiny) * rel * (r.iy < 0 ? -r.iy : r.iy);
    r.ez = ((r.oz < 0 ? -r.oz : r.oz) + B + tiny) * rel * (r.iz < 0 ? -r.iz : r.iz);
    widen<R>((R)t_min, r.tmin_lo, r.tmin_hi);
}

// First level in its cheapest form.  Per axis the four values (bound - o)/d -+ e for the two bounds come from one
// multiply-add each, a = bound * (1/d) + (-o * (1/d) -+ e): the margin is folded into the addend, and with it the
// four interval ends lo-e, lo+e, hi-e, hi+e are plain min / max of those.  The rounding of this form (product,
// addend, final sum: each < 2^-24 (|o| + B)|1/d|) stays inside the same e = 2^-18 (|o| + B)|1/d| as before.
// A ray whose constants are not all finite (axis-parallel: 1/d = inf; overflow) is marked `bad`: all its nodes are
// undecided and go to the second level / the exact test.  With finite constants and finite boxes no NaN can arise
// per node (an overflowing product gives +-inf, never inf - inf), so there is no per-node NaN test.
struct RayF { float ix, iy, iz, kpx, kpy, kpz, kmx, kmy, kmz, tmin_lo, tmin_hi; bool bad; };
template <class T> __device__ inline void ray_consts_f(RayF& r, V3<T> o, V3<T> d, T t_min, float B) {
    const float rel = 3.814697265625e-06f;  // 2^-18
    const float ox = (float)o.x, oy = (float)o.y, oz = (float)o.z;
    r.ix = 1.0f / (float)d.x; r.iy = 1.0f / (float)d.y; r.iz = 1.0f / (float)d.z;
    const float ex = (fabsf(ox) + B + 1e-30f) * rel * fabsf(r.ix);
    const float ey = (fabsf(oy) + B + 1e-30f) * rel * fabsf(r.iy);
    const float ez = (fabsf(oz) + B + 1e-30f) * rel * fabsf(r.iz);
    const float cx = -ox * r.ix, cy = -oy * r.iy, cz = -oz * r.iz;
    r.kpx = cx + ex; r.kmx = cx - ex;
    r.kpy = cy + ey; r.kmy = cy - ey;
    r.kpz = cz + ez; r.kmz = cz - ez;
    const float chk = (r.ix - r.ix) + (r.iy - r.iy) + (r.iz - r.iz) + (r.kpx - r.kpx) + (r.kpy - r.kpy) + (r.kpz - r.kpz) +
                      (r.kmx - r.kmx) + (r.kmy - r.kmy) + (r.kmz - r.kmz);  // 0 iff all nine are finite
    r.bad = !(chk == 0.0f);
    widen<float>(to_f32_sat(t_min), r.tmin_lo, r.tmin_hi);
}
__device__ inline int aabb_classify_f(const float* mn, const float* mx, const RayF& r, float tmax_lo, float tmax_hi) {
    const float x1p = fmaf(mn[0], r.ix, r.kpx), x2p = fmaf(mx[0], r.ix, r.kpx), x1m = fmaf(mn[0], r.ix, r.kmx), x2m = fmaf(mx[0], r.ix, r.kmx);
    const float y1p = fmaf(mn[1], r.iy, r.kpy), y2p = fmaf(mx[1], r.iy, r.kpy), y1m = fmaf(mn[1], r.iy, r.kmy), y2m = fmaf(mx[1], r.iy, r.kmy);
    const float z1p = fmaf(mn[2], r.iz, r.kpz), z2p = fmaf(mx[2], r.iz, r.kpz), z1m = fmaf(mn[2], r.iz, r.kmz), z2m = fmaf(mx[2], r.iz, r.kmz);
    const float in_lo = fmaxf(fmaxf(fminf(x1p, x2p), fminf(y1p, y2p)), fmaxf(fminf(z1p, z2p), r.tmin_hi));   // lo + e
    const float in_hi = fminf(fminf(fmaxf(x1m, x2m), fmaxf(y1m, y2m)), fminf(fmaxf(z1m, z2m), tmax_lo));     // hi - e
    const float out_lo = fmaxf(fmaxf(fminf(x1m, x2m), fminf(y1m, y2m)), fmaxf(fminf(z1m, z2m), r.tmin_lo));  // lo - e
    const float out_hi = fminf(fminf(fmaxf(x1p, x2p), fmaxf(y1p, y2p)), fminf(fmaxf(z1p, z2p), tmax_hi));    // hi + e
    if (r.bad) return 2;
    return in_hi - in_lo > 0.0f ? 1 : (out_hi - out_lo < 0.0f ? 0 : 2);
}

// true if the predicate holds in any active lane of the wave (the CPU build of these templates runs one "lane" at a time)
__device__ inline bool wave_any(bool v) {
#if defined(RT_HOST_SIM) || !defined(__HIP_DEVICE_COMPILE__)
    return v;
#else
    return __builtin_amdgcn_ballot_w64(v) != 0ull;
#endif
}
__device__ inline float canonical(float x) {
#if defined(RT_HOST_SIM) || !defined(__HIP_DEVICE_COMPILE__)
    return x;
#else
    return __builtin_canonicalizef(x);
#endif
}
// Entry i of the table staged in LDS.  The generic pointer is narrowed to the LDS address space by hand
// (its low 32 bits are the LDS offset): the optimiser cannot infer it through DScene, and a flat load
// would wait on both memory counters.
__device__ inline DBvhLite lds_entry(const DBvhLite* table, int i) {
#if defined(RT_HOST_SIM) || !defined(__HIP_DEVICE_COMPILE__)
    return table[i];
#else
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(3))) u4* lds_ptr;
    lds_ptr p = (lds_ptr)(uintptr_t)(uint32_t)(uintptr_t)table + 2 * i;
    const u4 a = p[0], b = p[1];
    DBvhLite e;
    e.mn[0] = __uint_as_float(a.x); e.mn[1] = __uint_as_float(a.y); e.mn[2] = __uint_as_float(a.z); e.mx[0] = __uint_as_float(a.w);
    e.mx[1] = __uint_as_float(b.x); e.mx[2] = __uint_as_float(b.y); e.skip = (int32_t)b.z; e.prim = (int32_t)b.w;
    return e;
#endif
}

// A threaded-BVH walk in progress (src/bvh.rs:114-129: box test, left subtree, then right subtree with
// t_max = best so far; strict/inclusive tie rules of the leaf primitives decide equal-t replacement; the
// visiting ORDER is the reference's, so ties resolve identically).
template <class T> struct Walk {
    int i, end;        // next entry, one past the instance's subtree
    T closest, t_min;
    bool any, leaf_cull, node_cull;
    bool cross_tie;    // a candidate distance equal to the distance the search STARTED from (a hit of another world instance)
    int prim, side;
    float B;           // largest |coordinate| of the tree's root box
    RayF r;            // per-ray constants of the first-level box classification
};
// fast: the device's own tree (walk_fast); else the reference's tree (walk_run)
template <class T, uint32_t F>
__device__ inline void walk_begin(const DScene<T>& sc, uint32_t geom_index, V3<T> o, V3<T> d, T time, T t_min, T t_max, Walk<T>& w, bool fast) {
    w.any = false;
    w.cross_tie = false;
    w.closest = t_max;
    w.t_min = t_min;
    w.prim = -1;
    w.side = 0;
    w.i = fast ? sc.fast_root_entry[geom_index] : sc.bvh_root_entry[geom_index];
    float B;
    {
        const DBvhLite& root = fast ? sc.fast_lite[w.i] : sc.bvh_lite[w.i];
        w.end = root.skip;
        B = fabsf(root.mn[0]);
        B = fmaxf(B, fabsf(root.mn[1])); B = fmaxf(B, fabsf(root.mn[2]));
        B = fmaxf(B, fabsf(root.mx[0])); B = fmaxf(B, fabsf(root.mx[1])); B = fmaxf(B, fabsf(root.mx[2]));
    }
    ray_consts_f<T>(w.r, o, d, t_min, B);  // 2^-18 (|o| + B)|1/d|: f32 rounding of o, box, products
    w.B = B;
    // leaves carry their primitive's own box: most leaf visits are misses (measured 94 % in random_scene) and
    // an exact sphere / cuboid test costs 70-270 f64 instructions.  Moving spheres' boxes cover sc.time_lo..time_hi.
#ifdef RT_DBG_NO_LEAF_CULL
    w.leaf_cull = false;
#else
    w.leaf_cull = time >= sc.time_lo && time <= sc.time_hi;
#endif
    // "Nothing inside this box can be hit" (class 0 of the node classification) lets walk_run skip a node the reference's loose
    // test would enter — harmless only while the primitives below the node ARE inside its box.  The reference builds its boxes
    // for the times it is given (bvh.rs:85-111, called with 0.0, 1.0); a moving sphere leaves them at ray times outside.
    // Then every node decision is the reference's own test.
    w.node_cull = time >= sc.node_cull_lo && time <= sc.node_cull_hi;
}
// Continue the walk.  "while-while" form with three kinds of step, each run for all lanes that want it before the
// next kind runs, so that box tests, the rare undecided nodes and the primitive tests are not serialised against
// each other inside a wave:
//   A  first-level classification, until the lane stands on an undecided node, on a leaf to test, or is done
//   B  undecided internal nodes: second level (f64), then the reference's exact test
//   C  leaf primitive tests
// The per-lane sequence of decisions is the sequential one; only the interleaving between lanes differs.
// Node records come from global memory (L2): the LDS copy the render kernels stage is of the device's own trees.
// This is the EXACT EMULATION of the reference's traversal (order, box decisions in the grazing cases); the render
// kernels run it only for the rays walk_fast could not vouch for.
#if defined(RT_NOINLINE_REFWALK) && !defined(RT_HOST_SIM)
#define RT_REFWALK_ATTR __attribute__((noinline))
#else
#define RT_REFWALK_ATTR inline
#endif
template <class T, uint32_t F>
__device__ RT_REFWALK_ATTR void walk_run(const DScene<T>& sc, Walk<T>& w, V3<T> o, V3<T> d, const RayInv<T>& inv, T time) {
    const int n_lds = 0;
    int i = w.i;
    const int end = w.end;
    T closest = w.closest;
#if defined(RT_HOST_SIM) && defined(RT_SIM_COUNTERS)
    const unsigned long long sim_n0 = g_sim_counters[0];
#endif
    for (;;) {
        int prim = -1;
        bool undecided = false;
        float tmax_lo, tmax_hi;
        widen<float>(to_f32_sat(closest), tmax_lo, tmax_hi);
        RT_PCB();
        DBvhLite e;
        e.skip = end;
        while (i != end) {
            if (i < n_lds) e = lds_entry(sc.bvh_lds, i); else e = sc.bvh_lite[i];
            prim = e.prim;
            const int c = aabb_classify_f(e.mn, e.mx, w.r, tmax_lo, tmax_hi);
            RT_SIM_COUNT(0);
            if (prim >= 0) {
                // a leaf the first level cannot rule out is tested (its box only serves to skip certain misses)
                if (c != 0 || !w.leaf_cull) break;
                RT_SIM_COUNT(4);
                i = i + 1;  // certainly missed: next entry
                continue;
            }
            if (c == 2 || (c == 0 && !w.node_cull)) { undecided = true; break; }
            i = c == 1 ? i + 1 : e.skip;
        }
        RT_PCE(2);
        if (i == end) break;
        if (undecided) {
            int c = 2;
            const DBvhEntry<T>& ex = sc.bvh[i];
            if (sizeof(T) == 8 && w.node_cull) {  // second level: f64 ray against the exact box (its ray constants are built here: the step is rare)
                RT_SIM_COUNT(5);
                RayR<T> r2;
                ray_consts<T, T>(r2, o, d, w.t_min, (T)w.B, T(1.4210854715202004e-14));  // 2^-46: f64 rounding of differences and products
                c = aabb_classify<T, T>(ex.mn, ex.mx, r2, closest, closest);
            }
            bool pass = c == 1;
            if (c == 2) {  // rare: the reference's exact decision on the exact box
                RT_SIM_COUNT(1);
                pass = aabb_hit(ex.mn, ex.mx, o, d, w.t_min, closest);
            }
            i = pass ? i + 1 : e.skip;
        } else {
            T t; int side; bool tie;
            RT_SIM_COUNT(2);
            const bool any_before = w.any;
            if (prim_hit_tie<T, F>(sc.prims[prim], o, d, inv, time, w.t_min, closest, t, side, tie)) {
                RT_SIM_COUNT(3);
                w.any = true; closest = t; w.prim = prim; w.side = side;
            }
            if (sizeof(T) == 8 && tie && !any_before) w.cross_tie = true;  // (closest was still the distance the search started from)
            i = i + 1;
        }
        RT_PCE(3);
    }
    w.i = i;
    w.closest = closest;
#if defined(RT_HOST_SIM) && defined(RT_SIM_COUNTERS)
    RT_SIM_WALK(sim_n0);
#endif
}

// The walk the render kernels run first: over the device's OWN tree of the same leaves (surface-area heuristic,
// rt_pack.hpp), with a two-way box decision.  Why the result is still the reference's, bit for bit:
//  * Box decision.  A box is skipped only when nothing inside it can be hit in (t_min, closest) — the outer interval
//    of aabb_classify_f, margins included — and entered otherwise.  Skipping is therefore harmless whatever the tree,
//    and entering more boxes than the reference would only costs time: the leaves decide, with the exact tests.
//    So this walk finds the closest of ALL hits the exact primitive tests accept.
//  * The reference finds the closest of the hits IT reaches, and it can fail to reach a leaf: its loose slab test
//    (bvh.rs:26-58) may reject, by rounding, a box whose primitive the exact test would accept (a ray grazing the box).
//    A hit of this walk is vouched for when the loose test of the leaf's PARENT NODE IN THE REFERENCE'S TREE passes for
//    every t_max the reference can use there (aabb_vouch): every ancestor box contains that box and the slab arithmetic
//    is monotone in the bounds and in t_max, so all of the reference's box tests on the path to that leaf pass and it
//    tests the leaf too (6 divisions, once per walk that hit something).
//  * Order.  With distinct distances the closest accepted hit does not depend on the order of the tests; with EQUAL
//    distances it does, and this walk visits leaves in another order: `tie` of prim_hit.
//  Rays whose hit is not vouched for, or that met a tie, are walked again by walk_run (returns false).  f32 mode has no
//  bit-exactness bar and skips both checks.
template <class T, uint32_t F>
__device__ inline bool walk_fast(const DScene<T>& sc, Walk<T>& w, V3<T> o, V3<T> d, const RayInv<T>& inv, T time) {
    const int n_lds = (int)sc.n_bvh_lds;
    int i = w.i;
    const int end = w.end;
    T closest = w.closest;
    const T t_max0 = w.closest;  // the distance the search starts from (the closest hit of the instances before this one)
    bool doubt = false;
    int ref_parent = 0;
    // per axis the bound the ray meets first: near = min for a positive direction (a bad ray never culls: the pick is irrelevant)
    const bool px = w.r.ix >= 0.0f, py = w.r.iy >= 0.0f, pz = w.r.iz >= 0.0f;
    const float tmin_lo_c = canonical(w.r.tmin_lo);  // (known canonical: fmaxf below needs no per-step quieting instruction for it)
#if defined(RT_HOST_SIM) && defined(RT_SIM_COUNTERS)
    const unsigned long long sim_n0 = g_sim_counters[0];
#endif
    for (;;) {
        int prim = -1;
        float tmax_lo, tmax_hi;
        widen<float>(to_f32_sat(closest), tmax_lo, tmax_hi);
        RT_PCB();
        DBvhLite e;
        e.skip = end;
        while (i != end) {
            if (i < n_lds) e = lds_entry(sc.bvh_lds, i); else e = sc.fast_lite[i];
            prim = e.prim;
            // outer interval only: lo - e on the near bounds, hi + e on the far ones (see aabb_classify_f)
#if defined(RT_HOST_SIM) || !defined(__HIP_DEVICE_COMPILE__) || !defined(RT_PK_FMA)
            const float lox = fmaf(px ? e.mn[0] : e.mx[0], w.r.ix, w.r.kmx), hix = fmaf(px ? e.mx[0] : e.mn[0], w.r.ix, w.r.kpx);
            const float loy = fmaf(py ? e.mn[1] : e.mx[1], w.r.iy, w.r.kmy), hiy = fmaf(py ? e.mx[1] : e.mn[1], w.r.iy, w.r.kpy);
            const float loz = fmaf(pz ? e.mn[2] : e.mx[2], w.r.iz, w.r.kmz), hiz = fmaf(pz ? e.mx[2] : e.mn[2], w.r.iz, w.r.kpz);
#else
            // EXPERIMENT (-DRT_PK_FMA, measured and not adopted): the (near, far) pair of an axis through ONE packed multiply-add
            // (v_pk_fma_f32): 3 instead of 6 of the node step's ~30 vector instructions, but the operand pairs cost registers —
            // spheres + BVH spills 160 -> 208 B, random_scene 2590 -> 2417 Msamples/s
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 tx = __builtin_elementwise_fma(f32x2{px ? e.mn[0] : e.mx[0], px ? e.mx[0] : e.mn[0]}, f32x2{w.r.ix, w.r.ix}, f32x2{w.r.kmx, w.r.kpx});
            const f32x2 ty = __builtin_elementwise_fma(f32x2{py ? e.mn[1] : e.mx[1], py ? e.mx[1] : e.mn[1]}, f32x2{w.r.iy, w.r.iy}, f32x2{w.r.kmy, w.r.kpy});
            const f32x2 tz = __builtin_elementwise_fma(f32x2{pz ? e.mn[2] : e.mx[2], pz ? e.mx[2] : e.mn[2]}, f32x2{w.r.iz, w.r.iz}, f32x2{w.r.kmz, w.r.kpz});
            const float lox = tx.x, hix = tx.y, loy = ty.x, hiy = ty.y, loz = tz.x, hiz = tz.y;
#endif
            const float out_lo = fmaxf(fmaxf(lox, loy), fmaxf(loz, tmin_lo_c));
            const float out_hi = fminf(fminf(hix, hiy), fminf(hiz, tmax_hi));
            const bool miss = !w.r.bad && (out_hi - out_lo < 0.0f);
            RT_SIM_COUNT(0);
#ifdef RT_WALK_NESTED_IFS
            if (prim >= 0) {
                if (!miss || !w.leaf_cull) break;  // a leaf that cannot be ruled out is tested
                RT_SIM_COUNT(4);
                i = i + 1;
                continue;
            }
            i = miss ? e.skip : i + 1;
#else
            // one exit test and one select per step: a leaf that cannot be ruled out is tested; a culled leaf and an entered node go on
            // to the next entry, a skipped node to the end of its subtree
            const bool leaf = prim >= 0;
            if (leaf & !(miss & w.leaf_cull)) break;
#if defined(RT_HOST_SIM) && defined(RT_SIM_COUNTERS)
            if (leaf) RT_SIM_COUNT(4);
#endif
            i = (miss & !leaf) ? e.skip : i + 1;
#endif
        }
        RT_PCE(2);
        if (i == end) break;
        T t; int side; bool tie;
        const bool any_before = w.any;
        RT_SIM_COUNT(2);
        if (prim_hit_tie<T, F>(sc.prims[prim], o, d, inv, time, w.t_min, closest, t, side, tie)) {
            RT_SIM_COUNT(3);
            w.any = true; closest = t; w.prim = prim; w.side = side;
            ref_parent = e.skip;  // a leaf's skip field: its parent node's entry in the reference's tree
        }
        if (sizeof(T) == 8 && tie && any_before) doubt = true;  // equal to a distance found in THIS walk: order decides
        if (sizeof(T) == 8 && tie && !any_before) w.cross_tie = true;  // equal to the distance the search started from (another instance's hit)
        i = i + 1;
        RT_PCE(3);
    }
    w.i = i;
    w.closest = closest;
    if (sizeof(T) == 8 && w.any && !doubt) {
        RT_SIM_COUNT(7);
        RT_PCB();
        const DBvhEntry<T>& pb = sc.bvh[ref_parent];
        if (!aabb_vouch(pb.mn, pb.mx, o, d, w.t_min, closest, closest < t_max0)) doubt = true;
        RT_PCE(10);
    }
#if defined(RT_HOST_SIM) && defined(RT_SIM_COUNTERS)
    RT_SIM_WALK(sim_n0);
#endif
    return !doubt;
}

// The primitive of a one-primitive list entry: the entry's own copy; the flat (u,v) variant reads the primitive table instead
// (measured: with the copy it spills 80 instead of 48 B per lane, earth -1.2 %).
template <class T, uint32_t F> __device__ inline const DPrim<T>& prim_of(const DScene<T>& sc, const DInstance<T>& in) {
    if (has<F>(F_UV) && !has<F>(F_BVH)) return sc.prims[in.geom_index];
    return in.prim;
}
// Geometry of an instance: one primitive, or a complete BVH walk.
template <class T, uint32_t F>
__device__ inline bool geom_hit(const DScene<T>& sc, const DInstance<T>& in, V3<T> o, V3<T> d, const RayInv<T>& inv, T time,
                                T t_min, T t_max, T& t_out, int& prim_out, int& side_out, bool& cross_tie, bool lead = false,
                                T t_max_list = T(0)) {
    const uint32_t geom_kind = in.geom_kind, geom_index = in.geom_index;
    // cross_tie (variants with trees, f64): set when a candidate distance equals t_max, the closest distance of the instances tested
    // before — the one case in which the ORDER of the world scan decides (world_hit)
    if (!has<F>(F_BVH) || geom_kind == 0) {
        prim_out = (int)geom_index;
        RT_PCB();
        bool h;
        if (has<F>(F_BVH) && sizeof(T) == 8) {
            bool tie;
            h = prim_hit_tie<T, F>(prim_of<T, F>(sc, in), o, d, inv, time, t_min, t_max, t_out, side_out, tie);
            cross_tie |= tie;
        } else {
            h = prim_hit<T, F>(prim_of<T, F>(sc, in), o, d, inv, time, t_min, t_max, t_out, side_out);
        }
        RT_PCE(6);
        return h;
    }
    Walk<T> w;
    bool sure = false;
#ifndef RT_NO_FAST_WALK
    if (time >= sc.fast_lo && time <= sc.fast_hi) {  // (the device's trees are built from boxes valid for these ray times)
        RT_PCB();
        walk_begin<T, F>(sc, geom_index, o, d, time, t_min, t_max, w, true);
        RT_PCE(9);
        sure = walk_fast<T, F>(sc, w, o, d, inv, time);
    }
#endif
#ifdef RT_DBG_NO_FALLBACK  // (measurement only: what the in-line copy of the exact emulation costs in registers)
    sure = true;
#endif
    if (!sure) {
        RT_SIM_COUNT(6);
        RT_PCB();
        // `lead`: this tree is the FIRST entry of the world list but is tested after the list's plain primitives (world_hit's zone
        // order).  The reference searches it from t_max_list, the scan's initial distance, and that is the search emulated here;
        // what it finds counts only if it is closer than the primitives' closest hit (an equal distance: the order decides).
        // Any other tree met in a reordered scan: what the emulation finds depends on a starting distance only the list order
        // gives, so world_hit repeats the scan in list order.
        walk_begin<T, F>(sc, geom_index, o, d, time, t_min, lead ? t_max_list : t_max, w, false);
        walk_run<T, F>(sc, w, o, d, inv, time);
        if (lead) {
            w.cross_tie = w.any && w.closest == t_max;
            if (w.any && !(w.closest < t_max)) { w.any = false; w.closest = t_max; }
        } else {
            w.cross_tie = true;
        }
        RT_PCE(11);
    }
    if (w.any) { prim_out = w.prim; side_out = w.side; }
    t_out = w.closest;
    cross_tie |= w.cross_tie;
    return w.any;
}

// src/transforms.rs:15-17, 90-103: world ray -> object ray through the instance's wrapper chain.
// transforms.rs:34-39,124-135: pdf_value / random forward the origin (and the direction through
// RotateY) the same way.
// Returns whether the direction changed (a RotateY in the chain): the caller's shared reciprocals of d then need renewing.
template <class T, uint32_t F> __device__ inline bool xform_ray(const DScene<T>& sc, const DInstance<T>& in, V3<T>& o, V3<T>& d) {
    if (!has<F>(F_XFORM)) return false;
    bool rotated = false;
    for (uint32_t j = 0; j < in.xform_count; j++) {
        const DXform<T>& x = sc.xforms[in.xform_begin + j];
        if (x.kind == 0) {
            o = o - V3<T>(x.v);
        } else {
            T s = x.v[0], c = x.v[1];
            o = V3<T>(c * o.x - s * o.z, o.y, s * o.x + c * o.z);
            d = V3<T>(c * d.x - s * d.z, d.y, s * d.x + c * d.z);
            rotated = true;
        }
    }
    return rotated;
}
// src/transforms.rs:18-21, 105-116: hit record back to world space (innermost wrapper first)
template <class T, uint32_t F> __device__ inline void xform_back(const DScene<T>& sc, const DInstance<T>& in, V3<T>& p, V3<T>& n) {
    if (!has<F>(F_XFORM)) return;
    for (int j = (int)in.xform_count - 1; j >= 0; j--) {
        const DXform<T>& x = sc.xforms[in.xform_begin + j];
        if (x.kind == 0) {
            p = p + V3<T>(x.v);
        } else {
            T s = x.v[0], c = x.v[1];
            p = V3<T>(c * p.x + s * p.z, p.y, -s * p.x + c * p.z);
            n = V3<T>(c * n.x + s * n.z, n.y, -s * n.x + c * n.z);
        }
    }
}
template <class T, uint32_t F> __device__ inline V3<T> xform_dir_back(const DScene<T>& sc, const DInstance<T>& in, V3<T> v) {
    if (!has<F>(F_XFORM)) return v;
    for (int j = (int)in.xform_count - 1; j >= 0; j--) {
        const DXform<T>& x = sc.xforms[in.xform_begin + j];
        if (x.kind == 1) {
            T s = x.v[0], c = x.v[1];
            v = V3<T>(c * v.x + s * v.z, v.y, -s * v.x + c * v.z);
        }
    }
    return v;
}

struct MediumXi {  // where ConstantMedium gets xi from: the path's generator, or a pinned probe value
    Rng* rng;
    double fixed;
};
template <class T> __device__ inline T medium_xi(MediumXi& m) { return m.rng ? m.rng->gen<T>() : (T)m.fixed; }

template <class T> struct Hit {
    T t;
    int inst, prim, side;  // side 7 = ConstantMedium event
};

// Both boundary searches of a ConstantMedium (src/objects.rs:529-536: boundary.hit(-inf, inf), then boundary.hit(t1 + 0.0001, inf))
// on ONE sphere or cuboid.  The roots of the sphere / the plane distances and in-face tests of the six sides do not depend on the
// search range: they are computed once, and each search only redoes its range comparisons — the same comparisons, in the same
// order, on the same values as two separate prim_hit calls (sphere_hit; the cuboid loop of prim_hit with its shrinking `closest`).
// Returns how many of the two searches found a hit (the second is not made when the first fails); ta / tb: their distances.
// cornell_smoke: 12 of the 30 rectangle tests of a cast drop out; final_scene: 2 of its 11 sphere tests.
template <class T, uint32_t F>
__device__ inline int boundary_two(const DPrim<T>& p, V3<T> o, V3<T> d, T time, T& ta, T& tb) {
    const T lo0 = -Lim<T>::max(), hi = Lim<T>::max();
    if ((has<F>(F_SPHERE) || has<F>(F_MOVING)) && p.kind <= 1) {
        V3<T> center(p.d);
        T radius = p.d[3];
        if (has<F>(F_MOVING) && (!has<F>(F_SPHERE) || p.kind == 1)) { center = moving_center(p, time); radius = p.d[8]; }
        V3<T> oc = o - center;
        T a = dot(d, d);
        T half_b = dot(oc, d);
        T c = dot(oc, oc) - radius * radius;
        T discriminant = half_b * half_b - a * c;
        if (!(discriminant > T(0))) return 0;
        T sq = sqrt_r(discriminant);
        const PlainDiv<T> inv_a(a);
        const T r1 = inv_a.div(-half_b - sq), r2 = inv_a.div(-half_b + sq);
        bool h1 = (r1 > lo0) & (r1 < hi), h2 = (r2 > lo0) & (r2 < hi);
        if (!(h1 | h2)) return 0;
        ta = h1 ? r1 : r2;
        const T lo1 = ta + T(0.0001);
        h1 = (r1 > lo1) & (r1 < hi); h2 = (r2 > lo1) & (r2 < hi);
        if (!(h1 | h2)) return 1;
        tb = h1 ? r1 : r2;
        return 2;
    }
    if (has<F>(F_CUBOID) && p.kind == 5) {
        T ts[6];
        bool in[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {  // rect_hit's arithmetic, once per side
            int axis; T a1, a2, b1, b2, k, face;
            cuboid_side(p, i, axis, a1, a2, b1, b2, k, face);
            const T ok = axis == 0 ? o.z : (axis == 1 ? o.y : o.x), dk = axis == 0 ? d.z : (axis == 1 ? d.y : d.x);
            const T oa = axis == 2 ? o.y : o.x, da = axis == 2 ? d.y : d.x;
            const T ob = axis == 0 ? o.y : o.z, db = axis == 0 ? d.y : d.z;
            const T t = (k - ok) / dk;
            const T a = oa + t * da;
            const T b = ob + t * db;
            ts[i] = t;
            in[i] = (a >= a1) & (a <= a2) & (b >= b1) & (b <= b2);
        }
        int found = 0;
        T lo = lo0;
#pragma unroll
        for (int s = 0; s < 2; s++) {  // the six sides in tuple order with a shrinking `closest` (src/objects.rs:458-485), per search
            bool any = false;
            T closest = hi;
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const bool h = in[i] & (ts[i] >= lo) & (ts[i] <= closest);
                closest = h ? ts[i] : closest;
                any |= h;
            }
            if (!any) return found;
            found++;
            if (s == 0) { ta = closest; lo = ta + T(0.0001); } else tb = closest;
        }
        return found;
    }
    return -1;  // another kind of boundary: the caller searches twice through geom_hit
}

// src/objects.rs:38-48 (ObjectList::hit over the world) with Translate/RotateY wrappers and
// ConstantMedium (src/objects.rs:529-556) folded in.  The instance loop is wave-uniform.  A medium
// runs the boundary search twice through the SAME geom_hit call site (pass 0: (-max, max),
// pass 1: (t1 + 0.0001, max)), so the traversal code exists once.
template <class T, uint32_t F>
__device__ inline bool world_hit(const DScene<T>& sc, const Ray<T>& ray, T t_min, T t_max, MediumXi& mxi, Hit<T>& hit) {
    bool any = false;
    T closest = t_max;
    // A ray with a NaN in its origin or direction (a bounce whose pdf was 0/0; the sample will be retried,
    // ray_tracing.rs:107) hits nothing: every primitive test compares a NaN and fails, and no medium draws.
    // In the reference it still walks every BVH completely, because a NaN slab passes the loose box test
    // (f64::min/max ignore NaN).  Same result, none of the work:
    {
        const T chk = ray.ori.x + ray.ori.y + ray.ori.z + ray.dir.x + ray.dir.y + ray.dir.z;
        if (chk != chk) {
            const bool nan = (ray.ori.x != ray.ori.x) | (ray.ori.y != ray.ori.y) | (ray.ori.z != ray.ori.z) |
                             (ray.dir.x != ray.dir.x) | (ray.dir.y != ray.dir.y) | (ray.dir.z != ray.dir.z);
            if (nan) return false;  // (chk is also NaN for inf - inf: those rays take the normal route)
        }
    }
    // The zone (DScene::world_order) is scanned in ITS order first.  Every hit is the closest of all accepted candidates whatever the
    // order — unless two candidate distances are EQUAL, where the reference's list order and the strict / inclusive comparisons of the
    // primitive kinds decide.  Each test reports a candidate equal to the closest distance so far (cross_tie); a wave in
    // which some ray saw one scans the zone again in list order, which is the reference's scan.  (No medium lies in the zone, so the
    // repeat draws nothing from the generator.)
    // Only where the device's own trees are walked (ray times inside the fast range, which is empty for a description whose node
    // boxes are not nested): walk_fast finds the closest of ALL accepted hits below any starting distance and vouches for it
    // independently of that distance, while the emulation of the reference's walk (walk_run) depends on the starting distance the
    // reference would have had — a walk that falls back to it also asks for the repeat.
    const uint32_t n_zone = has<F>(F_BVH) ? sc.n_zone : 0u;
    bool zone_order = n_zone != 0u && !wave_any(!(ray.time >= sc.fast_lo && ray.time <= sc.fast_hi)), zone_tie = false;
    for (uint32_t j = 0; j < sc.n_world; j++) {
        const uint32_t i = (has<F>(F_BVH) && zone_order && j < n_zone) ? sc.world_order[j] : j;
#if defined(RT_LAZY_RECORDS)
        const RecordOf<false, T> entry(sc.world + i);
#else
        // in one go where it measured faster (round 3, f64: cornell_box +2.1 %, two_perlin_spheres +0.4 %); the variants with media or
        // (u,v) are short of registers and lost (cornell_smoke -1.8 %, earth -3.5 %: 32 more scalars live, 48 -> 80 B of spills), and
        // with trees in the list the walks dominate
        const RecordOf<!(has<F>(F_BVH) || has<F>(F_MEDIUM) || has<F>(F_UV)), T> entry(sc.world + i);
#endif
        const DInstance<T>& in = entry.r;
        V3<T> o = ray.ori, d = ray.dir;
        RT_PCB();
#if defined(RT_SHARED_RECIP)
        // rectangle tests divide by a direction component: the reciprocal part of those divisions is shared per axis and,
        // where no RotateY intervenes, per ray (Recip)
        constexpr bool kRects = has<F>(F_RECT) || has<F>(F_CUBOID);
        const bool rotated = xform_ray<T, F>(sc, in, o, d);
        RayInv<T> inv(kRects ? ray.dir : V3<T>(T(1), T(1), T(1)));
        if (kRects && rotated) { inv.x = Recip<T>(d.x); inv.z = Recip<T>(d.z); }  // RotateY keeps d.y (wave-uniform branch)
#else
        xform_ray<T, F>(sc, in, o, d);
        const RayInv<T> inv(d);  // plain divisions: just the direction
#endif
        RT_PCE(8);
        const bool med = has<F>(F_MEDIUM) && in.is_medium;
        T ta = T(0);
        // what a medium does once both boundary distances are known (src/objects.rs:537-556)
        auto medium_event = [&](T tb) {
            T t1 = ta < t_min ? t_min : ta;
            if (tb > closest) tb = closest;
            if (t1 < tb) {
                T len = length(ray.dir);
                T distance_inside_boundary = (tb - t1) * len;
                T hit_distance = -(T(1) / in.density) * rlog(medium_xi<T>(mxi));
                if (hit_distance <= distance_inside_boundary) {
                    const T t = t1 + hit_distance / len;
                    any = true; closest = t; hit.t = t; hit.inst = (int)i; hit.side = 7;
                    int first = (int)in.geom_index;  // boundary's first primitive in pre-order
                    if (has<F>(F_BVH) && has<F>(F_MEDIUM_BVH) && in.geom_kind != 0) {
                        int e = sc.bvh_root_entry[in.geom_index];
                        while (sc.bvh[e].prim < 0) e++;
                        first = sc.bvh[e].prim;
                    }
                    hit.prim = first;
                }
            }
        };
        if (has<F>(F_MEDIUM) && med && in.geom_kind == 0) {  // a single sphere or cuboid as boundary: both searches share their arithmetic
            T tb = T(0);
            const int n = boundary_two<T, F>(prim_of<T, F>(sc, in), o, d, ray.time, ta, tb);
            if (n >= 0) {
                if (n == 2) medium_event(tb);
                RT_PCE(7);
                continue;
            }
        }
        if (has<F>(F_MEDIUM) && med) {
            // any other boundary: two searches, (-max, max) then (t1 + 0.0001, max) (src/objects.rs:529-536).  Without F_MEDIUM_BVH the
            // boundary is one primitive and the walk code below is not part of this branch (nor its state live across it).
            T tb = T(0);
            int found = 0;
#pragma nounroll
            for (int pass = 0; pass < 2; pass++) {
                const T lo = pass == 0 ? -Lim<T>::max() : ta + T(0.0001);
                T t = Lim<T>::max(); int prim = -1, side = 0;
                bool h;
                bool unused = false;
                if (has<F>(F_MEDIUM_BVH)) h = geom_hit<T, F>(sc, in, o, d, inv, ray.time, lo, Lim<T>::max(), t, prim, side, unused);
                else h = prim_hit<T, F>(prim_of<T, F>(sc, in), o, d, inv, ray.time, lo, Lim<T>::max(), t, side);
                if (!h) break;
                found++;
                if (pass == 0) ta = t; else tb = t;
            }
            if (found == 2) medium_event(tb);
            RT_PCE(7);
            continue;
        }
        {
            T t = closest; int prim = -1, side = 0;
            bool tie = false;
            const bool lead = has<F>(F_BVH) && zone_order && i == 0u;  // (list entry 0 met later than in the list: only as a tree of the zone)
            if (geom_hit<T, F>(sc, in, o, d, inv, ray.time, t_min, closest, t, prim, side, tie, lead, t_max)) {
                any = true; closest = t; hit.t = t; hit.inst = (int)i; hit.prim = prim; hit.side = side;
            }
            if (has<F>(F_BVH) && sizeof(T) == 8) zone_tie |= tie;
        }
        if (has<F>(F_BVH) && sizeof(T) == 8 && zone_order && j + 1u == n_zone) {
            zone_order = false;  // from here on (and in a repeat) the list order
            if (wave_any(zone_tie)) {
                RT_SIM_COUNT(6);
                any = false; closest = t_max;
                j = (uint32_t)-1;  // (wave-uniform restart)
            }
        }
    }
    return any;
}

// src/objects.rs:208-215
// The f64 form is a real CALL (noinline), not inlined code: ocml's f64 atan2 / asin want ~60 registers, and inlined at the end of the
// cast — where the whole path state is live — they were what the f64 BVH kernels spilled for (round 3, scratch per lane with the call:
// spheres + BVH 248 -> 16 B, generic 328 -> 96 B, flat spheres 230 -> 164 registers; final_scene +1.5 %, random_scene +0.4 %, earth
// -0.3 %).  The call is only made by waves in which some lane's material reads (u,v) (an image or uv-checker texture); in f32 the
// functions are small and stay inline (the call measured -1..-2 % there).
template <class T> struct UV { T u, v; };
template <class T> __device__ inline UV<T> sphere_uv_body(T px, T py, T pz) {
    T phi = ratan2(pz, px);
    T theta = rasin(py);
    UV<T> r;
    r.u = T(1) - (phi + Consts<T>::pi()) / (T(2) * Consts<T>::pi());
    r.v = (theta + Consts<T>::pi() / T(2)) / Consts<T>::pi();
    return r;
}
#if !defined(RT_HOST_SIM) && !defined(RT_INLINE_UV)
__device__ __attribute__((noinline)) inline UV<double> sphere_uv_call(double px, double py, double pz) { return sphere_uv_body<double>(px, py, pz); }
#else
__device__ inline UV<double> sphere_uv_call(double px, double py, double pz) { return sphere_uv_body<double>(px, py, pz); }
#endif
__device__ inline UV<float> sphere_uv_call(float px, float py, float pz) { return sphere_uv_body<float>(px, py, pz); }
template <class T> __device__ inline void sphere_uv(V3<T> p, T& u, T& v) {
    const UV<T> r = sphere_uv_call(p.x, p.y, p.z);
    u = r.u; v = r.v;
}

template <class T> struct HitRecord { T t; V3<T> p, normal; T u, v; };

// Fill the HitRecord of the winning hit (reference formulas of each primitive's `hit`).
template <class T, uint32_t F>
__device__ inline void finish_hit(const DScene<T>& sc, const Ray<T>& ray, const Hit<T>& h, bool need_uv, HitRecord<T>& rec) {
    const DInstance<T>& in = sc.world[h.inst];
    rec.t = h.t;
    rec.u = T(0); rec.v = T(0);
    if (has<F>(F_MEDIUM) && h.side == 7) {  // src/objects.rs:543-550
        rec.p = at(ray, h.t);
        rec.normal = V3<T>(T(1), T(0), T(0));
        return;
    }
    Ray<T> r = ray;
    xform_ray<T, F>(sc, in, r.ori, r.dir);
    const DPrim<T>& p = sc.prims[h.prim];
    rec.p = at(r, h.t);
    if ((has<F>(F_SPHERE) || has<F>(F_MOVING)) && p.kind <= 1) {
        V3<T> center = (!has<F>(F_MOVING) || p.kind == 0) ? V3<T>(p.d) : moving_center(p, ray.time);
        T radius = (!has<F>(F_MOVING) || p.kind == 0) ? p.d[3] : p.d[8];
        rec.normal = (rec.p - center) / radius;
        if (has<F>(F_UV) && need_uv) {  // uv from the FIRST root even when the second was returned (objects.rs:94-95)
            V3<T> oc = r.ori - center;
            T a = dot(r.dir, r.dir);
            T half_b = dot(oc, r.dir);
            T c = dot(oc, oc) - radius * radius;
            T discriminant = half_b * half_b - a * c;
            T ans = (-half_b - sqrt_r(discriminant)) / a;
            sphere_uv((at(r, ans) - center) / radius, rec.u, rec.v);
        }
    } else {
        int axis; T a1, a2, b1, b2, k, face;
        if (has<F>(F_CUBOID) && p.kind == 5) cuboid_side(p, h.side, axis, a1, a2, b1, b2, k, face);
        else { axis = (int)p.kind - 2; a1 = p.d[0]; a2 = p.d[1]; b1 = p.d[2]; b2 = p.d[3]; k = p.d[4]; face = p.d[5]; }
        const T z0 = T(0) * face;  // the reference multiplies the axis vector by face: the zero components carry face's sign
        rec.normal = V3<T>(axis == 2 ? face : z0, axis == 1 ? face : z0, axis == 0 ? face : z0);
        if (has<F>(F_UV) && need_uv) {
            const T oa = axis == 2 ? r.ori.y : r.ori.x, da = axis == 2 ? r.dir.y : r.dir.x;
            const T ob = axis == 0 ? r.ori.y : r.ori.z, db = axis == 0 ? r.dir.y : r.dir.z;
            T a = oa + h.t * da;
            T b = ob + h.t * db;
            rec.u = (a - a1) / (a2 - a1);
            rec.v = (b - b1) / (b2 - b1);
        }
    }
    xform_back<T, F>(sc, in, rec.p, rec.normal);
}

// ------------------------------------------------------------------------------------------
// textures: src/texture.rs
// ------------------------------------------------------------------------------------------
// Rust `as i32` / `as u32`: truncating, saturating, NaN -> 0 (negative -> 0 for u32).  That is, to the letter, what the hardware's
// conversion instructions do (V_CVT_I32_F64 etc.: round toward zero, out-of-range values and infinities saturate, NaN gives 0), while a
// C++ cast of an out-of-range value is undefined — so the portable form below spells the cases out, and the compiler turns each into
// ~14 instructions with two exec-mask branches around the very instruction that needs none of them.  The device code issues the
// instruction itself (a Perlin turbulence lookup converts 21 times: ~290 of its ~1250 instructions; round 3).
// tests/test_gpu_divide.py::test_float_to_int_conversions_are_rusts checks both forms against each other on the device.
__device__ inline int32_t as_i32_portable(double x) { return x != x ? 0 : (x >= 2147483647.0 ? 2147483647 : (x <= -2147483648.0 ? (-2147483647 - 1) : (int32_t)x)); }
__device__ inline int32_t as_i32_portable(float x) { return x != x ? 0 : (x >= 2147483648.0f ? 2147483647 : (x <= -2147483648.0f ? (-2147483647 - 1) : (int32_t)x)); }
__device__ inline uint32_t as_u32_portable(double x) { return x != x ? 0u : (x >= 4294967295.0 ? 4294967295u : (x <= 0.0 ? 0u : (uint32_t)x)); }
__device__ inline uint32_t as_u32_portable(float x) { return x != x ? 0u : (x >= 4294967296.0f ? 4294967295u : (x <= 0.0f ? 0u : (uint32_t)x)); }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_HOST_SIM) && !defined(RT_PORTABLE_CONVERT)
__device__ inline int32_t as_i32(double x) { int32_t r; asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(x)); return r; }
__device__ inline int32_t as_i32(float x) { int32_t r; asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x)); return r; }
__device__ inline uint32_t as_u32(double x) { uint32_t r; asm("v_cvt_u32_f64 %0, %1" : "=v"(r) : "v"(x)); return r; }
__device__ inline uint32_t as_u32(float x) { uint32_t r; asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(x)); return r; }
#else
__device__ inline int32_t as_i32(double x) { return as_i32_portable(x); }
__device__ inline int32_t as_i32(float x) { return as_i32_portable(x); }
__device__ inline uint32_t as_u32(double x) { return as_u32_portable(x); }
__device__ inline uint32_t as_u32(float x) { return as_u32_portable(x); }
#endif

// Where a Perlin table is read from: the scene's global record, or the render kernel's LDS copy.
template <class T> struct PerlinGlobal {
    const DPerlin<T>* p;
    __device__ uint32_t px(uint32_t i) const { return p->perm[0][i]; }
    __device__ uint32_t py(uint32_t i) const { return p->perm[1][i]; }
    __device__ uint32_t pz(uint32_t i) const { return p->perm[2][i]; }
    __device__ V3<T> vec(uint32_t i) const { return V3<T>(p->ranvec[i][0], p->ranvec[i][1], p->ranvec[i][2]); }
};
template <class T> struct PerlinLds {
#if defined(RT_HOST_SIM) || !defined(__HIP_DEVICE_COMPILE__)
    PerlinGlobal<T> g;
    __device__ explicit PerlinLds(const DPerlin<T>* q) : g{q} {}
    __device__ uint32_t px(uint32_t i) const { return g.px(i); }
    __device__ uint32_t py(uint32_t i) const { return g.py(i); }
    __device__ uint32_t pz(uint32_t i) const { return g.pz(i); }
    __device__ V3<T> vec(uint32_t i) const { return g.vec(i); }
#else
    // explicit LDS address space: ds_read_* instead of flat loads
    typedef const __attribute__((address_space(3))) uint8_t* lds_u8;
    typedef const __attribute__((address_space(3))) T* lds_t;
    lds_t v;
    lds_u8 b;
    __device__ explicit PerlinLds(const DPerlin<T>* q)
        : v((lds_t)(uintptr_t)(uint32_t)(uintptr_t)q), b((lds_u8)(uintptr_t)(uint32_t)(uintptr_t)q + sizeof(T) * 256u * DPerlin<T>::kStride) {}
    __device__ uint32_t px(uint32_t i) const { return b[i]; }
    __device__ uint32_t py(uint32_t i) const { return b[256u + i]; }
    __device__ uint32_t pz(uint32_t i) const { return b[512u + i]; }
    __device__ V3<T> vec(uint32_t i) const {
        constexpr uint32_t S = DPerlin<T>::kStride;
        return V3<T>(v[S * i], v[S * i + 1u], v[S * i + 2u]);
    }
#endif
};

// src/texture.rs:139-165
template <class T, class P> __device__ inline T perlin_noise(const P& pl, V3<T> p) {
    T fx = rfloor(p.x), fy = rfloor(p.y), fz = rfloor(p.z);
    T u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int32_t i = as_i32(fx), j = as_i32(fy), k = as_i32(fz);
    T uu = u * u * (T(3) - T(2) * u);
    T vv = v * v * (T(3) - T(2) * v);
    T ww = w * w * (T(3) - T(2) * w);
    const uint32_t hx[2] = {pl.px((uint32_t)i & 255u), pl.px((uint32_t)(i + 1) & 255u)};
    const uint32_t hy[2] = {pl.py((uint32_t)j & 255u), pl.py((uint32_t)(j + 1) & 255u)};
    const uint32_t hz[2] = {pl.pz((uint32_t)k & 255u), pl.pz((uint32_t)(k + 1) & 255u)};
    // texture.rs:156-160 writes a corner's weight as `i·uu + (1 − i)·(1 − uu)` with i = 0 or 1.  For every finite uu that IS uu
    // (i = 1: uu + 0·(1 − uu) = uu ± 0) or 1 − uu (i = 0: 0·uu + (1 − uu)), bit for bit — uu = u²(3 − 2u) ≥ +0, so no signed-zero case
    // arises — and for a NaN both forms are NaN.  The compiler may not drop `0·x` or `+ 0` by itself (NaN, signed zeros): written
    // out, 12 of a lookup's ~135 vector instructions go (two_perlin_spheres: seven lookups per shade).
#ifdef RT_PERLIN_REFERENCE_WEIGHTS
    const T wu[2] = {T(0) * uu + T(1) * (T(1) - uu), T(1) * uu + T(0) * (T(1) - uu)};
    const T wv[2] = {T(0) * vv + T(1) * (T(1) - vv), T(1) * vv + T(0) * (T(1) - vv)};
    const T wz[2] = {T(0) * ww + T(1) * (T(1) - ww), T(1) * ww + T(0) * (T(1) - ww)};
#else
    const T wu[2] = {T(1) - uu, uu}, wv[2] = {T(1) - vv, vv}, wz[2] = {T(1) - ww, ww};
#endif
    T accum = T(0);
#pragma unroll
    for (int di = 0; di < 2; di++)
#pragma unroll
        for (int dj = 0; dj < 2; dj++)
#pragma unroll
            for (int dk = 0; dk < 2; dk++) {
                V3<T> c = pl.vec(hx[di] ^ hy[dj] ^ hz[dk]);
                accum += wu[di] * wv[dj] * wz[dk] * dot(c, V3<T>(u - T(di), v - T(dj), w - T(dk)));
            }
    return accum;
}
// src/texture.rs:166-176
template <class T, class P> __device__ inline T perlin_turb(const P& pl, V3<T> p, int depth) {
    T accum = T(0);
    V3<T> temp_p = p;
    T weight = T(1);
#pragma nounroll
    for (int i = 0; i < depth; i++) {
        accum += weight * perlin_noise<T, P>(pl, temp_p);
        weight *= T(0.5);
        temp_p = temp_p * T(2);
    }
    return rabs(accum);
}

// CheckerTexture (texture.rs:42-49) only asks whether sin(10x)·sin(10y)·sin(10z) < 0 — the SIGN of three sines, not their values.
// sin_sign: the sign of sin(a) from one Cody–Waite step (fdlibm's __ieee754_rem_pio2 medium path: q = rint(|a|·2/π),
// r = (|a| − q·pio2_1) − q·pio2_1t; q·pio2_1 is exact for q < 2^20) and the quadrant: sin|a| is sin r, cos r, −sin r, −cos r for
// q mod 4 = 0..3, and cos r > 0 on |r| ≤ π/4.  The error of r is below 2·10^-20 + 2^-53·|r| for |a| < 2^20, so its sign is certain
// wherever |r| > 10^-9; closer to a multiple of π, beyond 2^20, or for a NaN, `sure` is cleared and the caller evaluates the sines
// themselves.  A correctly signed sine is what every libm returns (sin a = 0 only at a = 0: the product is then ±0, not negative),
// so the decision is the reference's, not an approximation of it: ~12 instructions per axis instead of a ~90-instruction call
// (tests/test_hostsim.py::test_checker_sign_decision_is_the_sine_product: 8·10^5 argument triples incl. neighbours of k·π).
__device__ inline int sin_sign(double a, bool& sure) {
    const double aa = ::fabs(a);
    if (!(aa < 1048576.0)) { sure = false; return 0; }
    if (aa == 0.0) return 0;
    const double q = ::rint(aa * 6.36619772367581382433e-01);
    const double r = (aa - q * 1.57079632673412561417e+00) - q * 6.07710050650619224932e-11;
    const int qi = (int)q & 3;
    int sgn;
    if (qi & 1) sgn = qi == 1 ? 1 : -1;
    else {
        if (!(::fabs(r) > 1e-9)) sure = false;
        sgn = (r > 0.0) == (qi == 0) ? 1 : -1;
    }
    return a < 0.0 ? -sgn : sgn;
}
__device__ inline bool checker_is_odd_body(double ax, double ay, double az) {
    bool sure = true;
    const int s = sin_sign(ax, sure) * sin_sign(ay, sure) * sin_sign(az, sure);
    if (__builtin_expect(!sure, 0)) return rsin(ax) * rsin(ay) * rsin(az) < 0.0;
    return s < 0;
}
// Used (inlined) by the spheres + BVH variant: random_scene +2.5 %.  The other variants keep the three sine calls: inlined, the body
// cost the generic variant 32 B of spills (final_scene, which has no checker at all, -3 %); as a call — a callee that calls in turn —
// 16 B of frame: 128 instead of 112 B of scratch per lane, with which the resident waves' scratch (33.5 MB) no longer fits the 32 MB
// of L2 and the memory-side traffic of a final_scene launch doubles (8.7 -> 18.6 GB); the flat sphere variant 48 -> 64 B for no
// measured gain.  f32 evaluates the sines.
template <class T, uint32_t F> __device__ inline bool checker_is_odd(T ax, T ay, T az) {
#ifndef RT_CHECKER_SINES
    if constexpr (sizeof(T) == 8 && has<F>(F_BVH) && !(has<F>(F_MEDIUM) || has<F>(F_XFORM) || has<F>(F_LIGHTS) || has<F>(F_NOISE)))
        return checker_is_odd_body(ax, ay, az);
#endif
    return rsin(ax) * rsin(ay) * rsin(az) < T(0);
}

template <class T, uint32_t F> __device__ inline V3<T> texture_value(const DScene<T>& sc, uint32_t tex, T u, T v, V3<T> p) {
    if (!has<F>(F_TEXTURED)) return V3<T>(sc.textures[tex].color);  // texture.rs:16-18
    for (int guard = 0; guard < 16; guard++) {  // nested checkers: bounded descent instead of recursion
        const DTexture<T>& t = sc.textures[tex];
        switch (t.kind) {
            case 0: return V3<T>(t.color);                                           // texture.rs:16-18
            case 1: {                                                                // texture.rs:42-49
                tex = checker_is_odd<T, F>(T(10) * p.x, T(10) * p.y, T(10) * p.z) ? t.odd : t.even;
                break;
            }
            case 2:                                                                  // texture.rs:27-33
                tex = (as_i32(u / T(0.01)) % 2 == as_i32(v / T(0.01)) % 2) ? t.odd : t.even;
                break;
            case 3:                                                                  // texture.rs:58-60
            {
                T turb;
                if (has<F>(F_NOISE) && t.table < sc.n_perlin_lds) turb = perlin_turb<T>(PerlinLds<T>(sc.perlin_lds + t.table), p, 7);
                else turb = perlin_turb<T>(PerlinGlobal<T>{&sc.perlins[t.table]}, p, 7);
                return V3<T>(T(1), T(1), T(1)) * T(0.5) * (T(1) + rsin(t.scale * p.z + T(10) * turb));
            }
            default: {                                                               // texture.rs:77-97
                const DImage& im = sc.images[t.table];
                T uc = u < T(0) ? T(0) : (u > T(1) ? T(1) : u);
                T vc = v < T(0) ? T(0) : (v > T(1) ? T(1) : v);
                uint32_t x = as_u32(uc * (T)im.width);
                if (x > im.width - 1) x = im.width - 1;
                uint32_t y = as_u32(vc * (T)im.height);
                if (y > im.height - 1) y = im.height - 1;
                const size_t texel = (size_t)(im.height - 1 - y) * im.width + x;
                T color_scale = T(1) / T(255);
#if defined(RT_TEX_LDS_CACHE) && defined(__HIP_DEVICE_COMPILE__) && !defined(RT_HOST_SIM)
                // EXPERIMENT (profiles/r02_texture_tiles.md): texels through a direct-mapped cache in LDS shared by the block.
                // An entry is one 8-byte word {tag = image, texel index + 1 | rgb}: written and read whole, so a racing
                // refill by another wave leaves either the old or the new pair, both consistent.
                if (sc.tex_cache) {
                    typedef __attribute__((address_space(3))) unsigned long long* lds_u64;
                    const uint32_t tag = (uint32_t)texel + 1u + (t.table << 26);
                    lds_u64 slot = (lds_u64)(uintptr_t)(uint32_t)(uintptr_t)sc.tex_cache + (((uint32_t)texel ^ ((uint32_t)texel >> 10)) & (uint32_t)(RT_TEX_LDS_CACHE - 1));
                    unsigned long long ent = *slot;
                    if ((uint32_t)(ent >> 32) != tag) {
                        const uint8_t* px = im.rgb + texel * 3;
                        ent = ((unsigned long long)tag << 32) | ((unsigned long long)px[0] << 16) | ((unsigned long long)px[1] << 8) | px[2];
                        *slot = ent;
                    }
                    return V3<T>((T)(uint32_t)((ent >> 16) & 255u) * color_scale, (T)(uint32_t)((ent >> 8) & 255u) * color_scale, (T)(uint32_t)(ent & 255u) * color_scale);
                }
#endif
                const uint8_t* px = im.rgb + texel * 3;
                return V3<T>((T)px[0] * color_scale, (T)px[1] * color_scale, (T)px[2] * color_scale);
            }
        }
    }
    return V3<T>(T(0), T(0), T(0));
}

}  // namespace rt
