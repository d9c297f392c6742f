// rt_path.hpp — light importance sampling, material scatter and one segment of the iterative path
// loop.  Restates src/pdf.rs, src/materials.rs and src/ray_tracing.rs:11-60 for the device.
// Pure device functions (no kernels): tests/hostsim compiles this header for the CPU and checks it
// against the oracle.
#pragma once
#include "rt_device.hpp"

namespace rt {

// ------------------------------------------------------------------------------------------
// light list: Object::pdf_value through ObjectList, Translate, RotateY
// ------------------------------------------------------------------------------------------
// src/objects.rs:251-261, 305-315, 359-369
template <class T>
__device__ inline T rect_pdf(int axis, T a1, T a2, T b1, T b2, T k, T face, V3<T> o, V3<T> v) {
    T t = T(0);
    const T vk = axis == 0 ? v.z : (axis == 1 ? v.y : v.x);
    if (rect_hit<T>(axis, a1, a2, b1, b2, k, o, v, PlainDiv<T>(vk), T(0.001), Lim<T>::max(), t)) {
        const T z0 = T(0) * face;
        V3<T> normal(axis == 2 ? face : z0, axis == 1 ? face : z0, axis == 0 ? face : z0);
        T area = (a2 - a1) * (b2 - b1);
        T distance_squared = t * t * sqlen(v);
        T cosine = rabs(dot(v, normal)) / length(v);
        return distance_squared / (cosine * area);
    }
    return T(0);
}

// p: the primitive's fields (may be a copy held in registers); pm: the same primitive in memory, for the run-time subscripts of
// the cuboid's side loop (a subscripted copy would stop being registers)
template <class T, uint32_t F> __device__ inline T prim_pdf_value(const DPrim<T>& p, const DPrim<T>& pm, V3<T> o, V3<T> v) {
    const uint32_t kind = p.kind;
    if (has<F>(F_SPHERE) && kind == 0) {  // src/objects.rs:126-137
        T t;
        V3<T> center(p.d);
        T radius = p.d[3];
        bool tie;
        if (sphere_hit(center, radius, o, v, T(0.001), Lim<T>::max(), t, tie)) {
            T cos_theta_max = sqrt_r(T(1) - radius * radius / sqlen(center - o));
            T solid_angle = T(2) * Consts<T>::pi() * (T(1) - cos_theta_max);
            return T(1) / solid_angle;
        }
        return T(0);
    }
    if (has<F>(F_RECT) && kind >= 2 && kind <= 4) {  // one copy per axis (see prim_hit)
        if (kind == 2) return rect_pdf<T>(0, p.d[0], p.d[1], p.d[2], p.d[3], p.d[4], p.d[5], o, v);
        if (kind == 3) return rect_pdf<T>(1, p.d[0], p.d[1], p.d[2], p.d[3], p.d[4], p.d[5], o, v);
        return rect_pdf<T>(2, p.d[0], p.d[1], p.d[2], p.d[3], p.d[4], p.d[5], o, v);
    }
    if (has<F>(F_CUBOID) && kind == 5) {  // src/objects.rs:492-500
        T weight = T(1) / T(6);
        T ans = T(0);
#pragma nounroll
        for (int i = 0; i < 6; i++) {
            int axis; T a1, a2, b1, b2, k, face;
            cuboid_side(pm, i, axis, a1, a2, b1, b2, k, face);
            T term = weight * rect_pdf<T>(axis, a1, a2, b1, b2, k, face, o, v);
            ans = i == 0 ? term : ans + term;
        }
        return ans;
    }
    return T(0);  // MovingSphere keeps the trait default (src/objects.rs:22-24)
}
// src/objects.rs:68-75 over the light list (wave-uniform loop)
template <class T, uint32_t F> __device__ inline T lights_pdf_value(const DScene<T>& sc, V3<T> origin, V3<T> v) {
    T weight = T(1) / (T)sc.n_lights;
    T ans = T(0);
    for (uint32_t i = 0; i < sc.n_lights; i++) {
#if defined(RT_LAZY_RECORDS) || defined(RT_LAZY_LIGHTS)
        const RecordOf<false, T> entry(sc.lights + i);
#else
        const RecordOf<!(has<F>(F_BVH) || has<F>(F_MEDIUM) || has<F>(F_UV)), T> entry(sc.lights + i);  // (as world_hit reads its list)
#endif
        const DInstance<T>& in = entry.r;
        T pv = T(0);
        if (!in.is_medium && in.geom_kind == 0) {  // ConstantMedium / BvhNode keep the default 0
            V3<T> o = origin, d = v;
            xform_ray<T, F>(sc, in, o, d);
            pv = prim_pdf_value<T, F>(in.prim, sc.lights[i].prim, o, d);
        }
        ans += weight * pv;
    }
    return ans;
}

// src/materials.rs:6-9 (powi(5) = x2*x2*x)
template <class T> __device__ inline T schlick(T cosine, T ref_idx) {
    T r0 = ((T(1) - ref_idx) / (T(1) + ref_idx)) * ((T(1) - ref_idx) / (T(1) + ref_idx));
    T x = T(1) - cosine;
    T x2 = x * x;
    return r0 + (T(1) - r0) * (x2 * x2 * x);
}

template <class T> __device__ inline bool is_nan3(V3<T> a) { return a.x != a.x || a.y != a.y || a.z != a.z; }

// One lane's path state (the recursion of ray_color unrolled into throughput/radiance form).
template <class T> struct PathState {
    Ray<T> ray;
    V3<T> beta, L;
    int seg;
};

// Direction of a Lambertian bounce and its mixture pdf (src/ray_tracing.rs:25-37, src/pdf.rs:32-94,
// Object::random of src/objects.rs:76-78,138-143,262-269,501-511 and src/transforms.rs:34-39,
// 136-149), written so that the wave stays converged: every lane draws the same two raw numbers,
// the cosine lobe and the cone towards a sphere light share ONE tangent-frame construction and ONE
// sincos, and the rectangle lights only differ in how the two numbers are scaled.  The draws are
// consumed in the reference's order (mixture choice, light index, [cuboid side,] then two
// numbers), so the f64 instantiation follows the oracle's paths sample for sample.
template <class T, uint32_t F>
__device__ inline void lambertian_sample(const DScene<T>& sc, const HitRecord<T>& rec, Rng& rng, V3<T>& dir_out, T& pdf_out,
                                         T& scattering_pdf_out) {
    typedef typename RawOf<T>::type raw_t;
    const V3<T> w_n = unit(rec.normal);  // CosinePDF's uvw.w (pdf.rs:12)
    // mode 0: cosine lobe   1: cone to a sphere light   2: point on a rect light   3: trait default (1,0,0)
    int mode = 0;
    uint32_t li = 0;
    int axis = 0;
    T q0 = T(0), q1 = T(1), q2 = T(0), q3 = T(1), q4 = T(0);  // rect: a1,a2,b1,b2,k   sphere: cx,cy,cz,radius
    V3<T> o_l = rec.p;  // the hit point in the light's object space
    bool medium_light = false;
    if (has<F>(F_LIGHTS) && sc.n_lights > 0) {
        if (rng.gen<T>() < T(0.5)) {  // MixturePDF::generate, pdf.rs:88
            li = rng.gen_index(sc.n_lights);  // objects.rs:77
            const DInstance<T>& in = sc.lights[li];
            mode = 3;
            medium_light = in.is_medium != 0;
            if (!in.is_medium && in.geom_kind == 0) {
                V3<T> dummy(T(0), T(0), T(0));
                xform_ray<T, F>(sc, in, o_l, dummy);
                const DPrim<T>& p = in.prim;
                const uint32_t kind = p.kind;
                if (has<F>(F_SPHERE) && kind == 0) {
                    mode = 1; q0 = p.d[0]; q1 = p.d[1]; q2 = p.d[2]; q3 = p.d[3];
                } else if (has<F>(F_RECT) && kind >= 2 && kind <= 4) {
                    mode = 2; axis = (int)kind - 2; q0 = p.d[0]; q1 = p.d[1]; q2 = p.d[2]; q3 = p.d[3]; q4 = p.d[4];
                } else if (has<F>(F_CUBOID) && kind == 5) {
                    T face;
                    cuboid_side(p, (int)rng.gen_index(6), axis, q0, q1, q2, q3, q4, face);  // objects.rs:502
                    mode = 2;
                }
            }
        }
    }
    V3<T> dir(T(1), T(0), T(0));  // Object::random default (objects.rs:25-27)
    if (mode != 3) {
        raw_t x1, x2;
        rng.raw(x1); rng.raw(x2);
        V3<T> local_pt(T(0), T(0), T(0));
        if (mode == 2) {  // objects.rs:262-269 etc.: gen_range(a1,a2), gen_range(b1,b2); a redraw (value rounds to hi) is rare
            T a = Rng::range_of(x1, q0, q1);
            if (!(a < q1)) { a = Rng::range_of(x2, q0, q1); while (!(a < q1)) a = rng.gen_range(q0, q1); rng.raw(x2); }
            T b = Rng::range_of(x2, q2, q3);
            while (!(b < q3)) b = rng.gen_range(q2, q3);
            V3<T> pt = axis == 0 ? V3<T>(a, b, q4) : (axis == 1 ? V3<T>(a, q4, b) : V3<T>(q4, a, b));
            local_pt = pt - o_l;
        }
        // cosine lobe around the normal (vec3.rs:231-239) / cone around the direction to the sphere (vec3.rs:240-248)
        const T r1 = Rng::unit_of(x1, T(0)), r2 = Rng::unit_of(x2, T(0));
        const bool cone = mode == 1;
        const V3<T> to_center = V3<T>(q0, q1, q2) - o_l;
        const T distance_squared = cone ? sqlen(to_center) : T(1);
        V3<T> w_f = w_n;
        if (cone) w_f = unit(to_center);  // ONB::build_from_w(direction), objects.rs:141
        V3<T> tu, tv;
        onb_tangents(w_f, tu, tv);
        const T zarg = cone ? T(1) - q3 * q3 / distance_squared : T(1) - r2;
        const T zs = sqrt_r(zarg);
        const T z = cone ? T(1) + r2 * (zs - T(1)) : zs;
        const T s = sqrt_r(cone ? T(1) - z * z : r2);
        const T phi = T(2) * Consts<T>::pi() * r1;
        T sn, cs;
        // phi = 2 pi r1, r1 in [0, 1): the range-limited routine, a call (rt_device.hpp: sincos_0_2pi)
        rsincos_0_2pi(phi, sn, cs);
        const T x = cs * s, y = sn * s;
        const V3<T> lobe = tu * x + tv * y + w_f * z;  // ONB::local, pdf.rs:19-24
        dir = mode == 2 ? local_pt : lobe;
    }
    if (has<F>(F_LIGHTS) && mode != 0 && !medium_light) dir = xform_dir_back<T, F>(sc, sc.lights[li], dir);
    // pdf.rs:43-46 (CosinePDF::value), pdf.rs:84-87 (MixturePDF::value), materials.rs:42-49 (scattering_pdf)
    const V3<T> ud = unit(dir);
    const T cosine = dot(ud, w_n);
    const T cos_pdf = cosine > T(0) ? cosine / Consts<T>::pi() : T(0);
    T pdf_value = cos_pdf;
    if (has<F>(F_LIGHTS) && sc.n_lights > 0) pdf_value = T(0.5) * lights_pdf_value<T, F>(sc, rec.p, dir) + T(0.5) * cos_pdf;
    const T sc_cos = dot(rec.normal, ud);
    scattering_pdf_out = sc_cos > T(0) ? sc_cos / Consts<T>::pi() : T(0);
    pdf_out = pdf_value;
    dir_out = dir;
}

// One segment of ray_color (src/ray_tracing.rs:19-59) in three pieces, so that the block-sorted
// render kernel can regroup lanes between the cast and the shading:
//   segment_cast   depth check, world.hit, hit record, emission / background  -> what happens next
//   shade_lambert  Lambertian bounce with mixture importance sampling
//   shade_specular Dielectric / FrostedDielectric / Metal / Isotropic
// path_segment() chains them for the sequential users (probe kernel, tests/hostsim).
enum : int { NEXT_LAMBERT = 0, NEXT_SPECULAR = 1, NEXT_DONE = 2 };

// After the call: NEXT_DONE -> st.L is final.  Otherwise st.ray.ori is the hit point, st.ray.dir still
// the INCOMING direction, rec.normal/u/v describe the hit and mat_index its material.
// The part of a segment after the world scan: background / emission / hit record / what happens next.
template <class T, uint32_t F>
__device__ inline int segment_post(const DScene<T>& sc, PathState<T>& st, bool any, const Hit<T>& h, HitRecord<T>& rec, uint32_t& mat_index) {
    if (!any) {  // :22, :59
        st.L = st.L + elemul(st.beta, V3<T>(sc.background));
        return NEXT_DONE;
    }
    mat_index = (has<F>(F_MEDIUM) && h.side == 7) ? sc.world[h.inst].phase_material : sc.prims[h.prim].material;
    const DMaterial<T>& m = sc.materials[mat_index];
    const uint32_t mkind = m.kind & 0xffu;
    finish_hit<T, F>(sc, st.ray, h, (m.kind & 0x100u) != 0, rec);
    st.ray.ori = rec.p;  // every scattered ray starts at the hit point and inherits ray.time (:29,34)
    if (mkind == 0) return NEXT_LAMBERT;
    if (mkind == 3) {  // DiffuseLight: src/materials.rs:150-159; scatter -> None -> return emitted
        // the unlit side returns Vec3::zero(): the recursion multiplies it by every bounce's weight on the way up,
        // so a NaN / inf throughput still turns the sample NaN (and it is retried) — keep that with beta * 0
        if (dot(st.ray.dir, rec.normal) < T(0)) st.L = st.L + elemul(st.beta, texture_value<T, F>(sc, m.texture, rec.u, rec.v, rec.p));
        else st.L = st.L + st.beta * T(0);
        return NEXT_DONE;
    }
    return NEXT_SPECULAR;
}
// Depth check of a new segment (:19-21).  false: the path ends here (st.L final).
template <class T> __device__ inline bool segment_begin(PathState<T>& st, int max_depth, uint32_t& n_casts) {
    if (st.seg >= max_depth) {  // depth <= 0 -> zero; the multiply keeps NaN/inf throughput NaN
        st.L = st.L + st.beta * T(0);
        return false;
    }
    st.seg++;
    n_casts++;
    return true;
}
template <class T, uint32_t F>
__device__ inline int segment_cast(const DScene<T>& sc, PathState<T>& st, int max_depth, Rng& rng, uint32_t& n_casts, HitRecord<T>& rec,
                                   uint32_t& mat_index) {
    if (!segment_begin<T>(st, max_depth, n_casts)) return NEXT_DONE;
    Hit<T> h;
    MediumXi mxi{&rng, 0.0};
    const bool any = world_hit<T, F>(sc, st.ray, T(0.001), Lim<T>::max(), mxi, h);
    RT_PCB();
    const int next = segment_post<T, F>(sc, st, any, h, rec, mat_index);
    RT_PCE(12);
    return next;
}

// Lambertian: src/materials.rs:36-50 + ray_tracing.rs:25-43 (emitted == 0)
template <class T, uint32_t F>
__device__ inline void shade_lambert(const DScene<T>& sc, PathState<T>& st, const HitRecord<T>& rec, uint32_t mat_index, Rng& rng) {
    const DMaterial<T>& m = sc.materials[mat_index];
    RT_PCB();
    V3<T> attenuation = texture_value<T, F>(sc, m.texture, rec.u, rec.v, rec.p);
    RT_PCE(13);
    V3<T> dir; T pdf_value, scattering_pdf;
    lambertian_sample<T, F>(sc, rec, rng, dir, pdf_value, scattering_pdf);
    RT_PCE(14);
    st.beta = elemul(st.beta, attenuation) * scattering_pdf / pdf_value;
    st.ray.dir = dir;
}

// Returns true when the path ends here (absorbed metal reflection, or a material kind outside F).
template <class T, uint32_t F>
__device__ inline bool shade_specular(const DScene<T>& sc, PathState<T>& st, const HitRecord<T>& rec, uint32_t mat_index, Rng& rng) {
    const DMaterial<T>& m = sc.materials[mat_index];
    const uint32_t mkind = m.kind & 0xffu;
    const V3<T> rdir = st.ray.dir;
    if (has<F>(F_DIELECTRIC) && (mkind == 2 || mkind == 5)) {  // Dielectric / FrostedDielectric: src/materials.rs:81-105, 114-142
        T etai_over_etat;
        V3<T> real_normal;
        if (dot(rdir, rec.normal) > T(0)) { etai_over_etat = m.ref_idx; real_normal = -rec.normal; }
        else { etai_over_etat = T(1) / m.ref_idx; real_normal = rec.normal; }
        V3<T> ud = unit(rdir);
        T cos_theta = rmin(dot(-ud, real_normal), T(1));
        T sin_theta = sqrt_r(T(1) - cos_theta * cos_theta);
        V3<T> dir;
        if (etai_over_etat * sin_theta <= T(1) && rng.gen<T>() > schlick(cos_theta, m.ref_idx)) {
            dir = refract(ud, real_normal, etai_over_etat);
            if (has<F>(F_FROSTED) && mkind == 5) dir = dir + random_in_unit_sphere<T>(rng) * m.fuzz;
        } else {
            dir = reflect(ud, rec.normal);
        }
        st.ray.dir = dir;  // attenuation = ones
        return false;
    }
    if (has<F>(F_METAL) && mkind == 1) {  // Metal: src/materials.rs:58-73
        V3<T> reflected = reflect(unit(rdir), rec.normal);
        V3<T> dir = reflected + random_in_unit_sphere<T>(rng) * m.fuzz;
        if (dot(dir, rec.normal) > T(0)) {
            st.beta = elemul(st.beta, V3<T>(m.albedo));
            st.ray.dir = dir;
            return false;
        }
        st.L = st.L + st.beta * T(0);  // absorbed: returns emitted == 0, weighted by the bounces above (NaN stays NaN)
        return true;
    }
    if (has<F>(F_MEDIUM) && mkind == 4) {  // Isotropic: src/materials.rs:167-172
        V3<T> attenuation = texture_value<T, F>(sc, m.texture, rec.u, rec.v, rec.p);
        st.beta = elemul(st.beta, attenuation);
        st.ray.dir = random_in_unit_sphere<T>(rng);
        return false;
    }
    return true;  // unreachable for a validated scene whose feature mask is covered by F
}

// Returns true when the path has terminated (st.L is final).
template <class T, uint32_t F>
__device__ inline bool path_segment(const DScene<T>& sc, PathState<T>& st, int max_depth, Rng& rng, uint32_t& n_casts) {
    HitRecord<T> rec;
    uint32_t mat_index = 0;
    RT_PCB();
    const int next = segment_cast<T, F>(sc, st, max_depth, rng, n_casts, rec, mat_index);
    RT_PCE(1);  // includes the traversal slots 2, 3
    if (next == NEXT_DONE) return true;
    if (next == NEXT_LAMBERT) { shade_lambert<T, F>(sc, st, rec, mat_index, rng); RT_PCE(4); return false; }
    const bool done = shade_specular<T, F>(sc, st, rec, mat_index, rng);
    RT_PCE(5);
    return done;
}

// Fresh camera path for sample (u,v): src/ray_tracing.rs:103
// The camera (24 numbers, 48 scalar registers in f64) is read where a path starts, from the kernel-argument segment (the scene is the
// FIRST argument of every kernel), behind an opaque copy of its address: as a member of the by-value scene the compiler loaded it once
// at kernel entry and carried it — in scalar registers and, beyond their number, in spill lanes of vector registers — through the
// whole sample loop (render_kernel_sorted<f64, cornell>: 98 -> 66 spilled scalars... see DESIGN §4).
template <class T> __device__ inline DCamera<T> scene_camera(const DScene<T>& sc, uint32_t varying) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_HOST_SIM) && !defined(RT_CAMERA_IN_REGISTERS)
    typedef const __attribute__((address_space(4))) char* kernarg_bytes;
    typedef const __attribute__((address_space(4))) DCamera<T>* kernarg_camera;
    kernarg_bytes p = (kernarg_bytes)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(DScene<T>, cam);
    // not a loop invariant the compiler may hoist: the address "depends" on a number that changes from path to path (a plain asm, not
    // a volatile one, which would count as a write to memory and cost the kernel its scalar scene reads)
    const int vary = __builtin_amdgcn_readfirstlane((int)varying);
    asm("" : "+s"(p) : "s"(vary));
    kernarg_camera c = (kernarg_camera)p;
    DCamera<T> cam;
    for (int k = 0; k < 3; k++) {
        cam.origin[k] = c->origin[k]; cam.llc[k] = c->llc[k]; cam.horizontal[k] = c->horizontal[k]; cam.vertical[k] = c->vertical[k];
        cam.u[k] = c->u[k]; cam.v[k] = c->v[k]; cam.w[k] = c->w[k];
    }
    cam.lens_radius = c->lens_radius; cam.t1 = c->t1; cam.t2 = c->t2;
    return cam;
#else
    return sc.cam;
#endif
}
template <class T> __device__ inline void start_path(const DScene<T>& sc, PathState<T>& st, T u, T v, Rng& rng) {
    st.ray = get_ray(scene_camera(sc, rng.s0), u, v, rng);
    st.beta = V3<T>(T(1), T(1), T(1));
    st.L = V3<T>(T(0), T(0), T(0));
    st.seg = 0;
}
// src/ray_tracing.rs:100-101: seeds the sample's generator and draws the pixel jitter
template <class T> __device__ inline void sample_uv(const RenderArgs& a, uint32_t x, uint32_t y, uint32_t s, Rng& rng, T& u, T& v) {
    rng.init(a.seed, y * a.width + x, s);
    u = ((T)x + rng.gen<T>()) / ((T)a.width - T(1));
    v = ((T)y + rng.gen<T>()) / ((T)a.height - T(1));
}

// Radiance of ONE accepted sample, NaN-retry loop included (src/ray_tracing.rs:100-110): the
// sequential form used by the one-sample-per-lane probe kernel and by tests/hostsim.
template <class T, uint32_t F>
__device__ inline V3<T> radiance_of_sample(const DScene<T>& sc, const RenderArgs& a, uint32_t x, uint32_t y, uint32_t s, uint32_t& n_casts) {
    Rng rng;
    T u, v;
    sample_uv<T>(a, x, y, s, rng, u, v);
    uint32_t attempts = 0;
    for (;;) {
        PathState<T> st;
        start_path(sc, st, u, v, rng);
        while (!path_segment<T, F>(sc, st, a.max_depth, rng, n_casts)) {}
        if (!is_nan3(st.L)) return st.L;
        if (++attempts > a.max_nan_retries) return V3<T>(T(0), T(0), T(0));
    }
}

// Deterministic parity probe: closest hit of the jitter-free, lens-free camera ray of pixel (x,y).
template <class T, uint32_t F>
__device__ inline bool probe_primary(const DScene<T>& sc, uint32_t width, uint32_t height, uint32_t x, uint32_t y, T time, double medium_xi_v,
                                     int32_t& id, double& t_out) {
    T u = ((T)x + T(0)) / ((T)width - T(1));
    T v = ((T)y + T(0)) / ((T)height - T(1));
    Ray<T> ray = camera_ray(sc.cam, u, v, V3<T>(T(0), T(0), T(0)), time);
    Hit<T> h;
    MediumXi mxi{nullptr, medium_xi_v};
    if (world_hit<T, F>(sc, ray, T(0.001), Lim<T>::max(), mxi, h)) {
        id = h.prim * 8 + h.side;
        t_out = (double)h.t;
        return true;
    }
    id = -1;
    t_out = -1.0;
    return false;
}

}  // namespace rt
