// rt_path.hpp — light importance sampling, material scatter and the iterative path loop.
// Restates src/pdf.rs, src/materials.rs and src/ray_tracing.rs:11-60,95-111 for the device.
#pragma once
#include "rt_device.hpp"

namespace rt {

// ------------------------------------------------------------------------------------------
// light list: Object::pdf_value / Object::random through ObjectList, Translate, RotateY
// ------------------------------------------------------------------------------------------
// src/objects.rs:251-261, 305-315, 359-369
template <class T>
__device__ inline T rect_pdf(int axis, T a1, T a2, T b1, T b2, T k, T face, V3<T> o, V3<T> v) {
    T t;
    if (rect_hit<T>(axis, a1, a2, b1, b2, k, o, v, T(0.001), Lim<T>::max(), t)) {
        V3<T> normal = (axis == 0 ? V3<T>(T(0), T(0), T(1)) : axis == 1 ? V3<T>(T(0), T(1), T(0)) : V3<T>(T(1), T(0), T(0))) * face;
        T area = (a2 - a1) * (b2 - b1);
        T distance_squared = t * t * sqlen(v);
        T cosine = rabs(dot(v, normal)) / length(v);
        return distance_squared / (cosine * area);
    }
    return T(0);
}
template <class T> __device__ inline V3<T> rect_random(int axis, T a1, T a2, T b1, T b2, T k, V3<T> origin, Rng& rng) {
    T a = rng.gen_range(a1, a2);
    T b = rng.gen_range(b1, b2);
    V3<T> random_point = axis == 0 ? V3<T>(a, b, k) : axis == 1 ? V3<T>(a, k, b) : V3<T>(k, a, b);
    return random_point - origin;
}

template <class T> __device__ inline T prim_pdf_value(const DPrim<T>& p, V3<T> o, V3<T> v) {
    switch (p.kind) {
        case 0: {  // src/objects.rs:126-137
            T t;
            V3<T> center(p.d);
            T radius = p.d[3];
            if (sphere_hit(center, radius, o, v, T(0.001), Lim<T>::max(), t)) {
                T cos_theta_max = sqrt_r(T(1) - radius * radius / sqlen(center - o));
                T solid_angle = T(2) * Consts<T>::pi() * (T(1) - cos_theta_max);
                return T(1) / solid_angle;
            }
            return T(0);
        }
        case 1: return T(0);  // MovingSphere keeps the trait default (src/objects.rs:22-24)
        case 2: case 3: case 4:
            return rect_pdf<T>((int)p.kind - 2, p.d[0], p.d[1], p.d[2], p.d[3], p.d[4], p.d[5], o, v);
        default: {  // src/objects.rs:492-500
            T weight = T(1) / T(6);
            T ans = T(0);
#pragma unroll
            for (int i = 0; i < 6; i++) {
                int axis; T a1, a2, b1, b2, k, face;
                cuboid_side(p, i, axis, a1, a2, b1, b2, k, face);
                T term = weight * rect_pdf<T>(axis, a1, a2, b1, b2, k, face, o, v);
                ans = i == 0 ? term : ans + term;
            }
            return ans;
        }
    }
}
template <class T> __device__ inline V3<T> prim_random(const DPrim<T>& p, V3<T> origin, Rng& rng) {
    switch (p.kind) {
        case 0: {  // src/objects.rs:138-143
            V3<T> center(p.d);
            V3<T> direction = center - origin;
            T distance_squared = sqlen(direction);
            ONB<T> uvw = onb_from_w(direction);
            return onb_local(uvw, random_to_sphere(p.d[3], distance_squared, rng));
        }
        case 1: return V3<T>(T(1), T(0), T(0));
        case 2: case 3: case 4:
            return rect_random<T>((int)p.kind - 2, p.d[0], p.d[1], p.d[2], p.d[3], p.d[4], origin, rng);
        default: {  // src/objects.rs:501-511
            int i = (int)rng.gen_index(6);
            int axis; T a1, a2, b1, b2, k, face;
            cuboid_side(p, i, axis, a1, a2, b1, b2, k, face);
            return rect_random<T>(axis, a1, a2, b1, b2, k, origin, rng);
        }
    }
}
// src/objects.rs:68-75 over the light list
template <class T> __device__ inline T lights_pdf_value(const DScene<T>& sc, V3<T> origin, V3<T> v) {
    T weight = T(1) / (T)sc.n_lights;
    T ans = T(0);
    for (uint32_t i = 0; i < sc.n_lights; i++) {
        const DInstance<T>& in = sc.lights[i];
        T pv = T(0);
        if (!in.is_medium && in.geom_kind == 0) {  // ConstantMedium / BvhNode keep the default 0
            V3<T> o = origin, d = v;
            xform_ray(sc, in, o, d);
            pv = prim_pdf_value(sc.prims[in.geom_index], o, d);
        }
        ans += weight * pv;
    }
    return ans;
}
// src/objects.rs:76-78
template <class T> __device__ inline V3<T> lights_random(const DScene<T>& sc, V3<T> origin, Rng& rng) {
    uint32_t i = rng.gen_index(sc.n_lights);
    const DInstance<T>& in = sc.lights[i];
    if (in.is_medium) return V3<T>(T(1), T(0), T(0));
    V3<T> o = origin, dummy(T(0), T(0), T(0));
    xform_ray(sc, in, o, dummy);
    V3<T> dir = in.geom_kind == 0 ? prim_random(sc.prims[in.geom_index], o, rng) : V3<T>(T(1), T(0), T(0));
    return xform_dir_back(sc, in, dir);
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

// One segment of ray_color (src/ray_tracing.rs:19-59): cast, emit, scatter, pick the next
// direction.  Returns true when the path has terminated (st.L is final).
template <class T>
__device__ inline bool path_segment(const DScene<T>& sc, PathState<T>& st, int max_depth, Rng& rng, uint32_t& n_casts,
                                    uint32_t& n_tests) {
    if (st.seg >= max_depth) {  // depth <= 0 -> zero (:19-21); the multiply keeps NaN/inf throughput NaN
        st.L = st.L + st.beta * T(0);
        return true;
    }
    st.seg++;
    n_casts++;
    Hit<T> h;
    MediumXi mxi{&rng, 0.0};
    if (!world_hit(sc, st.ray, T(0.001), Lim<T>::max(), mxi, h, n_tests)) {  // :22, :59
        st.L = st.L + elemul(st.beta, V3<T>(sc.background));
        return true;
    }
    uint32_t mat_index = h.side == 7 ? sc.world[h.inst].phase_material : sc.prims[h.prim].material;
    const DMaterial<T>& m = sc.materials[mat_index];
    HitRecord<T> rec;
    finish_hit(sc, st.ray, h, (m.kind & 0x100u) != 0, rec);
    const V3<T> rdir = st.ray.dir;
    switch (m.kind & 0xffu) {
        case 0: {  // Lambertian: src/materials.rs:36-50 + ray_tracing.rs:25-43
            V3<T> attenuation = texture_value(sc, m.texture, rec.u, rec.v, rec.p);
            ONB<T> uvw = onb_from_w(rec.normal);
            V3<T> dir;
            T pdf_value;
            if (sc.n_lights > 0) {  // MixturePDF(ObjectPDF(lights), CosinePDF): src/pdf.rs:84-93
                if (rng.gen<T>() < T(0.5)) dir = lights_random(sc, rec.p, rng);
                else dir = onb_local(uvw, random_cosine_direction<T>(rng));
                T cosine = dot(unit(dir), uvw.w);
                T cos_pdf = cosine > T(0) ? cosine / Consts<T>::pi() : T(0);
                pdf_value = T(0.5) * lights_pdf_value(sc, rec.p, dir) + T(0.5) * cos_pdf;
            } else {
                dir = onb_local(uvw, random_cosine_direction<T>(rng));
                T cosine = dot(unit(dir), uvw.w);
                pdf_value = cosine > T(0) ? cosine / Consts<T>::pi() : T(0);
            }
            T sc_cos = dot(rec.normal, unit(dir));
            T scattering_pdf = sc_cos > T(0) ? sc_cos / Consts<T>::pi() : T(0);
            st.beta = elemul(st.beta, attenuation) * scattering_pdf / pdf_value;  // emitted == 0
            st.ray.ori = rec.p;
            st.ray.dir = dir;
            return false;
        }
        case 1: {  // Metal: src/materials.rs:58-73
            V3<T> reflected = reflect(unit(rdir), rec.normal);
            V3<T> dir = reflected + random_in_unit_sphere<T>(rng) * m.fuzz;
            if (dot(dir, rec.normal) > T(0)) {
                st.beta = elemul(st.beta, V3<T>(m.albedo));
                st.ray.ori = rec.p;
                st.ray.dir = dir;
                return false;
            }
            return true;  // absorbed: returns emitted == 0
        }
        case 2: case 5: {  // Dielectric / FrostedDielectric: src/materials.rs:81-105, 114-142
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
                if ((m.kind & 0xffu) == 5) dir = dir + random_in_unit_sphere<T>(rng) * m.fuzz;
            } else {
                dir = reflect(ud, rec.normal);
            }
            st.ray.ori = rec.p;  // attenuation = ones
            st.ray.dir = dir;
            return false;
        }
        case 3: {  // DiffuseLight: src/materials.rs:150-159; scatter -> None -> return emitted
            if (dot(rdir, rec.normal) < T(0)) st.L = st.L + elemul(st.beta, texture_value(sc, m.texture, rec.u, rec.v, rec.p));
            return true;
        }
        default: {  // Isotropic: src/materials.rs:167-172
            V3<T> attenuation = texture_value(sc, m.texture, rec.u, rec.v, rec.p);
            st.beta = elemul(st.beta, attenuation);
            st.ray.ori = rec.p;
            st.ray.dir = random_in_unit_sphere<T>(rng);
            return false;
        }
    }
}

// ------------------------------------------------------------------------------------------
// render kernel: src/ray_tracing.rs:95-111.  One lane owns one pixel; the sample loop and the
// path loop are flattened into ONE loop so a lane whose path ended starts its next sample at once
// instead of idling until the longest path of the wave ends (in-place path regeneration).
// Block = 256 threads = a 16x16 pixel tile; each wave covers an 8x8 sub-tile.
// ------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void render_kernel(DScene<T> sc, RenderArgs a, float* __restrict__ accum,
                                                     unsigned long long* __restrict__ stats) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t x = blockIdx.x * 16u + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t y = blockIdx.y * 16u + (wave >> 1) * 8u + (lane >> 3);
    const bool valid = x < a.width && y < a.height;
    const uint32_t pixel = y * a.width + x;
    uint32_t s = a.spp_begin;
    const uint32_t s_end = valid ? a.spp_begin + a.spp_count : a.spp_begin;
    double ax = 0.0, ay = 0.0, az = 0.0;
    uint32_t n_casts = 0, n_tests = 0, n_retry = 0, n_drop = 0, n_acc = 0;
    bool alive = false;
    PathState<T> st;
    T u = T(0), v = T(0);
    Rng rng;
    uint32_t attempts = 0;
    for (;;) {
        if (!alive) {
            if (s >= s_end) break;
            rng.init(a.seed, pixel, s);
            u = ((T)x + rng.gen<T>()) / ((T)a.width - T(1));   // :100
            v = ((T)y + rng.gen<T>()) / ((T)a.height - T(1));  // :101
            attempts = 0;
            st.ray = get_ray(sc.cam, u, v, rng);
            st.beta = V3<T>(T(1), T(1), T(1));
            st.L = V3<T>(T(0), T(0), T(0));
            st.seg = 0;
            alive = true;
        }
        if (path_segment(sc, st, a.max_depth, rng, n_casts, n_tests)) {
            if (is_nan3(st.L)) {  // :107 — retry the SAME (u,v) with a fresh camera ray
                n_retry++;
                attempts++;
                if (attempts > a.max_nan_retries) { n_drop++; s++; alive = false; }
                else {
                    st.ray = get_ray(sc.cam, u, v, rng);
                    st.beta = V3<T>(T(1), T(1), T(1));
                    st.L = V3<T>(T(0), T(0), T(0));
                    st.seg = 0;
                }
            } else {
                ax += (double)st.L.x; ay += (double)st.L.y; az += (double)st.L.z;
                n_acc++; s++; alive = false;
            }
        }
    }
    if (valid) {
        float* px = accum + (size_t)pixel * 3;
        px[0] += (float)ax; px[1] += (float)ay; px[2] += (float)az;
    }
    if (stats) {
        // wave-level reduction, one atomic per counter per wave
        unsigned long long c0 = n_acc, c1 = n_casts, c2 = n_retry, c3 = n_drop;
        for (int off = 32; off > 0; off >>= 1) {
            c0 += __shfl_down(c0, off); c1 += __shfl_down(c1, off);
            c2 += __shfl_down(c2, off); c3 += __shfl_down(c3, off);
        }
        if (lane == 0) {
            atomicAdd(&stats[0], c0); atomicAdd(&stats[1], c1);
            if (c2) atomicAdd(&stats[2], c2);
            if (c3) atomicAdd(&stats[3], c3);
        }
    }
    (void)n_tests;
}

// Deterministic parity probe: closest hit of jitter-free, lens-free camera rays.
template <class T>
__global__ __launch_bounds__(256) void trace_primary_kernel(DScene<T> sc, uint32_t width, uint32_t height, T time, double medium_xi_v,
                                                            int32_t* __restrict__ ids, double* __restrict__ ts) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= width * height) return;
    const uint32_t x = idx % width, y = idx / width;
    T u = ((T)x + T(0)) / ((T)width - T(1));
    T v = ((T)y + T(0)) / ((T)height - T(1));
    Ray<T> ray = camera_ray(sc.cam, u, v, V3<T>(T(0), T(0), T(0)), time);
    Hit<T> h;
    MediumXi mxi{nullptr, medium_xi_v};
    uint32_t n_tests = 0;
    if (world_hit(sc, ray, T(0.001), Lim<T>::max(), mxi, h, n_tests)) {
        ids[idx] = h.prim * 8 + h.side;
        ts[idx] = (double)h.t;
    } else {
        ids[idx] = -1;
        ts[idx] = -1.0;
    }
}

// One sample's radiance per thread (for sample-for-sample comparison with the oracle):
// out[i*3..] = radiance of sample `sample0 + i` of pixel (x,y).
template <class T>
__global__ void sample_pixel_kernel(DScene<T> sc, RenderArgs a, uint32_t x, uint32_t y, uint32_t n, double* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t pixel = y * a.width + x, s = a.spp_begin + i;
    Rng rng;
    rng.init(a.seed, pixel, s);
    T u = ((T)x + rng.gen<T>()) / ((T)a.width - T(1));
    T v = ((T)y + rng.gen<T>()) / ((T)a.height - T(1));
    uint32_t n_casts = 0, n_tests = 0, attempts = 0;
    V3<T> result(T(0), T(0), T(0));
    for (;;) {
        PathState<T> st;
        st.ray = get_ray(sc.cam, u, v, rng);
        st.beta = V3<T>(T(1), T(1), T(1));
        st.L = V3<T>(T(0), T(0), T(0));
        st.seg = 0;
        while (!path_segment(sc, st, a.max_depth, rng, n_casts, n_tests)) {}
        if (!is_nan3(st.L)) { result = st.L; break; }
        if (++attempts > a.max_nan_retries) break;
    }
    out[i * 3] = (double)result.x; out[i * 3 + 1] = (double)result.y; out[i * 3 + 2] = (double)result.z;
}

// src/ray_tracing.rs:112-117,126-128
__global__ inline void resolve_kernel(const float* __restrict__ accum, uint32_t width, uint32_t height, uint32_t spp,
                                      uint8_t* __restrict__ rgb) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= width * height * 3) return;
    const uint32_t c = idx % 3, pix = idx / 3, x = pix % width, y = pix / width;
    double col = (double)accum[idx] / (double)spp;
    double g = ::sqrt(col);
    g = g < 0.0 ? 0.0 : (g > 0.99999 ? 0.99999 : g);
    double q = g * 256.0;
    uint8_t b = (q != q) ? 0 : (q >= 255.0 ? 255 : (q <= 0.0 ? 0 : (uint8_t)q));
    rgb[((size_t)(height - 1 - y) * width + x) * 3 + c] = b;
}

}  // namespace rt
