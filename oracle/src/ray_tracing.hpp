// ORACLE — TEST INFRASTRUCTURE ONLY.
// CPU f64 restatement of reference src/camera.rs (Camera::new/get_ray) and src/ray_tracing.rs
// (ray_color :11-60, recursive; per-pixel sample loop :95-118 incl. NaN retry and tonemap;
// threads*16 x-strip jobs :83-88).
#pragma once
#include <thread>
#include <atomic>
#include <mutex>
#include "materials.hpp"

namespace orc {

// camera.rs:7-66
struct Camera {
    Vec3 origin, lower_left_corner, horizontal, vertical, u, v, w;
    double lens_radius, t1, t2;
    Camera() : lens_radius(0), t1(0), t2(1) {}
    Camera(Vec3 lookfrom, Vec3 lookat, Vec3 vup, double vfov, double aspect_ratio, double aperture,
           double focus_dist, double t1_, double t2_) {
        double theta = vfov * PI / 180.0;
        double viewport_height = 2.0 * std::tan(theta / 2.0);
        double viewport_width = viewport_height * aspect_ratio;
        w = (lookfrom - lookat).unit();
        u = Vec3::cross(vup, w).unit();
        v = Vec3::cross(w, u).unit();
        origin = lookfrom;
        horizontal = u * viewport_width * focus_dist;
        vertical = v * viewport_height * focus_dist;
        lower_left_corner = origin - horizontal / 2.0 - vertical / 2.0 - w * focus_dist;
        lens_radius = aperture / 2.0;
        t1 = t1_;
        t2 = t2_;
    }
    // camera.rs:57-65. Always consumes RNG for the lens disk, even when lens_radius == 0.
    Ray get_ray(double s, double t, Rng& rng) const {
        Vec3 rd = random_in_unit_disk(rng) * lens_radius;
        Vec3 offset = u * rd.x + v * rd.y;
        double time = rng.gen_range(t1, t2);
        return Ray(origin + offset, lower_left_corner + horizontal * s + vertical * t - origin - offset, time);
    }
    // Parity probe ray: lens offset 0 (random_in_unit_disk skipped), fixed time.
    Ray probe_ray(double s, double t, double time) const {
        Vec3 rd = Vec3(0.0, 0.0, 0.0) * lens_radius;
        Vec3 offset = u * rd.x + v * rd.y;
        return Ray(origin + offset, lower_left_corner + horizontal * s + vertical * t - origin - offset, time);
    }
};

struct Scene {
    std::shared_ptr<ObjectList> world;
    Vec3 background;
    Camera cam;
    std::shared_ptr<ObjectList> lights;  // null == None
    int n_prims = 0;
    void finalize() {
        int next = 0;
        world->assign_ids(next);
        n_prims = next;
    }
};

// ray_tracing.rs:11-60
inline Vec3 ray_color(const Ray& ray, const ObjectList& world, Vec3 background, const ObjectList* lights,
                      int depth, Rng& rng) {
    if (depth <= 0) {
        ORC_CNT(g_cnt.depth_exhausted++);
        return Vec3::zero();
    }
    HitRecord rec;
    ORC_CNT(g_cnt.casts++);
    if (world.hit(ray, 0.001, DBL_MAX, rec)) {
        Vec3 emitted = rec.mat_ptr->emitted(ray, rec, rec.u, rec.v, rec.p);
        ScatterRecord s = rec.mat_ptr->scatter(ray, rec, rng);
        ORC_CNT(g_cnt.scatters[rec.mat_ptr->kind()]++);
        if (s.kind == ScatterRecord::DIFFUSE) {
            CosinePDF pdf(s.pdf_w);
            Ray scattered;
            double pdf_value;
            if (lights) {
                ObjectPDF lights_pdf(lights, rec.p);
                MixturePDF p(&lights_pdf, &pdf);
                scattered = Ray(rec.p, p.generate(rng), ray.time);
                pdf_value = p.value(scattered.dir);
            } else {
                scattered = Ray(rec.p, pdf.generate(rng), ray.time);
                pdf_value = pdf.value(scattered.dir);
            }
            ORC_CNT(g_cnt.flops += 12);
            // emitted + elemul(att, L) * spdf / pdf   (ray_tracing.rs:38-43; the recursion runs
            // before scattering_pdf is evaluated: Rust evaluates operands left to right)
            Vec3 li = ray_color(scattered, world, background, lights, depth - 1, rng);
            return emitted + Vec3::elemul(s.attenuation, li) * rec.mat_ptr->scattering_pdf(ray, rec, scattered) / pdf_value;
        }
        if (s.kind == ScatterRecord::SPECULAR) {
            ORC_CNT(g_cnt.flops += 3);
            return Vec3::elemul(s.attenuation, ray_color(s.specular_ray, world, background, lights, depth - 1, rng));
        }
        return emitted;
    }
    ORC_CNT(g_cnt.misses++);
    return background;
}

// One accepted sample of pixel (x,y) (y counted upward as in ray_tracing.rs:97,127), sample
// index s: ray_tracing.rs:100-110.  The generator is keyed by (seed, y*W+x, s).
inline Vec3 sample_pixel(const Scene& sc, uint32_t W, uint32_t H, uint32_t x, uint32_t y, uint32_t s,
                         uint64_t seed, int max_depth, int max_retries = 64) {
    Rng rng = Rng::for_sample(seed, y * W + x, s);
    g_cast.rng = &rng;
    g_cast.fixed_xi = -1.0;
    double u = ((double)x + rng.gen()) / ((double)W - 1.0);
    double v = ((double)y + rng.gen()) / ((double)H - 1.0);
    ORC_CNT(g_cnt.paths++);
    for (int attempt = 0;; attempt++) {
        Ray ray = sc.cam.get_ray(u, v, rng);
        Vec3 rec = ray_color(ray, *sc.world, sc.background, sc.lights.get(), max_depth, rng);
        if (!(rec.x != rec.x || rec.y != rec.y || rec.z != rec.z)) return rec;
        ORC_CNT(g_cnt.nan_retries++);
        if (attempt >= max_retries) { ORC_CNT(g_cnt.dropped++); return Vec3::zero(); }  // device caps retries the same way
    }
}

// Render: accum[(y*W+x)*3+c] += sum over samples [s0, s0+n) of the accepted radiance (linear,
// un-normalised, f64).  Jobs are x-strips, n_jobs = threads*16, as ray_tracing.rs:83-88.
inline Counters render(const Scene& sc, uint32_t W, uint32_t H, uint32_t s0, uint32_t n, int max_depth,
                       uint64_t seed, int threads, double* accum, int max_retries = 64) {
    if (threads < 1) threads = 1;
    uint32_t n_jobs = (uint32_t)threads * 16;
    std::atomic<uint32_t> next_job(0);
    Counters total;
    std::mutex mu;
    auto worker = [&]() {
        g_cnt = Counters();
        for (;;) {
            uint32_t i = next_job.fetch_add(1);
            if (i >= n_jobs) break;
            uint32_t start_x = (uint32_t)((uint64_t)W * i / n_jobs);
            uint32_t end_x = (uint32_t)((uint64_t)W * (i + 1) / n_jobs);
            for (uint32_t x = start_x; x < end_x; x++)
                for (uint32_t y = 0; y < H; y++) {
                    Vec3 color(0.0, 0.0, 0.0);
                    for (uint32_t s = s0; s < s0 + n; s++) color += sample_pixel(sc, W, H, x, y, s, seed, max_depth, max_retries);
                    double* a = accum + ((size_t)y * W + x) * 3;
                    a[0] += color.x; a[1] += color.y; a[2] += color.z;
                }
        }
        std::lock_guard<std::mutex> g(mu);
        total.add(g_cnt);
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) pool.emplace_back(worker);
    for (auto& t : pool) t.join();
    return total;
}

// Deterministic parity probe: jitter-free grid u=x/(W-1), v=y/(H-1), lens
// offset 0, fixed time, every ConstantMedium uses xi = medium_xi; closest hit over the world with
// (0.001, f64::MAX) as ray_tracing.rs:22.  ids[y*W+x] = hit_id or -1, ts = t or -1.
inline void trace_primary(const Scene& sc, uint32_t W, uint32_t H, double time, double medium_xi,
                          int32_t* ids, double* ts) {
    g_cast.rng = nullptr;
    g_cast.fixed_xi = medium_xi;
    for (uint32_t y = 0; y < H; y++)
        for (uint32_t x = 0; x < W; x++) {
            double u = ((double)x + 0.0) / ((double)W - 1.0);
            double v = ((double)y + 0.0) / ((double)H - 1.0);
            Ray ray = sc.cam.probe_ray(u, v, time);
            HitRecord rec;
            bool h = sc.world->hit(ray, 0.001, DBL_MAX, rec);
            ids[(size_t)y * W + x] = h ? rec.hit_id : -1;
            ts[(size_t)y * W + x] = h ? rec.t : -1.0;
        }
    g_cast.fixed_xi = -1.0;
}

// ray_tracing.rs:112-117,126-128: mean, sqrt, clamp[0,0.99999], *256 `as u8` (saturating,
// NaN -> 0), written to row H-1-y.
inline void resolve(const double* accum, uint32_t W, uint32_t H, uint32_t spp, uint8_t* rgb) {
    for (uint32_t y = 0; y < H; y++)
        for (uint32_t x = 0; x < W; x++)
            for (int c = 0; c < 3; c++) {
                double col = accum[((size_t)y * W + x) * 3 + c] / (double)spp;
                double g = std::sqrt(col);
                g = g < 0.0 ? 0.0 : (g > 0.99999 ? 0.99999 : g);
                double q = g * 256.0;
                uint8_t b = (q != q) ? 0 : (q >= 255.0 ? 255 : (q <= 0.0 ? 0 : (uint8_t)q));
                rgb[((size_t)(H - 1 - y) * W + x) * 3 + c] = b;
            }
}

}  // namespace orc
