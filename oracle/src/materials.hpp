// ORACLE — TEST INFRASTRUCTURE ONLY.
// CPU f64 restatement of reference src/pdf.rs (ONB, CosinePDF, ObjectPDF, MixturePDF) and
// src/materials.rs (schlick, ScatterRecord, Lambertian, Metal, Dielectric, FrostedDielectric,
// DiffuseLight, Isotropic).
#pragma once
#include "objects.hpp"
#include "texture.hpp"

namespace orc {

// pdf.rs:5-25
struct ONB {
    Vec3 u, v, w;
    static ONB build_from_w(Vec3 w_in) {
        ORC_CNT(g_cnt.flops += 45);
        Vec3 w = w_in.unit();
        Vec3 a = std::fabs(w.x) > 0.9 ? Vec3(0.0, 1.0, 0.0) : Vec3(1.0, 0.0, 0.0);
        Vec3 v = Vec3::cross(w, a).unit();
        Vec3 u = Vec3::cross(w, v).unit();
        return ONB{u, v, w};
    }
    Vec3 local(Vec3 a) const { return u * a.x + v * a.y + w * a.z; }
};

// objects.rs:138-143 (Sphere::random)
inline Vec3 sphere_random_dir(Vec3 center, double radius, Vec3 origin, Rng& rng) {
    Vec3 direction = center - origin;
    double distance_squared = direction.squared_length();
    ONB uvw = ONB::build_from_w(direction);
    return uvw.local(random_to_sphere(radius, distance_squared, rng));
}

// pdf.rs:27-30
struct PDF {
    virtual ~PDF() {}
    virtual double value(Vec3 direction) const = 0;
    virtual Vec3 generate(Rng& rng) const = 0;
};
// pdf.rs:32-54
struct CosinePDF : PDF {
    ONB uvw;
    explicit CosinePDF(Vec3 w) : uvw(ONB::build_from_w(w)) {}
    double value(Vec3 direction) const override {
        ORC_CNT(g_cnt.flops += 15);
        double cosine = direction.unit() * uvw.w;
        return cosine > 0.0 ? cosine / PI : 0.0;
    }
    Vec3 generate(Rng& rng) const override {
        ORC_CNT(g_cnt.flops += 10);
        return uvw.local(random_cosine_direction(rng));
    }
};
// pdf.rs:56-72
struct ObjectPDF : PDF {
    const Object* object;
    Vec3 origin;
    ObjectPDF(const Object* o, Vec3 org) : object(o), origin(org) {}
    double value(Vec3 direction) const override { return object->pdf_value(origin, direction); }
    Vec3 generate(Rng& rng) const override { return object->random(origin, rng); }
};
// pdf.rs:74-94
struct MixturePDF : PDF {
    const PDF *p1, *p2;
    MixturePDF(const PDF* a, const PDF* b) : p1(a), p2(b) {}
    double value(Vec3 direction) const override { return 0.5 * p1->value(direction) + 0.5 * p2->value(direction); }
    Vec3 generate(Rng& rng) const override {
        if (rng.gen() < 0.5) return p1->generate(rng);
        return p2->generate(rng);
    }
};

// materials.rs:6-9
inline double schlick(double cosine, double ref_idx) {
    double r0 = ((1.0 - ref_idx) / (1.0 + ref_idx)) * ((1.0 - ref_idx) / (1.0 + ref_idx));
    double x = 1.0 - cosine;
    // f64::powi(5): llvm.powi lowers to repeated multiplication x*x -> ^2, ^4, * x
    double x2 = x * x;
    return r0 + (1.0 - r0) * (x2 * x2 * x);
}

// materials.rs:11-20
struct ScatterRecord {
    enum Kind { NONE, SPECULAR, DIFFUSE } kind = NONE;
    Vec3 attenuation;
    Ray specular_ray;
    Vec3 pdf_w;  // CosinePDF is rebuilt from the normal by the caller
};

enum MatKind { M_LAMBERTIAN = 0, M_METAL, M_DIELECTRIC, M_DIFFUSE_LIGHT, M_ISOTROPIC, M_FROSTED };

// materials.rs:21-29
struct Material {
    virtual ~Material() {}
    virtual ScatterRecord scatter(const Ray& r_in, const HitRecord& rec, Rng& rng) const = 0;
    virtual Vec3 emitted(const Ray&, const HitRecord&, double, double, Vec3) const { return Vec3::zero(); }
    virtual double scattering_pdf(const Ray&, const HitRecord&, const Ray&) const { return 0.0; }
    virtual int kind() const = 0;
};

// materials.rs:31-50
struct Lambertian : Material {
    TexPtr albedo;
    explicit Lambertian(TexPtr a) : albedo(std::move(a)) {}
    ScatterRecord scatter(const Ray&, const HitRecord& rec, Rng&) const override {
        ScatterRecord s;
        s.kind = ScatterRecord::DIFFUSE;
        s.attenuation = albedo->value(rec.u, rec.v, rec.p);
        s.pdf_w = rec.normal;
        return s;
    }
    double scattering_pdf(const Ray&, const HitRecord& rec, const Ray& scattered) const override {
        ORC_CNT(g_cnt.flops += 15);
        double cosine = rec.normal * scattered.dir.unit();
        return cosine > 0.0 ? cosine / PI : 0.0;
    }
    int kind() const override { return M_LAMBERTIAN; }
};

// materials.rs:52-74
struct Metal : Material {
    Vec3 albedo;
    double fuzz;
    Metal(Vec3 a, double f) : albedo(a), fuzz(f) {}
    ScatterRecord scatter(const Ray& r_in, const HitRecord& rec, Rng& rng) const override {
        ORC_CNT(g_cnt.flops += 30);
        Vec3 reflected = reflect(r_in.dir.unit(), rec.normal);
        Ray scattered(rec.p, reflected + random_in_unit_sphere(rng) * fuzz, r_in.time);
        ScatterRecord s;
        if (scattered.dir * rec.normal > 0.0) {
            s.kind = ScatterRecord::SPECULAR;
            s.attenuation = albedo;
            s.specular_ray = scattered;
        }
        return s;
    }
    int kind() const override { return M_METAL; }
};

// materials.rs:76-106
struct Dielectric : Material {
    double ref_idx;
    explicit Dielectric(double r) : ref_idx(r) {}
    ScatterRecord scatter(const Ray& r_in, const HitRecord& rec, Rng& rng) const override {
        ORC_CNT(g_cnt.flops += 60);
        double etai_over_etat;
        Vec3 real_normal;
        if (r_in.dir * rec.normal > 0.0) {
            etai_over_etat = ref_idx;
            real_normal = -rec.normal;
        } else {
            etai_over_etat = 1.0 / ref_idx;
            real_normal = rec.normal;
        }
        double cos_theta = std::fmin((-r_in.dir.unit()) * real_normal, 1.0);
        double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
        ScatterRecord s;
        s.kind = ScatterRecord::SPECULAR;
        s.attenuation = Vec3::ones();
        // `&&` short-circuit: xi is drawn only when refraction is possible (materials.rs:93)
        if (etai_over_etat * sin_theta <= 1.0 && rng.gen() > schlick(cos_theta, ref_idx)) {
            Vec3 refracted = refract(r_in.dir.unit(), real_normal, etai_over_etat);
            s.specular_ray = Ray(rec.p, refracted, r_in.time);
            return s;
        }
        s.specular_ray = Ray(rec.p, reflect(r_in.dir.unit(), rec.normal), r_in.time);
        return s;
    }
    int kind() const override { return M_DIELECTRIC; }
};

// materials.rs:108-143
struct FrostedDielectric : Material {
    double ref_idx, fuzz;
    FrostedDielectric(double r, double f) : ref_idx(r), fuzz(f) {}
    ScatterRecord scatter(const Ray& r_in, const HitRecord& rec, Rng& rng) const override {
        ORC_CNT(g_cnt.flops += 70);
        double etai_over_etat;
        Vec3 real_normal;
        if (r_in.dir * rec.normal > 0.0) {
            etai_over_etat = ref_idx;
            real_normal = -rec.normal;
        } else {
            etai_over_etat = 1.0 / ref_idx;
            real_normal = rec.normal;
        }
        double cos_theta = std::fmin((-r_in.dir.unit()) * real_normal, 1.0);
        double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
        ScatterRecord s;
        s.kind = ScatterRecord::SPECULAR;
        s.attenuation = Vec3::ones();
        if (etai_over_etat * sin_theta <= 1.0 && rng.gen() > schlick(cos_theta, ref_idx)) {
            Vec3 refracted = refract(r_in.dir.unit(), real_normal, etai_over_etat);
            s.specular_ray = Ray(rec.p, refracted + random_in_unit_sphere(rng) * fuzz, r_in.time);
            return s;
        }
        s.specular_ray = Ray(rec.p, reflect(r_in.dir.unit(), rec.normal), r_in.time);
        return s;
    }
    int kind() const override { return M_FROSTED; }
};

// materials.rs:145-160
struct DiffuseLight : Material {
    TexPtr emit;
    explicit DiffuseLight(TexPtr e) : emit(std::move(e)) {}
    ScatterRecord scatter(const Ray&, const HitRecord&, Rng&) const override { return ScatterRecord(); }
    Vec3 emitted(const Ray& r_in, const HitRecord& rec, double u, double v, Vec3 p) const override {
        if (r_in.dir * rec.normal < 0.0) return emit->value(u, v, p);
        return Vec3::zero();
    }
    int kind() const override { return M_DIFFUSE_LIGHT; }
};

// materials.rs:162-173
struct Isotropic : Material {
    TexPtr albedo;
    explicit Isotropic(TexPtr a) : albedo(std::move(a)) {}
    ScatterRecord scatter(const Ray& r_in, const HitRecord& rec, Rng& rng) const override {
        ScatterRecord s;
        s.kind = ScatterRecord::SPECULAR;
        s.attenuation = albedo->value(rec.u, rec.v, rec.p);
        s.specular_ray = Ray(rec.p, random_in_unit_sphere(rng), r_in.time);
        return s;
    }
    int kind() const override { return M_ISOTROPIC; }
};

}  // namespace orc
