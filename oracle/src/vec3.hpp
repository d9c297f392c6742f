// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked, imported or executed by the product path.
// CPU f64 restatement of reference src/vec3.rs:4-257 (Vec3 value type, sampling helpers, optics).
// Arithmetic is written term-by-term in the reference's evaluation order and the library is
// compiled with -ffp-contract=off, so each expression rounds exactly like the Rust f64 code.
#pragma once
#include <cmath>
#include "rng.hpp"

namespace orc {

struct Vec3 {
    double x, y, z;
    Vec3() : x(0), y(0), z(0) {}
    Vec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
    static Vec3 ones() { return Vec3(1.0, 1.0, 1.0); }  // vec3.rs:16
    static Vec3 zero() { return Vec3(0.0, 0.0, 0.0); }  // vec3.rs:20
    double squared_length() const { return x * x + y * y + z * z; }      // vec3.rs:24
    double length() const { return std::sqrt(x * x + y * y + z * z); }   // vec3.rs:27
    // vec3.rs:30-36. The reference panics on a zero vector; the oracle lets the division produce
    // NaN/inf instead (the sample is then discarded by the NaN retry of ray_tracing.rs:107) and
    // counts the event, so a parity run can assert it never happened.
    Vec3 unit() const;
    static Vec3 elemul(Vec3 a, Vec3 b) { return Vec3(a.x * b.x, a.y * b.y, a.z * b.z); }  // :37
    static Vec3 cross(Vec3 a, Vec3 b) {                                                    // :40
        return Vec3(a.y * b.z - b.y * a.z, b.x * a.z - a.x * b.z, a.x * b.y - b.x * a.y);
    }
    // vec3.rs:47-53
    static Vec3 random(double lo, double hi, Rng& rng) {
        double a = rng.gen_range(lo, hi);
        double b = rng.gen_range(lo, hi);
        double c = rng.gen_range(lo, hi);
        return Vec3(a, b, c);
    }
    bool operator==(const Vec3& o) const { return x == o.x && y == o.y && z == o.z; }
};

inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator+(Vec3 a, double b) { return Vec3(a.x + b, a.y + b, a.z + b); }
inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 operator-(Vec3 a, double b) { return Vec3(a.x - b, a.y - b, a.z - b); }
inline Vec3 operator-(Vec3 a) { return Vec3(-a.x, -a.y, -a.z); }
// vec3.rs:144-150: `Vec3 * Vec3` is the DOT product.
inline double operator*(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 operator*(Vec3 a, double b) { return Vec3(a.x * b, a.y * b, a.z * b); }
inline Vec3 operator/(Vec3 a, double b) { return Vec3(a.x / b, a.y / b, a.z / b); }
inline Vec3& operator+=(Vec3& a, Vec3 b) { a = a + b; return a; }
inline Vec3& operator+=(Vec3& a, double b) { a = a + b; return a; }
inline Vec3& operator-=(Vec3& a, Vec3 b) { a = a - b; return a; }
inline Vec3& operator-=(Vec3& a, double b) { a = a - b; return a; }
inline Vec3& operator*=(Vec3& a, double b) { a = a * b; return a; }
inline Vec3& operator/=(Vec3& a, double b) { a = a / b; return a; }

extern thread_local unsigned long long g_unit_zero_events;
inline Vec3 Vec3::unit() const {
    double len = length();
    if (len == 0.0) g_unit_zero_events++;
    return Vec3(x / len, y / len, z / len);
}

constexpr double PI = 3.14159265358979323846;  // std::f64::consts::PI

// vec3.rs:208-216
inline Vec3 random_in_unit_sphere(Rng& rng) {
    for (;;) {
        double a = rng.gen();
        double b = rng.gen();
        double c = rng.gen();
        Vec3 t = Vec3(a, b, c) * 2.0 - Vec3::ones();
        if (t.squared_length() <= 1.0) return t;
    }
}
// vec3.rs:217-222 (unused by the hot path; kept for completeness)
inline Vec3 random_unit_vector(Rng& rng) {
    double a = rng.gen_range(0.0, 2.0 * PI);
    double z = rng.gen_range(-1.0, 1.0);
    double r = std::sqrt(1.0 - z * z);
    return Vec3(r * std::cos(a), r * std::sin(a), z);
}
// vec3.rs:223-230
inline Vec3 random_in_unit_disk(Rng& rng) {
    for (;;) {
        double a = rng.gen();
        double b = rng.gen();
        Vec3 t = Vec3(a, b, 0.0) * 2.0 - Vec3(1.0, 1.0, 0.0);
        if (t.squared_length() <= 1.0) return t;
    }
}
// vec3.rs:231-239
inline Vec3 random_cosine_direction(Rng& rng) {
    double r1 = rng.gen();
    double r2 = rng.gen();
    double z = std::sqrt(1.0 - r2);
    double phi = 2.0 * PI * r1;
    double x = std::cos(phi) * std::sqrt(r2);
    double y = std::sin(phi) * std::sqrt(r2);
    return Vec3(x, y, z);
}
// vec3.rs:240-248
inline Vec3 random_to_sphere(double radius, double distance_squared, Rng& rng) {
    double r1 = rng.gen();
    double r2 = rng.gen();
    double z = 1.0 + r2 * (std::sqrt(1.0 - radius * radius / distance_squared) - 1.0);
    double phi = 2.0 * PI * r1;
    double x = std::cos(phi) * std::sqrt(1.0 - z * z);
    double y = std::sin(phi) * std::sqrt(1.0 - z * z);
    return Vec3(x, y, z);
}
// vec3.rs:249-251   v - n * 2.0 * (v * n)
inline Vec3 reflect(Vec3 v, Vec3 n) { return v - n * 2.0 * (v * n); }
// vec3.rs:252-257
inline Vec3 refract(Vec3 v, Vec3 n, double etai_over_etat) {
    double cos_theta = std::fmin((-v) * n, 1.0);
    Vec3 r_out_perp = (v + n * cos_theta) * etai_over_etat;
    Vec3 r_out_parallel = (-n) * std::sqrt(std::fabs(1.0 - r_out_perp.squared_length()));
    return r_out_perp + r_out_parallel;
}

// ray.rs:3-15
struct Ray {
    Vec3 ori, dir;
    double time;
    Ray() : time(0) {}
    Ray(Vec3 o, Vec3 d, double t) : ori(o), dir(d), time(t) {}
    Vec3 at(double t) const { return ori + dir * t; }
};

}  // namespace orc
