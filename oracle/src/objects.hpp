// ORACLE — TEST INFRASTRUCTURE ONLY.
// CPU f64 restatement of reference src/objects.rs (HitRecord, Object, ObjectList, Sphere,
// MovingSphere, RectXY/XZ/YZ, Cuboid, ConstantMedium), src/bvh.rs (Aabb, BvhNode) and
// src/transforms.rs (Translate, RotateY): virtual dispatch, pointer-linked tree, recursion —
// the same structure as the reference's `Box<dyn Object>` graph.
//
// Additions that are NOT in the reference (needed by the deterministic parity probe, SURVEY
// Appendix C): every leaf primitive carries `prim_id` = its pre-order index in the scene graph,
// and HitRecord carries `hit_id = prim_id * 8 + side` (side = cuboid face 0..5, 0 for other
// primitives, 7 for a ConstantMedium event).  ConstantMedium draws its free-flight xi from the
// per-sample generator (the reference uses the thread-local `rand::random`, objects.rs:540); the
// probe can pin xi to a constant instead.
#pragma once
#include <cfloat>
#include <memory>
#include <vector>
#include "vec3.hpp"

namespace orc {

struct Counters {
    unsigned long long paths = 0, casts = 0, prim_tests = 0, node_visits = 0, nan_retries = 0;
    unsigned long long scatters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long misses = 0, depth_exhausted = 0;
    unsigned long long dropped = 0;  // samples given up after max_retries NaN results (the cap is this build's, not the reference's)
    double flops = 0;  // algorithmic flops by the conventions of SURVEY.md 8(d)
    void add(const Counters& o) {
        paths += o.paths; casts += o.casts; prim_tests += o.prim_tests;
        node_visits += o.node_visits; nan_retries += o.nan_retries;
        for (int i = 0; i < 8; i++) scatters[i] += o.scatters[i];
        misses += o.misses; depth_exhausted += o.depth_exhausted; dropped += o.dropped; flops += o.flops;
    }
};
extern thread_local Counters g_cnt;
#ifdef ORACLE_NO_COUNTERS
#define ORC_CNT(stmt) do {} while (0)
#else
#define ORC_CNT(stmt) do { stmt; } while (0)
#endif

struct Material;

// Per-cast context that the reference does not need: where ConstantMedium gets its xi from.
struct CastCtx {
    Rng* rng = nullptr;     // per-sample generator (render)
    double fixed_xi = -1.0;  // >= 0: probe mode, every medium uses this xi
    double medium_xi() { return fixed_xi >= 0.0 ? fixed_xi : rng->gen(); }
};
extern thread_local CastCtx g_cast;

// objects.rs:10-18
struct HitRecord {
    double t = 0;
    Vec3 p, normal;
    const Material* mat_ptr = nullptr;
    double u = 0, v = 0;
    int hit_id = -1;  // oracle addition
};

// bvh.rs:6-59
struct Aabb {
    Vec3 min, max;
    static Aabb surrounding_box(const Aabb& a, const Aabb& b) {
        return Aabb{Vec3(std::fmin(a.min.x, b.min.x), std::fmin(a.min.y, b.min.y), std::fmin(a.min.z, b.min.z)),
                    Vec3(std::fmax(a.max.x, b.max.x), std::fmax(a.max.y, b.max.y), std::fmax(a.max.z, b.max.z))};
    }
    // bvh.rs:26-58 — the LOOSE slab test: every axis is compared against the ORIGINAL
    // [t_min,t_max]; true divisions; f64::min/max == fmin/fmax (a NaN operand is ignored).
    bool hit(const Ray& ray, double t_min, double t_max) const {
        ORC_CNT(g_cnt.node_visits++; g_cnt.flops += 24);
        {
            double t1 = (min.x - ray.ori.x) / ray.dir.x;
            double t2 = (max.x - ray.ori.x) / ray.dir.x;
            if (t1 > t2) std::swap(t1, t2);
            if (std::fmin(t_max, t2) <= std::fmax(t_min, t1)) return false;
        }
        {
            double t1 = (min.y - ray.ori.y) / ray.dir.y;
            double t2 = (max.y - ray.ori.y) / ray.dir.y;
            if (t1 > t2) std::swap(t1, t2);
            if (std::fmin(t_max, t2) <= std::fmax(t_min, t1)) return false;
        }
        {
            double t1 = (min.z - ray.ori.z) / ray.dir.z;
            double t2 = (max.z - ray.ori.z) / ray.dir.z;
            if (t1 > t2) std::swap(t1, t2);
            if (std::fmin(t_max, t2) <= std::fmax(t_min, t1)) return false;
        }
        return true;
    }
};

// objects.rs:19-28
struct Object {
    virtual ~Object() {}
    virtual bool hit(const Ray& ray, double t_min, double t_max, HitRecord& rec) const = 0;
    virtual bool bounding_box(double t1, double t2, Aabb& out) const = 0;
    virtual double pdf_value(Vec3 /*origin*/, Vec3 /*v*/) const { return 0.0; }
    virtual Vec3 random(Vec3 /*origin*/, Rng& /*rng*/) const { return Vec3(1.0, 0.0, 0.0); }
    // oracle additions: pre-order primitive numbering and BVH dump for the flattening parity test
    virtual void assign_ids(int& next) = 0;
    virtual int first_prim_id() const = 0;
    virtual const struct BvhNode* as_bvh_node() const { return nullptr; }
    virtual void collect_bvh_roots(std::vector<const struct BvhNode*>&) const {}
};
using ObjPtr = std::shared_ptr<Object>;

// objects.rs:29-79
struct ObjectList : Object {
    std::vector<ObjPtr> objects;
    void add(ObjPtr o) { objects.push_back(std::move(o)); }
    bool hit(const Ray& ray, double t_min, double t_max, HitRecord& rec) const override {
        bool any = false;
        double closest = t_max;
        HitRecord tmp;
        for (const auto& x : objects) {
            if (x->hit(ray, t_min, closest, tmp)) {
                rec = tmp;
                any = true;
                closest = tmp.t;
            }
        }
        return any;
    }
    bool bounding_box(double t1, double t2, Aabb& out) const override {
        if (objects.empty()) return false;
        Aabb ans;
        if (!objects[0]->bounding_box(t1, t2, ans)) return false;
        for (size_t i = 1; i < objects.size(); i++) {
            Aabb tmp;
            if (!objects[i]->bounding_box(t1, t2, tmp)) return false;
            ans = Aabb::surrounding_box(ans, tmp);
        }
        out = ans;
        return true;
    }
    double pdf_value(Vec3 origin, Vec3 v) const override {
        double weight = 1.0 / (double)objects.size();
        double ans = 0.0;
        for (const auto& x : objects) ans += weight * x->pdf_value(origin, v);
        return ans;
    }
    Vec3 random(Vec3 origin, Rng& rng) const override {
        return objects[rng.gen_index((uint32_t)objects.size())]->random(origin, rng);
    }
    void assign_ids(int& next) override { for (auto& o : objects) o->assign_ids(next); }
    int first_prim_id() const override { return objects.empty() ? -1 : objects[0]->first_prim_id(); }
    void collect_bvh_roots(std::vector<const BvhNode*>& r) const override { for (auto& o : objects) o->collect_bvh_roots(r); }
};

struct ONB;  // pdf.hpp
Vec3 sphere_random_dir(Vec3 center, double radius, Vec3 origin, Rng& rng);  // defined in pdf.hpp

// objects.rs:208-215
inline void get_sphere_uv(Vec3 p, double& u, double& v) {
    double phi = std::atan2(p.z, p.x);
    double theta = std::asin(p.y);
    u = 1.0 - (phi + PI) / (2.0 * PI);
    v = (theta + PI / 2.0) / PI;
}

struct Primitive : Object {
    int prim_id = -1;
    void assign_ids(int& next) override { prim_id = next++; }
    int first_prim_id() const override { return prim_id; }
};

// objects.rs:81-144
struct Sphere : Primitive {
    Vec3 center;
    double radius;
    std::shared_ptr<Material> material;
    Sphere(Vec3 c, double r, std::shared_ptr<Material> m) : center(c), radius(r), material(std::move(m)) {}
    bool hit(const Ray& ray, double t_min, double t_max, HitRecord& rec) const override {
        ORC_CNT(g_cnt.prim_tests++; g_cnt.flops += 30);
        Vec3 oc = ray.ori - center;
        double a = ray.dir * ray.dir;
        double half_b = oc * ray.dir;
        double c = oc * oc - radius * radius;
        double discriminant = half_b * half_b - a * c;
        if (discriminant > 0.0) {
            double ans = (-half_b - std::sqrt(discriminant)) / a;
            double tu, tv;
            get_sphere_uv((ray.at(ans) - center) / radius, tu, tv);  // uv from the FIRST root (:95)
            ORC_CNT(g_cnt.flops += 2);
            if (ans > t_min && ans < t_max) {
                ORC_CNT(g_cnt.flops += 12);
                rec.t = ans; rec.p = ray.at(ans); rec.normal = (ray.at(ans) - center) / radius;
                rec.mat_ptr = material.get(); rec.u = tu; rec.v = tv; rec.hit_id = prim_id * 8;
                return true;
            }
            ans = (-half_b + std::sqrt(discriminant)) / a;
            if (ans > t_min && ans < t_max) {
                ORC_CNT(g_cnt.flops += 12);
                rec.t = ans; rec.p = ray.at(ans); rec.normal = (ray.at(ans) - center) / radius;
                rec.mat_ptr = material.get(); rec.u = tu; rec.v = tv; rec.hit_id = prim_id * 8;
                return true;
            }
        }
        return false;
    }
    bool bounding_box(double, double, Aabb& out) const override {
        out = Aabb{center - Vec3(radius, radius, radius), center + Vec3(radius, radius, radius)};
        return true;
    }
    double pdf_value(Vec3 origin, Vec3 v) const override {
        HitRecord rec;
        if (hit(Ray(origin, v, 0.0), 0.001, DBL_MAX, rec)) {
            double cos_theta_max = std::sqrt(1.0 - radius * radius / (center - origin).squared_length());
            double solid_angle = 2.0 * PI * (1.0 - cos_theta_max);
            return 1.0 / solid_angle;
        }
        return 0.0;
    }
    Vec3 random(Vec3 origin, Rng& rng) const override { return sphere_random_dir(center, radius, origin, rng); }
};

// objects.rs:146-206
struct MovingSphere : Primitive {
    Vec3 center1, center2;
    double t1, t2, radius;
    std::shared_ptr<Material> material;
    MovingSphere(Vec3 c1, Vec3 c2, double t1_, double t2_, double r, std::shared_ptr<Material> m)
        : center1(c1), center2(c2), t1(t1_), t2(t2_), radius(r), material(std::move(m)) {}
    Vec3 get_center(double t) const { return center1 + (center2 - center1) * ((t - t1) / (t2 - t1)); }
    bool hit(const Ray& ray, double t_min, double t_max, HitRecord& rec) const override {
        ORC_CNT(g_cnt.prim_tests++; g_cnt.flops += 30 + 10);
        Vec3 oc = ray.ori - get_center(ray.time);
        double a = ray.dir * ray.dir;
        double half_b = oc * ray.dir;
        double c = oc * oc - radius * radius;
        double discriminant = half_b * half_b - a * c;
        if (discriminant > 0.0) {
            double ans = (-half_b - std::sqrt(discriminant)) / a;
            double tu, tv;
            get_sphere_uv((ray.at(ans) - get_center(ray.time)) / radius, tu, tv);
            ORC_CNT(g_cnt.flops += 2);
            if (ans > t_min && ans < t_max) {
                ORC_CNT(g_cnt.flops += 12);
                rec.t = ans; rec.p = ray.at(ans); rec.normal = (ray.at(ans) - get_center(ray.time)) / radius;
                rec.mat_ptr = material.get(); rec.u = tu; rec.v = tv; rec.hit_id = prim_id * 8;
                return true;
            }
            ans = (-half_b + std::sqrt(discriminant)) / a;
            if (ans > t_min && ans < t_max) {
                ORC_CNT(g_cnt.flops += 12);
                rec.t = ans; rec.p = ray.at(ans); rec.normal = (ray.at(ans) - get_center(ray.time)) / radius;
                rec.mat_ptr = material.get(); rec.u = tu; rec.v = tv; rec.hit_id = prim_id * 8;
                return true;
            }
        }
        return false;
    }
    bool bounding_box(double ta, double tb, Aabb& out) const override {
        Vec3 r(radius, radius, radius);
        out = Aabb::surrounding_box(Aabb{get_center(ta) - r, get_center(ta) + r},
                                    Aabb{get_center(tb) - r, get_center(tb) + r});
        return true;
    }
};

// objects.rs:217-378. One class, axis-tagged: AXIS 0 = RectXY (k on z), 1 = RectXZ (k on y),
// 2 = RectYZ (k on x).  (a1,a2) and (b1,b2) are the two in-plane ranges in the reference's
// field order: XY -> (x,y), XZ -> (x,z), YZ -> (y,z).
struct Rect : Primitive {
    int axis;
    double a1, a2, b1, b2, k, face;
    std::shared_ptr<Material> material;
    int side = 0;          // cuboid face index (oracle addition)
    bool own_id = true;    // false when owned by a Cuboid (the cuboid holds the prim_id)
    Rect(int axis_, double a1_, double a2_, double b1_, double b2_, double k_, double face_,
         std::shared_ptr<Material> m)
        : axis(axis_), a1(a1_), a2(a2_), b1(b1_), b2(b2_), k(k_), face(face_), material(std::move(m)) {}
    bool hit(const Ray& ray, double t_min, double t_max, HitRecord& rec) const override {
        ORC_CNT(g_cnt.prim_tests++; g_cnt.flops += 8);
        double ok, dk, oa, da, ob, db;
        if (axis == 0) { ok = ray.ori.z; dk = ray.dir.z; oa = ray.ori.x; da = ray.dir.x; ob = ray.ori.y; db = ray.dir.y; }
        else if (axis == 1) { ok = ray.ori.y; dk = ray.dir.y; oa = ray.ori.x; da = ray.dir.x; ob = ray.ori.z; db = ray.dir.z; }
        else { ok = ray.ori.x; dk = ray.dir.x; oa = ray.ori.y; da = ray.dir.y; ob = ray.ori.z; db = ray.dir.z; }
        double t = (k - ok) / dk;
        if (t >= t_min && t <= t_max) {  // INCLUSIVE (objects.rs:229,283,337)
            double a = oa + t * da;
            double b = ob + t * db;
            if (a >= a1 && a <= a2 && b >= b1 && b <= b2) {
                ORC_CNT(g_cnt.flops += 14);
                rec.t = t;
                rec.p = ray.at(t);
                rec.normal = (axis == 0 ? Vec3(0.0, 0.0, 1.0) : axis == 1 ? Vec3(0.0, 1.0, 0.0) : Vec3(1.0, 0.0, 0.0)) * face;
                rec.mat_ptr = material.get();
                rec.u = (a - a1) / (a2 - a1);
                rec.v = (b - b1) / (b2 - b1);
                rec.hit_id = prim_id * 8 + side;
                return true;
            }
        }
        return false;
    }
    bool bounding_box(double, double, Aabb& out) const override {
        if (axis == 0) out = Aabb{Vec3(a1, b1, k - 0.0001), Vec3(a2, b2, k + 0.0001)};
        else if (axis == 1) out = Aabb{Vec3(a1, k - 0.0001, b1), Vec3(a2, k + 0.0001, b2)};
        else out = Aabb{Vec3(k - 0.0001, a1, b1), Vec3(k + 0.0001, a2, b2)};
        return true;
    }
    double pdf_value(Vec3 origin, Vec3 v) const override {
        HitRecord rec;
        if (hit(Ray(origin, v, 0.0), 0.001, DBL_MAX, rec)) {
            ORC_CNT(g_cnt.flops += 16);
            double area = (a2 - a1) * (b2 - b1);
            double distance_squared = rec.t * rec.t * v.squared_length();
            double cosine = std::fabs(v * rec.normal) / v.length();
            return distance_squared / (cosine * area);
        }
        return 0.0;
    }
    Vec3 random(Vec3 origin, Rng& rng) const override {
        double a = rng.gen_range(a1, a2);
        double b = rng.gen_range(b1, b2);
        Vec3 random_point = axis == 0 ? Vec3(a, b, k) : axis == 1 ? Vec3(a, k, b) : Vec3(k, a, b);
        return random_point - origin;
    }
};

// objects.rs:380-512
struct Cuboid : Primitive {
    Vec3 box_min, box_max;
    std::shared_ptr<Rect> sides[6];
    Cuboid(Vec3 mn, Vec3 mx, std::shared_ptr<Material> m) : box_min(mn), box_max(mx) {
        sides[0] = std::make_shared<Rect>(0, mn.x, mx.x, mn.y, mx.y, mn.z, -1.0, m);
        sides[1] = std::make_shared<Rect>(0, mn.x, mx.x, mn.y, mx.y, mx.z, 1.0, m);
        sides[2] = std::make_shared<Rect>(1, mn.x, mx.x, mn.z, mx.z, mn.y, -1.0, m);
        sides[3] = std::make_shared<Rect>(1, mn.x, mx.x, mn.z, mx.z, mx.y, 1.0, m);
        sides[4] = std::make_shared<Rect>(2, mn.y, mx.y, mn.z, mx.z, mn.x, -1.0, m);
        sides[5] = std::make_shared<Rect>(2, mn.y, mx.y, mn.z, mx.z, mx.x, 1.0, m);
        for (int i = 0; i < 6; i++) { sides[i]->side = i; sides[i]->own_id = false; }
    }
    void assign_ids(int& next) override {
        prim_id = next++;
        for (auto& s : sides) s->prim_id = prim_id;
    }
    bool hit(const Ray& ray, double t_min, double t_max, HitRecord& rec) const override {
        bool any = false;
        double closest = t_max;
        HitRecord tmp;
        for (int i = 0; i < 6; i++) {
            if (sides[i]->hit(ray, t_min, closest, tmp)) {
                rec = tmp;
                any = true;
                closest = tmp.t;
            }
        }
        return any;
    }
    bool bounding_box(double, double, Aabb& out) const override { out = Aabb{box_min, box_max}; return true; }
    double pdf_value(Vec3 origin, Vec3 v) const override {
        double weight = 1.0 / 6.0;
        return weight * sides[0]->pdf_value(origin, v) + weight * sides[1]->pdf_value(origin, v) +
               weight * sides[2]->pdf_value(origin, v) + weight * sides[3]->pdf_value(origin, v) +
               weight * sides[4]->pdf_value(origin, v) + weight * sides[5]->pdf_value(origin, v);
    }
    Vec3 random(Vec3 origin, Rng& rng) const override { return sides[rng.gen_index(6)]->random(origin, rng); }
};

// objects.rs:514-560
struct ConstantMedium : Object {
    ObjPtr boundary;
    std::shared_ptr<Material> phase_function;  // Isotropic{texture}
    double density;
    ConstantMedium(ObjPtr b, std::shared_ptr<Material> phase, double d)
        : boundary(std::move(b)), phase_function(std::move(phase)), density(d) {}
    bool hit(const Ray& ray, double t_min, double t_max, HitRecord& rec) const override {
        HitRecord rec1, rec2;
        if (boundary->hit(ray, -DBL_MAX, DBL_MAX, rec1)) {  // f64::MIN == -DBL_MAX
            if (boundary->hit(ray, rec1.t + 0.0001, DBL_MAX, rec2)) {
                if (rec1.t < t_min) rec1.t = t_min;
                if (rec2.t > t_max) rec2.t = t_max;
                if (rec1.t < rec2.t) {
                    ORC_CNT(g_cnt.flops += 12);
                    double distance_inside_boundary = (rec2.t - rec1.t) * ray.dir.length();
                    double hit_distance = -(1.0 / density) * std::log(g_cast.medium_xi());
                    if (hit_distance <= distance_inside_boundary) {
                        double t = rec1.t + hit_distance / ray.dir.length();
                        rec.t = t; rec.p = ray.at(t); rec.normal = Vec3(1.0, 0.0, 0.0);
                        rec.mat_ptr = phase_function.get(); rec.u = 0.0; rec.v = 0.0;
                        rec.hit_id = boundary->first_prim_id() * 8 + 7;
                        return true;
                    }
                }
            }
        }
        return false;
    }
    bool bounding_box(double t1, double t2, Aabb& out) const override { return boundary->bounding_box(t1, t2, out); }
    void assign_ids(int& next) override { boundary->assign_ids(next); }
    int first_prim_id() const override { return boundary->first_prim_id(); }
    void collect_bvh_roots(std::vector<const BvhNode*>& r) const override { boundary->collect_bvh_roots(r); }
};

// transforms.rs:5-40
struct Translate : Object {
    ObjPtr object;
    Vec3 offset;
    Translate(ObjPtr o, Vec3 off) : object(std::move(o)), offset(off) {}
    bool hit(const Ray& ray, double t_min, double t_max, HitRecord& rec) const override {
        ORC_CNT(g_cnt.flops += 3);
        Ray moved_ray(ray.ori - offset, ray.dir, ray.time);
        if (object->hit(moved_ray, t_min, t_max, rec)) {
            ORC_CNT(g_cnt.flops += 3);
            rec.p = rec.p + offset;
            return true;
        }
        return false;
    }
    bool bounding_box(double t1, double t2, Aabb& out) const override {
        Aabb r;
        if (!object->bounding_box(t1, t2, r)) return false;
        out = Aabb{r.min + offset, r.max + offset};
        return true;
    }
    double pdf_value(Vec3 origin, Vec3 v) const override { return object->pdf_value(origin - offset, v); }
    Vec3 random(Vec3 origin, Rng& rng) const override { return object->random(origin - offset, rng); }
    void assign_ids(int& next) override { object->assign_ids(next); }
    int first_prim_id() const override { return object->first_prim_id(); }
    void collect_bvh_roots(std::vector<const BvhNode*>& r) const override { object->collect_bvh_roots(r); }
};

// transforms.rs:42-150
struct RotateY : Object {
    ObjPtr object;
    double sin_theta, cos_theta;
    bool has_box;
    Aabb boxx;
    RotateY(ObjPtr o, double angle) : object(std::move(o)) {
        double radians = angle * PI / 180.0;
        sin_theta = std::sin(radians);
        cos_theta = std::cos(radians);
        init_box();
    }
    // oracle addition: rebuild from a flattened scene, where sin/cos are stored (transforms.rs:44-45 fields)
    RotateY(ObjPtr o, double sin_t, double cos_t, int /*from_sincos*/) : object(std::move(o)), sin_theta(sin_t), cos_theta(cos_t) { init_box(); }
    void init_box() {  // transforms.rs:53-87
        Aabb bb;
        has_box = object->bounding_box(0.0, 1.0, bb);
        if (has_box) {
            Vec3 mn(DBL_MAX, DBL_MAX, DBL_MAX), mx(-DBL_MAX, -DBL_MAX, -DBL_MAX);
            for (int i = 0; i < 2; i++)
                for (int j = 0; j < 2; j++)
                    for (int k = 0; k < 2; k++) {
                        double x = (double)i * bb.max.x + (1.0 - (double)i) * bb.min.x;
                        double y = (double)j * bb.max.y + (1.0 - (double)j) * bb.min.y;
                        double z = (double)k * bb.max.z + (1.0 - (double)k) * bb.min.z;
                        Vec3 tester(cos_theta * x + sin_theta * z, y, -sin_theta * x + cos_theta * z);
                        mn.x = std::fmin(mn.x, tester.x); mn.y = std::fmin(mn.y, tester.y); mn.z = std::fmin(mn.z, tester.z);
                        mx.x = std::fmax(mx.x, tester.x); mx.y = std::fmax(mx.y, tester.y); mx.z = std::fmax(mx.z, tester.z);
                    }
            boxx = Aabb{mn, mx};
        }
    }
    bool hit(const Ray& ray, double t_min, double t_max, HitRecord& rec) const override {
        ORC_CNT(g_cnt.flops += 12);
        Ray rotated_ray(Vec3(cos_theta * ray.ori.x - sin_theta * ray.ori.z, ray.ori.y,
                             sin_theta * ray.ori.x + cos_theta * ray.ori.z),
                        Vec3(cos_theta * ray.dir.x - sin_theta * ray.dir.z, ray.dir.y,
                             sin_theta * ray.dir.x + cos_theta * ray.dir.z),
                        ray.time);
        if (object->hit(rotated_ray, t_min, t_max, rec)) {
            ORC_CNT(g_cnt.flops += 12);
            Vec3 p(cos_theta * rec.p.x + sin_theta * rec.p.z, rec.p.y, -sin_theta * rec.p.x + cos_theta * rec.p.z);
            Vec3 n(cos_theta * rec.normal.x + sin_theta * rec.normal.z, rec.normal.y,
                   -sin_theta * rec.normal.x + cos_theta * rec.normal.z);
            rec.p = p;
            rec.normal = n;
            return true;
        }
        return false;
    }
    bool bounding_box(double, double, Aabb& out) const override {
        if (has_box) out = boxx;
        return has_box;
    }
    double pdf_value(Vec3 origin, Vec3 v) const override {
        Vec3 ro(cos_theta * origin.x - sin_theta * origin.z, origin.y, sin_theta * origin.x + cos_theta * origin.z);
        Vec3 rv(cos_theta * v.x - sin_theta * v.z, v.y, sin_theta * v.x + cos_theta * v.z);
        return object->pdf_value(ro, rv);
    }
    Vec3 random(Vec3 origin, Rng& rng) const override {
        Vec3 ro(cos_theta * origin.x - sin_theta * origin.z, origin.y, sin_theta * origin.x + cos_theta * origin.z);
        Vec3 rec = object->random(ro, rng);
        return Vec3(cos_theta * rec.x + sin_theta * rec.z, rec.y, -sin_theta * rec.x + cos_theta * rec.z);
    }
    void assign_ids(int& next) override { object->assign_ids(next); }
    int first_prim_id() const override { return object->first_prim_id(); }
    void collect_bvh_roots(std::vector<const BvhNode*>& r) const override { object->collect_bvh_roots(r); }
};

// bvh.rs:79-133, 174-203.  The reference draws the split axis from `thread_rng` (unseeded,
// bvh.rs:89); the oracle draws it from the seeded scene generator in the same call order
// (axis first, then sort, then the len==1 early-out, then left before right).
struct BvhNode : Object {
    ObjPtr left, right;
    Aabb boxx;
    static ObjPtr build(std::vector<ObjPtr>& objects, double t1, double t2, Rng& rng);
    bool hit(const Ray& ray, double t_min, double t_max, HitRecord& rec) const override {
        if (!boxx.hit(ray, t_min, t_max)) return false;
        HitRecord tmp;
        if (left->hit(ray, t_min, t_max, tmp)) {
            HitRecord tmp2;
            if (right->hit(ray, t_min, tmp.t, tmp2)) { rec = tmp2; return true; }
            rec = tmp;
            return true;
        }
        if (right->hit(ray, t_min, t_max, tmp)) { rec = tmp; return true; }
        return false;
    }
    bool bounding_box(double, double, Aabb& out) const override { out = boxx; return true; }
    void assign_ids(int& next) override { left->assign_ids(next); right->assign_ids(next); }
    int first_prim_id() const override { return left->first_prim_id(); }
    const BvhNode* as_bvh_node() const override { return this; }
    void collect_bvh_roots(std::vector<const BvhNode*>& r) const override { r.push_back(this); }
};

}  // namespace orc
