// scene_api.hpp — host-side mirror of the reference's scene vocabulary.
//
// Same type names, fields and constructors as the reference so a scene written against
// src/scenes.rs reads the same here:
//   Vec3 (src/vec3.rs:4-54)                      Camera (src/camera.rs:7-56)
//   SolidColor / CheckerTextureUV / CheckerTexture / NoiseTexture / ImageTexture / Perlin
//                                                 (src/texture.rs:11-137)
//   Lambertian / Metal / Dielectric / FrostedDielectric / DiffuseLight / Isotropic
//                                                 (src/materials.rs:31-173)
//   Sphere / MovingSphere / RectXY / RectXZ / RectYZ / Cuboid / ConstantMedium / ObjectList
//                                                 (src/objects.rs:29-560)
//   Translate / RotateY (src/transforms.rs)       Aabb / Bvh / BvhNode (src/bvh.rs)
//
// These are DESCRIPTIONS: they know their fields and their bounding box (needed by the BVH build
// and RotateY::new) but they have no hit/scatter/value methods — all evaluation happens on the
// GPU behind include/rt_hip.h.  There is no CPU render path in the product.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

namespace rth {

struct Vec3 {
    double x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(double a, double b, double c) : x(a), y(b), z(c) {}
    static Vec3 ones() { return {1.0, 1.0, 1.0}; }
    static Vec3 zero() { return {0.0, 0.0, 0.0}; }
    double length() const { return std::sqrt(x * x + y * y + z * z); }
    Vec3 unit() const {
        double len = length();
        if (len == 0.0) throw std::runtime_error("division by zero");  // vec3.rs:32-34 panics
        return {x / len, y / len, z / len};
    }
    static Vec3 elemul(Vec3 a, Vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
    static Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - b.y * a.z, b.x * a.z - a.x * b.z, a.x * b.y - b.x * a.y}; }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator/(Vec3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }

// Seeded stand-in for the reference's entropy-seeded SmallRng / thread_rng during scene
// construction (scenes.rs:10,174,844; bvh.rs:89; texture.rs:122,131): splitmix64-keyed
// xoshiro128++, the generator specified in DESIGN.md (the device path uses the same one per
// sample).  gen / gen_range / gen_index follow rand 0.7.3's distributions.
class SceneRng {
  public:
    SceneRng(uint64_t seed, uint32_t stream);
    double gen();
    double gen_range(double lo, double hi);
    uint32_t gen_index(uint32_t n);
    Vec3 random_vec3(double lo, double hi);  // Vec3::random, vec3.rs:47-53

  private:
    uint32_t next_u32();
    uint64_t next_u64();
    uint32_t s_[4];
};

struct Aabb {
    Vec3 min, max;
    static Aabb surrounding_box(const Aabb& a, const Aabb& b);  // bvh.rs:12-25
};

// ---- textures --------------------------------------------------------------------------------
struct Perlin {  // texture.rs:100-137
    std::vector<int32_t> perm_x, perm_y, perm_z;
    std::vector<Vec3> ranvec;
    explicit Perlin(SceneRng& rng);
};
struct RgbImage {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgb;
};
// jpeg.cpp: baseline JPEG -> RGB8 (what `image::open(..).to_rgb()` does for images/earthmap.jpg, texture.rs:70-72)
bool decode_jpeg(const uint8_t* data, size_t n, RgbImage& out, std::string& err);
std::shared_ptr<const RgbImage> load_image(const std::string& filename);
struct Texture {
    enum Kind { SOLID, CHECKER_UV, CHECKER, NOISE, IMAGE } kind;
    explicit Texture(Kind k) : kind(k) {}
    virtual ~Texture() = default;
};
using TexturePtr = std::shared_ptr<const Texture>;
struct SolidColor : Texture {
    Vec3 color;
    explicit SolidColor(Vec3 c) : Texture(SOLID), color(c) {}
};
struct CheckerTextureUV : Texture {
    TexturePtr odd, even;
    CheckerTextureUV(TexturePtr o, TexturePtr e) : Texture(CHECKER_UV), odd(std::move(o)), even(std::move(e)) {}
};
struct CheckerTexture : Texture {
    TexturePtr odd, even;
    CheckerTexture(TexturePtr o, TexturePtr e) : Texture(CHECKER), odd(std::move(o)), even(std::move(e)) {}
};
struct NoiseTexture : Texture {
    std::shared_ptr<const Perlin> noise;
    double scale;
    NoiseTexture(std::shared_ptr<const Perlin> n, double s) : Texture(NOISE), noise(std::move(n)), scale(s) {}
};
struct ImageTexture : Texture {
    std::shared_ptr<const RgbImage> img;
    // texture.rs:68-74: a baseline JPEG (decoded by jpeg.cpp) or a pre-decoded raster (binary PPM)
    explicit ImageTexture(const std::string& filename);
    explicit ImageTexture(std::shared_ptr<const RgbImage> i) : Texture(IMAGE), img(std::move(i)) {}
};

// ---- materials -------------------------------------------------------------------------------
struct Material {
    enum Kind { LAMBERTIAN, METAL, DIELECTRIC, DIFFUSE_LIGHT, ISOTROPIC, FROSTED_DIELECTRIC } kind;
    explicit Material(Kind k) : kind(k) {}
    virtual ~Material() = default;
};
using MaterialPtr = std::shared_ptr<const Material>;
struct Lambertian : Material {
    TexturePtr albedo;
    explicit Lambertian(TexturePtr a) : Material(LAMBERTIAN), albedo(std::move(a)) {}
};
struct Metal : Material {
    Vec3 albedo;
    double fuzz;
    Metal(Vec3 a, double f) : Material(METAL), albedo(a), fuzz(f) {}
};
struct Dielectric : Material {
    double ref_idx;
    explicit Dielectric(double r) : Material(DIELECTRIC), ref_idx(r) {}
};
struct FrostedDielectric : Material {
    double ref_idx, fuzz;
    FrostedDielectric(double r, double f) : Material(FROSTED_DIELECTRIC), ref_idx(r), fuzz(f) {}
};
struct DiffuseLight : Material {
    TexturePtr emit;
    explicit DiffuseLight(TexturePtr e) : Material(DIFFUSE_LIGHT), emit(std::move(e)) {}
};
struct Isotropic : Material {
    TexturePtr albedo;
    explicit Isotropic(TexturePtr a) : Material(ISOTROPIC), albedo(std::move(a)) {}
};

// ---- objects ---------------------------------------------------------------------------------
struct Object {
    enum Kind { LIST, SPHERE, MOVING_SPHERE, RECT_XY, RECT_XZ, RECT_YZ, CUBOID, CONSTANT_MEDIUM, TRANSLATE, ROTATE_Y, BVH, BVH_NODE } kind;
    explicit Object(Kind k) : kind(k) {}
    virtual ~Object() = default;
    virtual std::optional<Aabb> bounding_box(double t1, double t2) const = 0;  // objects.rs:21
};
using ObjectPtr = std::shared_ptr<const Object>;

struct ObjectList : Object {
    std::vector<ObjectPtr> objects;
    ObjectList() : Object(LIST) {}
    void add(ObjectPtr o) { objects.push_back(std::move(o)); }
    std::optional<Aabb> bounding_box(double t1, double t2) const override;  // objects.rs:49-67
};
struct Sphere : Object {
    Vec3 center;
    double radius;
    MaterialPtr material;
    Sphere(Vec3 c, double r, MaterialPtr m) : Object(SPHERE), center(c), radius(r), material(std::move(m)) {}
    std::optional<Aabb> bounding_box(double, double) const override;  // objects.rs:120-125
};
struct MovingSphere : Object {
    Vec3 center1, center2;
    double t1, t2, radius;
    MaterialPtr material;
    MovingSphere(Vec3 c1, Vec3 c2, double ta, double tb, double r, MaterialPtr m)
        : Object(MOVING_SPHERE), center1(c1), center2(c2), t1(ta), t2(tb), radius(r), material(std::move(m)) {}
    Vec3 get_center(double t) const;                                        // objects.rs:155-157
    std::optional<Aabb> bounding_box(double ta, double tb) const override;  // objects.rs:194-205
};
// RectXY { x1,x2,y1,y2,k,face }, RectXZ { x1,x2,z1,z2,k,face }, RectYZ { y1,y2,z1,z2,k,face }
struct Rect : Object {
    double a1, a2, b1, b2, k, face;
    MaterialPtr material;
    Rect(Kind kind_, double a1_, double a2_, double b1_, double b2_, double k_, double face_, MaterialPtr m)
        : Object(kind_), a1(a1_), a2(a2_), b1(b1_), b2(b2_), k(k_), face(face_), material(std::move(m)) {}
    std::optional<Aabb> bounding_box(double, double) const override;  // objects.rs:245-250,299-304,353-358
};
struct RectXY : Rect {
    RectXY(double x1, double x2, double y1, double y2, double k, double face, MaterialPtr m) : Rect(RECT_XY, x1, x2, y1, y2, k, face, std::move(m)) {}
};
struct RectXZ : Rect {
    RectXZ(double x1, double x2, double z1, double z2, double k, double face, MaterialPtr m) : Rect(RECT_XZ, x1, x2, z1, z2, k, face, std::move(m)) {}
};
struct RectYZ : Rect {
    RectYZ(double y1, double y2, double z1, double z2, double k, double face, MaterialPtr m) : Rect(RECT_YZ, y1, y2, z1, z2, k, face, std::move(m)) {}
};
struct Cuboid : Object {  // objects.rs:381-456; the six sides are implied by (box_min, box_max)
    Vec3 box_min, box_max;
    MaterialPtr material;
    Cuboid(Vec3 mn, Vec3 mx, MaterialPtr m) : Object(CUBOID), box_min(mn), box_max(mx), material(std::move(m)) {}
    std::optional<Aabb> bounding_box(double, double) const override { return Aabb{box_min, box_max}; }
};
struct ConstantMedium : Object {  // objects.rs:514-527
    ObjectPtr boundary;
    std::shared_ptr<const Isotropic> phase_function;
    double density;
    ConstantMedium(ObjectPtr b, TexturePtr texture, double d)
        : Object(CONSTANT_MEDIUM), boundary(std::move(b)), phase_function(std::make_shared<Isotropic>(std::move(texture))), density(d) {}
    std::optional<Aabb> bounding_box(double t1, double t2) const override { return boundary->bounding_box(t1, t2); }
};
struct Translate : Object {  // transforms.rs:5-33
    ObjectPtr object;
    Vec3 offset;
    Translate(ObjectPtr o, Vec3 off) : Object(TRANSLATE), object(std::move(o)), offset(off) {}
    std::optional<Aabb> bounding_box(double t1, double t2) const override;
};
struct RotateY : Object {  // transforms.rs:42-88
    ObjectPtr object;
    double sin_theta, cos_theta;
    std::optional<Aabb> boxx;
    RotateY(ObjectPtr o, double angle_degrees);
    std::optional<Aabb> bounding_box(double, double) const override { return boxx; }
};
struct BvhNode : Object {  // bvh.rs:79-133
    ObjectPtr left, right;
    Aabb boxx;
    BvhNode() : Object(BVH_NODE) {}
    // bvh.rs:85-111; the split axis comes from `rng` instead of thread_rng
    static ObjectPtr build(std::vector<ObjectPtr>& objects, double t1, double t2, SceneRng& rng);
    std::optional<Aabb> bounding_box(double, double) const override { return boxx; }
};
struct Bvh : Object {  // bvh.rs:61-78
    ObjectPtr root;
    Bvh(std::vector<ObjectPtr>& objects, double t1, double t2, SceneRng& rng) : Object(BVH), root(BvhNode::build(objects, t1, t2, rng)) {}
    std::optional<Aabb> bounding_box(double t1, double t2) const override { return root->bounding_box(t1, t2); }
};

struct Camera {  // camera.rs:7-56
    Vec3 origin, lower_left_corner, horizontal, vertical, u, v, w;
    double lens_radius = 0, t1 = 0, t2 = 1;
    Camera() = default;
    Camera(Vec3 lookfrom, Vec3 lookat, Vec3 vup, double vfov, double aspect_ratio, double aperture, double focus_dist, double t1, double t2);
};

// What every scenes.rs constructor returns (3-tuple variants: lights = nullptr)
struct Scene {
    std::shared_ptr<ObjectList> world;
    Vec3 background;
    Camera cam;
    std::shared_ptr<ObjectList> lights;
};

// scenes.rs constructors (scenes.cpp)
Scene random_scene(double aspect_ratio, uint64_t seed);
Scene random_scene_light(double aspect_ratio, uint64_t seed);
Scene two_checker_spheres(double aspect_ratio);
Scene two_perlin_spheres(double aspect_ratio, uint64_t seed);
Scene earth(double aspect_ratio, const std::string& image_path);
Scene simple_light(double aspect_ratio, uint64_t seed);
Scene cornell_box(double aspect_ratio);
Scene cornell_smoke(double aspect_ratio);
Scene final_scene(double aspect_ratio, uint64_t seed, const std::string& image_path);

}  // namespace rth
