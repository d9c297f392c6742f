// scene_api.cpp — bounding boxes, BVH build, Perlin tables, camera set-up (host, once per scene).
#include "scene_api.hpp"

#include <algorithm>
#include <cstdio>

namespace rth {

namespace {
constexpr double kPi = 3.14159265358979323846;
inline uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
inline uint64_t splitmix64(uint64_t& state) {
    state += 0x9E3779B97F4A7C15ull;
    uint64_t z = state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
}  // namespace

// key = ~stream (sample streams use key = pixel<<32 | sample, so scene streams cannot collide)
SceneRng::SceneRng(uint64_t seed, uint32_t stream) {
    uint64_t state = seed + (~(uint64_t)stream) * 0x9E3779B97F4A7C15ull;
    uint64_t w0 = splitmix64(state), w1 = splitmix64(state);
    s_[0] = (uint32_t)w0; s_[1] = (uint32_t)(w0 >> 32); s_[2] = (uint32_t)w1; s_[3] = (uint32_t)(w1 >> 32);
    if (!(s_[0] | s_[1] | s_[2] | s_[3])) s_[0] = 1;
}
uint32_t SceneRng::next_u32() {  // xoshiro128++
    const uint32_t out = rotl32(s_[0] + s_[3], 7) + s_[0];
    const uint32_t t = s_[1] << 9;
    s_[2] ^= s_[0]; s_[3] ^= s_[1]; s_[1] ^= s_[2]; s_[0] ^= s_[3];
    s_[2] ^= t;
    s_[3] = rotl32(s_[3], 11);
    return out;
}
uint64_t SceneRng::next_u64() {
    uint64_t hi = next_u32();
    return (hi << 32) | next_u32();
}
double SceneRng::gen() { return (double)(next_u64() >> 11) * 0x1.0p-53; }
double SceneRng::gen_range(double lo, double hi) {
    if (!(lo < hi)) throw std::runtime_error("gen_range: low >= high");  // rand asserts
    for (;;) {
        double frac = (double)(next_u64() >> 12) * 0x1.0p-52;
        double v = lo + (hi - lo) * frac;
        if (v < hi) return v;
    }
}
uint32_t SceneRng::gen_index(uint32_t n) {
    uint64_t m = (uint64_t)next_u32() * n;
    uint32_t low = (uint32_t)m;
    if (low < n) {
        const uint32_t zone = (0u - n) % n;
        while (low < zone) { m = (uint64_t)next_u32() * n; low = (uint32_t)m; }
    }
    return (uint32_t)(m >> 32);
}
Vec3 SceneRng::random_vec3(double lo, double hi) {
    double a = gen_range(lo, hi), b = gen_range(lo, hi), c = gen_range(lo, hi);
    return {a, b, c};
}

Aabb Aabb::surrounding_box(const Aabb& a, const Aabb& b) {
    return {{std::fmin(a.min.x, b.min.x), std::fmin(a.min.y, b.min.y), std::fmin(a.min.z, b.min.z)},
            {std::fmax(a.max.x, b.max.x), std::fmax(a.max.y, b.max.y), std::fmax(a.max.z, b.max.z)}};
}

Perlin::Perlin(SceneRng& rng) {
    auto make_perm = [&rng]() {
        std::vector<int32_t> p(256);
        for (int i = 0; i < 256; i++) p[i] = i;
        for (uint32_t i = 255; i >= 1; i--) std::swap(p[i], p[rng.gen_index(i + 1)]);  // SliceRandom::shuffle
        return p;
    };
    perm_x = make_perm();
    perm_y = make_perm();
    perm_z = make_perm();
    ranvec.reserve(256);
    for (int i = 0; i < 256; i++) ranvec.push_back(rng.random_vec3(-1.0, 1.0).unit());
}

ImageTexture::ImageTexture(const std::string& filename) : Texture(IMAGE) { img = load_image(filename); }

// texture.rs:68-74: image::open(filename).to_rgb().  JPEG (baseline, jpeg.cpp) or 8-bit binary PPM, told apart by content.
std::shared_ptr<const RgbImage> load_image(const std::string& filename) {
    FILE* f = std::fopen(filename.c_str(), "rb");
    if (!f) throw std::runtime_error("ImageTexture:failed to open file " + filename);
    auto raster = std::make_shared<RgbImage>();
    unsigned char magic[2] = {0, 0};
    if (std::fread(magic, 1, 2, f) == 2 && magic[0] == 0xFF && magic[1] == 0xD8) {
        std::vector<uint8_t> data;
        std::fseek(f, 0, SEEK_END);
        const long size = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        data.resize(size > 0 ? (size_t)size : 0);
        const bool read_ok = std::fread(data.data(), 1, data.size(), f) == data.size();
        std::fclose(f);
        std::string err;
        if (!read_ok || !decode_jpeg(data.data(), data.size(), *raster, err)) throw std::runtime_error("ImageTexture: " + filename + ": " + (read_ok ? err : "read error"));
        return raster;
    }
    std::fseek(f, 0, SEEK_SET);
    int maxval = 0;
    bool ok = std::fscanf(f, "P6 %u %u %d", &raster->width, &raster->height, &maxval) == 3 && maxval == 255 && std::fgetc(f) != EOF;
    if (ok) {
        raster->rgb.resize((size_t)raster->width * raster->height * 3);
        ok = std::fread(raster->rgb.data(), 1, raster->rgb.size(), f) == raster->rgb.size();
    }
    std::fclose(f);
    if (!ok) throw std::runtime_error("ImageTexture: " + filename + " is neither a baseline JPEG nor an 8-bit binary PPM");
    return raster;
}

std::optional<Aabb> ObjectList::bounding_box(double t1, double t2) const {
    if (objects.empty()) return std::nullopt;
    std::optional<Aabb> acc = objects[0]->bounding_box(t1, t2);
    for (size_t i = 1; acc && i < objects.size(); i++) {
        std::optional<Aabb> b = objects[i]->bounding_box(t1, t2);
        if (!b) return std::nullopt;
        acc = Aabb::surrounding_box(*acc, *b);
    }
    return acc;
}
std::optional<Aabb> Sphere::bounding_box(double, double) const {
    Vec3 r(radius, radius, radius);
    return Aabb{center - r, center + r};
}
Vec3 MovingSphere::get_center(double t) const { return center1 + (center2 - center1) * ((t - t1) / (t2 - t1)); }
std::optional<Aabb> MovingSphere::bounding_box(double ta, double tb) const {
    Vec3 r(radius, radius, radius);
    return Aabb::surrounding_box(Aabb{get_center(ta) - r, get_center(ta) + r}, Aabb{get_center(tb) - r, get_center(tb) + r});
}
std::optional<Aabb> Rect::bounding_box(double, double) const {
    switch (kind) {
        case RECT_XY: return Aabb{{a1, b1, k - 0.0001}, {a2, b2, k + 0.0001}};
        case RECT_XZ: return Aabb{{a1, k - 0.0001, b1}, {a2, k + 0.0001, b2}};
        default: return Aabb{{k - 0.0001, a1, b1}, {k + 0.0001, a2, b2}};
    }
}
std::optional<Aabb> Translate::bounding_box(double t1, double t2) const {
    std::optional<Aabb> b = object->bounding_box(t1, t2);
    if (!b) return std::nullopt;
    return Aabb{b->min + offset, b->max + offset};
}
RotateY::RotateY(ObjectPtr o, double angle_degrees) : Object(ROTATE_Y), object(std::move(o)) {
    const double radians = angle_degrees * kPi / 180.0;
    sin_theta = std::sin(radians);
    cos_theta = std::cos(radians);
    if (std::optional<Aabb> b = object->bounding_box(0.0, 1.0)) {  // transforms.rs:53-78: hull of the 8 rotated corners
        Vec3 lo(DBL_MAX, DBL_MAX, DBL_MAX), hi(-DBL_MAX, -DBL_MAX, -DBL_MAX);
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++)
                for (int k = 0; k < 2; k++) {
                    const double x = i * b->max.x + (1.0 - i) * b->min.x;
                    const double y = j * b->max.y + (1.0 - j) * b->min.y;
                    const double z = k * b->max.z + (1.0 - k) * b->min.z;
                    const Vec3 c(cos_theta * x + sin_theta * z, y, -sin_theta * x + cos_theta * z);
                    lo = {std::fmin(lo.x, c.x), std::fmin(lo.y, c.y), std::fmin(lo.z, c.z)};
                    hi = {std::fmax(hi.x, c.x), std::fmax(hi.y, c.y), std::fmax(hi.z, c.z)};
                }
        boxx = Aabb{lo, hi};
    }
}

ObjectPtr BvhNode::build(std::vector<ObjectPtr>& objects, double t1, double t2, SceneRng& rng) {
    if (objects.empty()) throw std::runtime_error("BvhNode::build on an empty list");
    const uint32_t axis = rng.gen_index(3);
    // comparators use bounding_box(0,0), not (t1,t2): bvh.rs:174-203
    std::vector<std::pair<double, ObjectPtr>> keyed;
    keyed.reserve(objects.size());
    for (auto& o : objects) {
        std::optional<Aabb> b = o->bounding_box(0.0, 0.0);
        if (!b) throw std::runtime_error("No bounding box in BvhNode::build.");
        const double key = axis == 0 ? b->min.x : axis == 1 ? b->min.y : b->min.z;
        if (key != key) throw std::runtime_error("No bounding box in BvhNode::build.");  // partial_cmp -> None
        keyed.emplace_back(key, o);
    }
    std::stable_sort(keyed.begin(), keyed.end(), [](const auto& l, const auto& r) { return l.first < r.first; });
    for (size_t i = 0; i < objects.size(); i++) objects[i] = keyed[i].second;
    if (objects.size() == 1) return objects[0];  // a single object IS the leaf (bvh.rs:97-99)
    const size_t half = objects.size() / 2;
    std::vector<ObjectPtr> upper(objects.begin() + half, objects.end());  // split_off(len/2)
    objects.resize(half);
    auto node = std::make_shared<BvhNode>();
    node->left = build(objects, t1, t2, rng);
    node->right = build(upper, t1, t2, rng);
    std::optional<Aabb> bl = node->left->bounding_box(t1, t2), br = node->right->bounding_box(t1, t2);
    if (!bl || !br) throw std::runtime_error("No bounding box in BvhNode::build.");
    node->boxx = Aabb::surrounding_box(*bl, *br);
    return node;
}

Camera::Camera(Vec3 lookfrom, Vec3 lookat, Vec3 vup, double vfov, double aspect_ratio, double aperture, double focus_dist,
               double t1_, double t2_) {
    const double theta = vfov * kPi / 180.0;
    const double viewport_height = 2.0 * std::tan(theta / 2.0);
    const double viewport_width = viewport_height * aspect_ratio;
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

}  // namespace rth
