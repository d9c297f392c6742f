// ORACLE — TEST INFRASTRUCTURE ONLY.
// Scene constructors restating reference src/scenes.rs with its exact constants:
//   random_scene :7-110, random_scene_light :171-333, two_checker_spheres :399-444,
//   two_perlin_spheres :445-482, earth :483-507, simple_light :508-558, cornell_box :559-702,
//   cornell_smoke :703-840, final_scene :841-1014.
// The reference seeds every generator from entropy (scenes.rs:10,174,844); the oracle uses the
// seeded scene generator (rng.hpp, stream 0) and consumes draws in the reference's source order.
// 3-tuple constructors (no lights) are normalised to lights = None.
#include <algorithm>
#include <cstring>
#include "ray_tracing.hpp"

namespace orc {

thread_local Counters g_cnt;
thread_local CastCtx g_cast;
thread_local unsigned long long g_unit_zero_events = 0;

// bvh.rs:85-111 + comparators :174-203
ObjPtr BvhNode::build(std::vector<ObjPtr>& objects, double t1, double t2, Rng& rng) {
    uint32_t axis = rng.gen_index(3);
    auto key = [axis](const ObjPtr& o) {
        Aabb b;
        if (!o->bounding_box(0.0, 0.0, b)) throw std::runtime_error("No bounding box in BvhNode::build.");
        return axis == 0 ? b.min.x : axis == 1 ? b.min.y : b.min.z;
    };
    std::stable_sort(objects.begin(), objects.end(), [&](const ObjPtr& a, const ObjPtr& b) { return key(a) < key(b); });
    size_t len = objects.size();
    if (len == 1) return objects[0];
    std::vector<ObjPtr> objects2(objects.begin() + len / 2, objects.end());
    objects.resize(len / 2);
    auto node = std::make_shared<BvhNode>();
    node->left = build(objects, t1, t2, rng);
    node->right = build(objects2, t1, t2, rng);
    Aabb bl, br;
    if (!node->left->bounding_box(t1, t2, bl) || !node->right->bounding_box(t1, t2, br))
        throw std::runtime_error("No bounding box in BvhNode::build.");
    node->boxx = Aabb::surrounding_box(bl, br);
    return node;
}

static TexPtr solid(double r, double g, double b) { return std::make_shared<SolidColor>(Vec3(r, g, b)); }
static std::shared_ptr<Material> lambert(TexPtr t) { return std::make_shared<Lambertian>(std::move(t)); }
static std::shared_ptr<Material> light(TexPtr t) { return std::make_shared<DiffuseLight>(std::move(t)); }
static std::shared_ptr<Material> glass(double r) { return std::make_shared<Dielectric>(r); }
static std::shared_ptr<Rect> rect_xy(double x1, double x2, double y1, double y2, double k, double face, std::shared_ptr<Material> m) {
    return std::make_shared<Rect>(0, x1, x2, y1, y2, k, face, std::move(m));
}
static std::shared_ptr<Rect> rect_xz(double x1, double x2, double z1, double z2, double k, double face, std::shared_ptr<Material> m) {
    return std::make_shared<Rect>(1, x1, x2, z1, z2, k, face, std::move(m));
}
static std::shared_ptr<Rect> rect_yz(double y1, double y2, double z1, double z2, double k, double face, std::shared_ptr<Material> m) {
    return std::make_shared<Rect>(2, y1, y2, z1, z2, k, face, std::move(m));
}
static ObjPtr medium(ObjPtr boundary, TexPtr tex, double density) {
    return std::make_shared<ConstantMedium>(std::move(boundary), std::make_shared<Isotropic>(std::move(tex)), density);
}
static Camera std_camera(Vec3 from, Vec3 at, double vfov, double aspect, double aperture, double focus) {
    return Camera(from, at, Vec3(0.0, 1.0, 0.0), vfov, aspect, aperture, focus, 0.0, 1.0);
}

static Scene random_scene(double aspect, uint64_t seed) {
    Rng rng = Rng::for_scene(seed, 0);
    Scene sc;
    sc.world = std::make_shared<ObjectList>();
    sc.world->add(std::make_shared<Sphere>(Vec3(0.0, -1000.0, 0.0), 1000.0,
                                           lambert(std::make_shared<CheckerTexture>(solid(0.2, 0.3, 0.1), solid(0.9, 0.9, 0.9)))));
    std::vector<ObjPtr> box1;
    for (int a = -11; a < 11; a++)
        for (int b = -11; b < 11; b++) {
            double cx = (double)a + 0.9 * rng.gen();
            double cz = (double)b + 0.9 * rng.gen();
            Vec3 center(cx, 0.2, cz);
            if ((center - Vec3(4.0, 0.2, 0.0)).length() > 0.9) {
                double rd = rng.gen();
                if (rd < 0.8) {
                    Vec3 c2 = center + Vec3(0.0, rng.gen_range(0.0, 0.5), 0.0);
                    Vec3 r1 = Vec3::random(0.0, 1.0, rng);
                    Vec3 r2 = Vec3::random(0.0, 1.0, rng);
                    Vec3 col = Vec3::elemul(r1, r2);
                    box1.push_back(std::make_shared<MovingSphere>(center, c2, 0.0, 1.0, 0.2, lambert(solid(col.x, col.y, col.z))));
                } else if (rd < 0.95) {
                    Vec3 alb = Vec3::random(0.0, 1.0, rng);
                    double fuzz = rng.gen_range(0.0, 0.5);
                    box1.push_back(std::make_shared<Sphere>(center, 0.2, std::make_shared<Metal>(alb, fuzz)));
                } else {
                    box1.push_back(std::make_shared<Sphere>(center, 0.2, glass(1.5)));
                }
            }
        }
    sc.world->add(BvhNode::build(box1, 0.0, 1.0, rng));
    sc.world->add(std::make_shared<Sphere>(Vec3(-4.0, 1.0, 0.0), 1.0, lambert(solid(0.4, 0.2, 0.1))));
    sc.world->add(std::make_shared<Sphere>(Vec3(4.0, 1.0, 0.0), 1.0, std::make_shared<Metal>(Vec3(0.7, 0.6, 0.5), 0.0)));
    sc.world->add(std::make_shared<Sphere>(Vec3(0.0, 1.0, 0.0), 1.0, glass(1.5)));
    sc.background = Vec3(0.7, 0.8, 1.0);
    sc.cam = std_camera(Vec3(13.0, 2.0, 3.0), Vec3(0.0, 0.0, 0.0), 20.0, aspect, 0.1, 10.0);
    return sc;
}

static Scene random_scene_light(double aspect, uint64_t seed) {
    Rng rng = Rng::for_scene(seed, 0);
    Scene sc;
    sc.world = std::make_shared<ObjectList>();
    sc.world->add(std::make_shared<Sphere>(Vec3(0.0, -1000.0, 0.0), 1000.0,
                                           lambert(std::make_shared<CheckerTexture>(solid(0.2, 0.3, 0.1), solid(0.9, 0.9, 0.9)))));
    std::vector<ObjPtr> box1;
    std::vector<std::pair<Vec3, double>> placed = {{Vec3(0.0, 0.8, 0.0), 0.8}, {Vec3(1.3, 0.5, 0.0), 0.5}, {Vec3(-1.3, 0.5, 0.0), 0.5}};
    for (int a = -11; a < 11; a++)
        for (int b = -11; b < 11; b++) {
            auto draw = [&](double& radius, Vec3& center) {
                radius = rng.gen_range(0.08, 0.25);
                double cx = ((double)a + 0.9 * rng.gen()) / 2.0;
                double cz = ((double)b + 0.9 * rng.gen()) / 2.0;
                center = Vec3(cx, radius, cz);
            };
            double radius;
            Vec3 center;
            draw(radius, center);
            for (;;) {
                bool done = true;
                for (auto& cr : placed)
                    if ((cr.first - center).length() < (cr.second + radius)) { done = false; break; }
                if (done) break;
                draw(radius, center);
            }
            placed.push_back({center, radius});
            if ((center - Vec3(0.0, radius, 0.0)).length() > 1.3) {
                double rd = rng.gen();
                if (rd < 0.2) {
                    Vec3 c = Vec3::random(0.1, 0.9, rng);
                    box1.push_back(std::make_shared<Sphere>(center, radius, lambert(solid(c.x, c.y, c.z))));
                } else if (rd < 0.4) {
                    Vec3 alb = Vec3::random(0.0, 1.0, rng);
                    double fuzz = rng.gen_range(0.0, 0.5);
                    box1.push_back(std::make_shared<Sphere>(center, radius, std::make_shared<Metal>(alb, fuzz)));
                } else if (rd < 0.6) {
                    box1.push_back(std::make_shared<Sphere>(center, radius, glass(rng.gen_range(1.5, 2.0))));
                } else if (rd < 0.8) {
                    Vec3 c = Vec3::random(0.1, 0.9, rng);
                    box1.push_back(std::make_shared<Sphere>(center, radius * 0.9, light(solid(c.x, c.y, c.z))));
                    box1.push_back(std::make_shared<Sphere>(center, radius, glass(1.5)));
                } else {
                    Vec3 c = Vec3::random(0.1, 0.9, rng);
                    box1.push_back(std::make_shared<Sphere>(center, radius * 0.5, light(solid(c.x, c.y, c.z))));
                    box1.push_back(std::make_shared<Sphere>(center, radius, std::make_shared<FrostedDielectric>(1.5, rng.gen_range(0.3, 0.5))));
                }
            }
        }
    sc.world->add(BvhNode::build(box1, 0.0, 1.0, rng));
    sc.world->add(std::make_shared<Sphere>(Vec3(0.0, 0.8, 0.0), 0.8,
                                           light(std::make_shared<CheckerTexture>(solid(1.0, 0.6, 0.4), solid(1.0, 0.75, 0.3)))));
    sc.world->add(std::make_shared<Sphere>(Vec3(1.3, 0.5, 0.0), 0.5, glass(1.5)));
    sc.world->add(std::make_shared<Sphere>(Vec3(-1.3, 0.5, 0.0), 0.5, std::make_shared<Metal>(Vec3(0.8, 0.6, 0.3), 0.0)));
    sc.background = Vec3::zero();
    sc.cam = std_camera(Vec3(6.0, 3.0, 6.0), Vec3(0.0, 0.0, 0.0), 40.0, aspect, 0.05, 9.0);
    return sc;
}

static Scene two_checker_spheres(double aspect, uint64_t) {
    Scene sc;
    sc.world = std::make_shared<ObjectList>();
    sc.world->add(std::make_shared<Sphere>(Vec3(0.0, -10.0, 0.0), 10.0,
                                           lambert(std::make_shared<CheckerTextureUV>(solid(0.2, 0.3, 0.1), solid(0.9, 0.9, 0.9)))));
    sc.world->add(std::make_shared<Sphere>(Vec3(0.0, 10.0, 0.0), 10.0,
                                           lambert(std::make_shared<CheckerTexture>(solid(0.2, 0.3, 0.1), solid(0.9, 0.9, 0.9)))));
    sc.background = Vec3(0.7, 0.8, 1.0);
    sc.cam = std_camera(Vec3(13.0, 2.0, 3.0), Vec3(0.0, 0.0, 0.0), 20.0, aspect, 0.0, 10.0);
    return sc;
}

static Scene two_perlin_spheres(double aspect, uint64_t seed) {
    Rng rng = Rng::for_scene(seed, 0);
    Scene sc;
    sc.world = std::make_shared<ObjectList>();
    auto n1 = std::make_shared<Perlin>(rng);
    sc.world->add(std::make_shared<Sphere>(Vec3(0.0, -1000.0, 0.0), 1000.0, lambert(std::make_shared<NoiseTexture>(n1, 4.0))));
    auto n2 = std::make_shared<Perlin>(rng);
    sc.world->add(std::make_shared<Sphere>(Vec3(0.0, 2.0, 0.0), 2.0, lambert(std::make_shared<NoiseTexture>(n2, 4.0))));
    sc.background = Vec3(0.7, 0.8, 1.0);
    sc.cam = std_camera(Vec3(13.0, 2.0, 3.0), Vec3(0.0, 0.0, 0.0), 20.0, aspect, 0.0, 10.0);
    return sc;
}

static Scene earth(double aspect, uint64_t, const std::string& image_path) {
    Scene sc;
    sc.world = std::make_shared<ObjectList>();
    sc.world->add(std::make_shared<Sphere>(Vec3(0.0, 0.0, 0.0), 2.0,
                                           lambert(std::make_shared<ImageTexture>(RgbImage::load_ppm(image_path)))));
    sc.background = Vec3(0.7, 0.8, 1.0);
    sc.cam = std_camera(Vec3(13.0, 2.0, 3.0), Vec3(0.0, 0.0, 0.0), 20.0, aspect, 0.0, 10.0);
    return sc;
}

static Scene simple_light(double aspect, uint64_t seed) {
    Rng rng = Rng::for_scene(seed, 0);
    Scene sc;
    sc.world = std::make_shared<ObjectList>();
    auto n1 = std::make_shared<Perlin>(rng);
    sc.world->add(std::make_shared<Sphere>(Vec3(0.0, -1000.0, 0.0), 1000.0, lambert(std::make_shared<NoiseTexture>(n1, 4.0))));
    auto n2 = std::make_shared<Perlin>(rng);
    sc.world->add(std::make_shared<Sphere>(Vec3(0.0, 2.0, 0.0), 2.0, lambert(std::make_shared<NoiseTexture>(n2, 4.0))));
    sc.world->add(rect_xy(3.0, 5.0, 1.0, 3.0, -2.0, 1.0, light(solid(4.0, 4.0, 4.0))));
    sc.background = Vec3::zero();
    sc.cam = std_camera(Vec3(26.0, 3.0, 6.0), Vec3(0.0, 2.0, 0.0), 20.0, aspect, 0.0, 26.7);
    return sc;
}

static void cornell_walls(ObjectList& w, double lx1, double lx2, double lz1, double lz2, double emit) {
    w.add(rect_yz(0.0, 555.0, 0.0, 555.0, 555.0, -1.0, lambert(solid(0.12, 0.45, 0.15))));
    w.add(rect_yz(0.0, 555.0, 0.0, 555.0, 0.0, 1.0, lambert(solid(0.65, 0.05, 0.05))));
    w.add(rect_xz(lx1, lx2, lz1, lz2, 554.0, -1.0, light(solid(emit, emit, emit))));
    w.add(rect_xz(0.0, 555.0, 0.0, 555.0, 0.0, 1.0, lambert(solid(0.73, 0.73, 0.73))));
    w.add(rect_xz(0.0, 555.0, 0.0, 555.0, 555.0, -1.0, lambert(solid(0.73, 0.73, 0.73))));
    w.add(rect_xy(0.0, 555.0, 0.0, 555.0, 555.0, -1.0, lambert(solid(0.73, 0.73, 0.73))));
}

static Scene cornell_box(double aspect, uint64_t) {
    Scene sc;
    sc.world = std::make_shared<ObjectList>();
    cornell_walls(*sc.world, 213.0, 343.0, 227.0, 332.0, 15.0);
    sc.world->add(std::make_shared<Translate>(
        std::make_shared<RotateY>(std::make_shared<Cuboid>(Vec3(0.0, 0.0, 0.0), Vec3(165.0, 330.0, 165.0), lambert(solid(0.73, 0.73, 0.73))), 15.0),
        Vec3(265.0, 0.0, 295.0)));
    sc.world->add(std::make_shared<Translate>(std::make_shared<Sphere>(Vec3(0.0, 90.0, 0.0), 90.0, glass(1.5)), Vec3(190.0, 0.0, 190.0)));
    sc.lights = std::make_shared<ObjectList>();
    sc.lights->add(rect_xz(213.0, 343.0, 227.0, 332.0, 554.0, -1.0, light(solid(15.0, 15.0, 15.0))));
    sc.lights->add(std::make_shared<Translate>(std::make_shared<Sphere>(Vec3(0.0, 90.0, 0.0), 90.0, glass(1.5)), Vec3(190.0, 0.0, 190.0)));
    sc.background = Vec3::zero();
    sc.cam = std_camera(Vec3(278.0, 278.0, -800.0), Vec3(278.0, 278.0, 0.0), 40.0, aspect, 0.0, 28.3);
    return sc;
}

static Scene cornell_smoke(double aspect, uint64_t) {
    Scene sc;
    sc.world = std::make_shared<ObjectList>();
    cornell_walls(*sc.world, 113.0, 443.0, 127.0, 432.0, 7.0);
    sc.world->add(medium(std::make_shared<Translate>(
                             std::make_shared<RotateY>(std::make_shared<Cuboid>(Vec3(0.0, 0.0, 0.0), Vec3(165.0, 330.0, 165.0), lambert(solid(0.73, 0.73, 0.73))), 15.0),
                             Vec3(265.0, 0.0, 295.0)),
                         solid(0.0, 0.0, 0.0), 0.01));
    sc.world->add(medium(std::make_shared<Translate>(
                             std::make_shared<RotateY>(std::make_shared<Cuboid>(Vec3(0.0, 0.0, 0.0), Vec3(165.0, 165.0, 165.0), lambert(solid(0.73, 0.73, 0.73))), -18.0),
                             Vec3(130.0, 0.0, 65.0)),
                         solid(1.0, 1.0, 1.0), 0.01));
    sc.background = Vec3::zero();
    sc.cam = std_camera(Vec3(278.0, 278.0, -800.0), Vec3(278.0, 278.0, 0.0), 40.0, aspect, 0.0, 28.3);
    return sc;
}

static Scene final_scene(double aspect, uint64_t seed, const std::string& image_path) {
    Rng rng = Rng::for_scene(seed, 0);
    Scene sc;
    sc.world = std::make_shared<ObjectList>();
    std::vector<ObjPtr> box1;
    for (int i = 0; i < 20; i++)
        for (int j = 0; j < 20; j++) {
            double w = 100.0;
            double x1 = -1000.0 + (double)i * w;
            double y1 = 0.0;
            double z1 = -1000.0 + (double)j * w;
            double x2 = x1 + w;
            double y2 = rng.gen_range(1.0, 101.0);
            double z2 = z1 + w;
            box1.push_back(std::make_shared<Cuboid>(Vec3(x1, y1, z1), Vec3(x2, y2, z2), lambert(solid(0.48, 0.83, 0.53))));
        }
    sc.world->add(BvhNode::build(box1, 0.0, 1.0, rng));
    sc.world->add(rect_xz(123.0, 423.0, 147.0, 412.0, 554.0, -1.0, light(solid(7.0, 7.0, 7.0))));
    sc.world->add(std::make_shared<MovingSphere>(Vec3(400.0, 400.0, 200.0), Vec3(400.0, 400.0, 200.0) + Vec3(30.0, 0.0, 0.0), 0.0, 1.0,
                                                 50.0, lambert(solid(0.7, 0.3, 0.1))));
    sc.world->add(std::make_shared<Sphere>(Vec3(260.0, 150.0, 45.0), 50.0, glass(1.5)));
    sc.world->add(std::make_shared<Sphere>(Vec3(0.0, 150.0, 145.0), 50.0, std::make_shared<Metal>(Vec3(0.8, 0.8, 0.9), 1.0)));
    sc.world->add(std::make_shared<Sphere>(Vec3(360.0, 150.0, 145.0), 70.0, glass(1.5)));
    sc.world->add(medium(std::make_shared<Sphere>(Vec3(360.0, 150.0, 145.0), 70.0, glass(1.5)), solid(0.2, 0.4, 0.9), 0.2));
    sc.world->add(medium(std::make_shared<Sphere>(Vec3(0.0, 0.0, 0.0), 5000.0, glass(1.5)), solid(1.0, 1.0, 1.0), 0.0001));
    sc.world->add(std::make_shared<Sphere>(Vec3(400.0, 200.0, 400.0), 100.0,
                                           lambert(std::make_shared<ImageTexture>(RgbImage::load_ppm(image_path)))));
    auto pn = std::make_shared<Perlin>(rng);
    sc.world->add(std::make_shared<Sphere>(Vec3(220.0, 280.0, 300.0), 80.0, lambert(std::make_shared<NoiseTexture>(pn, 0.1))));
    std::vector<ObjPtr> box2;
    for (int i = 0; i < 1000; i++)
        box2.push_back(std::make_shared<Sphere>(Vec3::random(0.0, 165.0, rng), 10.0, lambert(solid(0.73, 0.73, 0.73))));
    sc.world->add(std::make_shared<Translate>(std::make_shared<RotateY>(BvhNode::build(box2, 0.0, 1.0, rng), 15.0), Vec3(-100.0, 270.0, 395.0)));
    sc.lights = std::make_shared<ObjectList>();
    sc.lights->add(rect_xz(123.0, 423.0, 147.0, 412.0, 554.0, -1.0, light(solid(7.0, 7.0, 7.0))));
    sc.lights->add(std::make_shared<Sphere>(Vec3(260.0, 150.0, 45.0), 50.0, glass(1.5)));
    sc.lights->add(std::make_shared<Sphere>(Vec3(0.0, 150.0, 145.0), 50.0, std::make_shared<Metal>(Vec3(0.8, 0.8, 0.9), 1.0)));
    sc.lights->add(std::make_shared<Sphere>(Vec3(360.0, 150.0, 145.0), 70.0, glass(1.5)));
    sc.background = Vec3::zero();
    sc.cam = std_camera(Vec3(478.0, 278.0, -600.0), Vec3(278.0, 278.0, 0.0), 40.0, aspect, 0.0, 28.3);
    return sc;
}

Scene build_scene(const std::string& name, double aspect, uint64_t seed, const std::string& image_path) {
    Scene sc;
    if (name == "random_scene") sc = random_scene(aspect, seed);
    else if (name == "random_scene_light") sc = random_scene_light(aspect, seed);
    else if (name == "two_checker_spheres") sc = two_checker_spheres(aspect, seed);
    else if (name == "two_perlin_spheres") sc = two_perlin_spheres(aspect, seed);
    else if (name == "earth") sc = earth(aspect, seed, image_path);
    else if (name == "simple_light") sc = simple_light(aspect, seed);
    else if (name == "cornell_box") sc = cornell_box(aspect, seed);
    else if (name == "cornell_smoke") sc = cornell_smoke(aspect, seed);
    else if (name == "final_scene") sc = final_scene(aspect, seed, image_path);
    else throw std::runtime_error("unknown scene: " + name);
    sc.finalize();
    return sc;
}

}  // namespace orc
