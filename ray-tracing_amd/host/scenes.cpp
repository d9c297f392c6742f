// scenes.cpp — the reference's scene constructors (src/scenes.rs), constants kept one-for-one.
// Random draws are consumed in the reference's source order from one seeded SceneRng (stream 0).
#include "scene_api.hpp"

namespace rth {

namespace {

template <class T, class... A> std::shared_ptr<const T> mk(A&&... a) { return std::make_shared<T>(std::forward<A>(a)...); }
TexturePtr rgb(double r, double g, double b) { return mk<SolidColor>(Vec3(r, g, b)); }
TexturePtr rgb(Vec3 c) { return mk<SolidColor>(c); }
MaterialPtr matte(TexturePtr t) { return mk<Lambertian>(std::move(t)); }
MaterialPtr emitter(TexturePtr t) { return mk<DiffuseLight>(std::move(t)); }
Camera look(Vec3 from, Vec3 at, double vfov, double aspect, double aperture, double focus) {
    return Camera(from, at, Vec3(0.0, 1.0, 0.0), vfov, aspect, aperture, focus, 0.0, 1.0);
}

// the five walls + ceiling light shared by cornell_box (scenes.rs:563-640) and cornell_smoke (:705-782)
void add_cornell_shell(ObjectList& world, double lx1, double lx2, double lz1, double lz2, double power) {
    const MaterialPtr green = matte(rgb(0.12, 0.45, 0.15)), red = matte(rgb(0.65, 0.05, 0.05));
    world.add(mk<RectYZ>(0.0, 555.0, 0.0, 555.0, 555.0, -1.0, green));
    world.add(mk<RectYZ>(0.0, 555.0, 0.0, 555.0, 0.0, 1.0, red));
    world.add(mk<RectXZ>(lx1, lx2, lz1, lz2, 554.0, -1.0, emitter(rgb(power, power, power))));
    world.add(mk<RectXZ>(0.0, 555.0, 0.0, 555.0, 0.0, 1.0, matte(rgb(0.73, 0.73, 0.73))));
    world.add(mk<RectXZ>(0.0, 555.0, 0.0, 555.0, 555.0, -1.0, matte(rgb(0.73, 0.73, 0.73))));
    world.add(mk<RectXY>(0.0, 555.0, 0.0, 555.0, 555.0, -1.0, matte(rgb(0.73, 0.73, 0.73))));
}
ObjectPtr placed_box(Vec3 size, double degrees, Vec3 at) {
    return mk<Translate>(mk<RotateY>(mk<Cuboid>(Vec3(0.0, 0.0, 0.0), size, matte(rgb(0.73, 0.73, 0.73))), degrees), at);
}
ObjectPtr ground_checker_sphere() {  // scenes.rs:12-25, 176-189
    return mk<Sphere>(Vec3(0.0, -1000.0, 0.0), 1000.0, matte(mk<CheckerTexture>(rgb(0.2, 0.3, 0.1), rgb(0.9, 0.9, 0.9))));
}

}  // namespace

Scene random_scene(double aspect_ratio, uint64_t seed) {  // scenes.rs:7-110
    SceneRng rng(seed, 0);
    Scene s;
    s.world = std::make_shared<ObjectList>();
    s.world->add(ground_checker_sphere());
    std::vector<ObjectPtr> small;
    for (int a = -11; a < 11; a++) {
        for (int b = -11; b < 11; b++) {
            const double cx = a + 0.9 * rng.gen();
            const double cz = b + 0.9 * rng.gen();
            const Vec3 center(cx, 0.2, cz);
            if (!((center - Vec3(4.0, 0.2, 0.0)).length() > 0.9)) continue;
            const double pick = rng.gen();
            if (pick < 0.8) {
                const Vec3 center2 = center + Vec3(0.0, rng.gen_range(0.0, 0.5), 0.0);
                const Vec3 c1 = rng.random_vec3(0.0, 1.0);
                const Vec3 c2 = rng.random_vec3(0.0, 1.0);
                small.push_back(mk<MovingSphere>(center, center2, 0.0, 1.0, 0.2, matte(rgb(Vec3::elemul(c1, c2)))));
            } else if (pick < 0.95) {
                const Vec3 albedo = rng.random_vec3(0.0, 1.0);
                const double fuzz = rng.gen_range(0.0, 0.5);
                small.push_back(mk<Sphere>(center, 0.2, mk<Metal>(albedo, fuzz)));
            } else {
                small.push_back(mk<Sphere>(center, 0.2, mk<Dielectric>(1.5)));
            }
        }
    }
    s.world->add(BvhNode::build(small, 0.0, 1.0, rng));
    s.world->add(mk<Sphere>(Vec3(-4.0, 1.0, 0.0), 1.0, matte(rgb(0.4, 0.2, 0.1))));
    s.world->add(mk<Sphere>(Vec3(4.0, 1.0, 0.0), 1.0, mk<Metal>(Vec3(0.7, 0.6, 0.5), 0.0)));
    s.world->add(mk<Sphere>(Vec3(0.0, 1.0, 0.0), 1.0, mk<Dielectric>(1.5)));
    s.background = Vec3(0.7, 0.8, 1.0);
    s.cam = look(Vec3(13.0, 2.0, 3.0), Vec3(0.0, 0.0, 0.0), 20.0, aspect_ratio, 0.1, 10.0);
    return s;
}

Scene random_scene_light(double aspect_ratio, uint64_t seed) {  // scenes.rs:171-333
    SceneRng rng(seed, 0);
    Scene s;
    s.world = std::make_shared<ObjectList>();
    s.world->add(ground_checker_sphere());
    struct Ball { Vec3 c; double r; };
    std::vector<Ball> taken = {{Vec3(0.0, 0.8, 0.0), 0.8}, {Vec3(1.3, 0.5, 0.0), 0.5}, {Vec3(-1.3, 0.5, 0.0), 0.5}};
    std::vector<ObjectPtr> small;
    for (int a = -11; a < 11; a++) {
        for (int b = -11; b < 11; b++) {
            double radius;
            Vec3 center;
            bool clear;
            do {  // redraw until the new sphere overlaps none of the placed ones
                radius = rng.gen_range(0.08, 0.25);
                const double cx = (a + 0.9 * rng.gen()) / 2.0;
                const double cz = (b + 0.9 * rng.gen()) / 2.0;
                center = Vec3(cx, radius, cz);
                clear = true;
                for (const Ball& t : taken)
                    if ((t.c - center).length() < (t.r + radius)) { clear = false; break; }
            } while (!clear);
            taken.push_back({center, radius});
            if (!((center - Vec3(0.0, radius, 0.0)).length() > 1.3)) continue;
            const double pick = rng.gen();
            if (pick < 0.2) {
                small.push_back(mk<Sphere>(center, radius, matte(rgb(rng.random_vec3(0.1, 0.9)))));
            } else if (pick < 0.4) {
                const Vec3 albedo = rng.random_vec3(0.0, 1.0);
                const double fuzz = rng.gen_range(0.0, 0.5);
                small.push_back(mk<Sphere>(center, radius, mk<Metal>(albedo, fuzz)));
            } else if (pick < 0.6) {
                small.push_back(mk<Sphere>(center, radius, mk<Dielectric>(rng.gen_range(1.5, 2.0))));
            } else if (pick < 0.8) {
                small.push_back(mk<Sphere>(center, radius * 0.9, emitter(rgb(rng.random_vec3(0.1, 0.9)))));
                small.push_back(mk<Sphere>(center, radius, mk<Dielectric>(1.5)));
            } else {
                small.push_back(mk<Sphere>(center, radius * 0.5, emitter(rgb(rng.random_vec3(0.1, 0.9)))));
                small.push_back(mk<Sphere>(center, radius, mk<FrostedDielectric>(1.5, rng.gen_range(0.3, 0.5))));
            }
        }
    }
    s.world->add(BvhNode::build(small, 0.0, 1.0, rng));
    s.world->add(mk<Sphere>(Vec3(0.0, 0.8, 0.0), 0.8, emitter(mk<CheckerTexture>(rgb(1.0, 0.6, 0.4), rgb(1.0, 0.75, 0.3)))));
    s.world->add(mk<Sphere>(Vec3(1.3, 0.5, 0.0), 0.5, mk<Dielectric>(1.5)));
    s.world->add(mk<Sphere>(Vec3(-1.3, 0.5, 0.0), 0.5, mk<Metal>(Vec3(0.8, 0.6, 0.3), 0.0)));
    s.background = Vec3::zero();
    s.cam = look(Vec3(6.0, 3.0, 6.0), Vec3(0.0, 0.0, 0.0), 40.0, aspect_ratio, 0.05, 9.0);
    return s;
}

Scene two_checker_spheres(double aspect_ratio) {  // scenes.rs:399-444
    Scene s;
    s.world = std::make_shared<ObjectList>();
    s.world->add(mk<Sphere>(Vec3(0.0, -10.0, 0.0), 10.0, matte(mk<CheckerTextureUV>(rgb(0.2, 0.3, 0.1), rgb(0.9, 0.9, 0.9)))));
    s.world->add(mk<Sphere>(Vec3(0.0, 10.0, 0.0), 10.0, matte(mk<CheckerTexture>(rgb(0.2, 0.3, 0.1), rgb(0.9, 0.9, 0.9)))));
    s.background = Vec3(0.7, 0.8, 1.0);
    s.cam = look(Vec3(13.0, 2.0, 3.0), Vec3(0.0, 0.0, 0.0), 20.0, aspect_ratio, 0.0, 10.0);
    return s;
}

Scene two_perlin_spheres(double aspect_ratio, uint64_t seed) {  // scenes.rs:445-482; each sphere has its OWN Perlin::new()
    SceneRng rng(seed, 0);
    Scene s;
    s.world = std::make_shared<ObjectList>();
    auto ground_noise = mk<Perlin>(rng);
    s.world->add(mk<Sphere>(Vec3(0.0, -1000.0, 0.0), 1000.0, matte(mk<NoiseTexture>(ground_noise, 4.0))));
    auto ball_noise = mk<Perlin>(rng);
    s.world->add(mk<Sphere>(Vec3(0.0, 2.0, 0.0), 2.0, matte(mk<NoiseTexture>(ball_noise, 4.0))));
    s.background = Vec3(0.7, 0.8, 1.0);
    s.cam = look(Vec3(13.0, 2.0, 3.0), Vec3(0.0, 0.0, 0.0), 20.0, aspect_ratio, 0.0, 10.0);
    return s;
}

Scene earth(double aspect_ratio, const std::string& image_path) {  // scenes.rs:483-507
    Scene s;
    s.world = std::make_shared<ObjectList>();
    s.world->add(mk<Sphere>(Vec3(0.0, 0.0, 0.0), 2.0, matte(mk<ImageTexture>(image_path))));
    s.background = Vec3(0.7, 0.8, 1.0);
    s.cam = look(Vec3(13.0, 2.0, 3.0), Vec3(0.0, 0.0, 0.0), 20.0, aspect_ratio, 0.0, 10.0);
    return s;
}

Scene simple_light(double aspect_ratio, uint64_t seed) {  // scenes.rs:508-558
    SceneRng rng(seed, 0);
    Scene s;
    s.world = std::make_shared<ObjectList>();
    auto ground_noise = mk<Perlin>(rng);
    s.world->add(mk<Sphere>(Vec3(0.0, -1000.0, 0.0), 1000.0, matte(mk<NoiseTexture>(ground_noise, 4.0))));
    auto ball_noise = mk<Perlin>(rng);
    s.world->add(mk<Sphere>(Vec3(0.0, 2.0, 0.0), 2.0, matte(mk<NoiseTexture>(ball_noise, 4.0))));
    s.world->add(mk<RectXY>(3.0, 5.0, 1.0, 3.0, -2.0, 1.0, emitter(rgb(4.0, 4.0, 4.0))));
    s.background = Vec3::zero();
    s.cam = look(Vec3(26.0, 3.0, 6.0), Vec3(0.0, 2.0, 0.0), 20.0, aspect_ratio, 0.0, 26.7);
    return s;
}

Scene cornell_box(double aspect_ratio) {  // scenes.rs:559-702
    Scene s;
    s.world = std::make_shared<ObjectList>();
    add_cornell_shell(*s.world, 213.0, 343.0, 227.0, 332.0, 15.0);
    s.world->add(placed_box(Vec3(165.0, 330.0, 165.0), 15.0, Vec3(265.0, 0.0, 295.0)));
    // a glass SPHERE, not the book's short box (scenes.rs:656-663)
    s.world->add(mk<Translate>(mk<Sphere>(Vec3(0.0, 90.0, 0.0), 90.0, mk<Dielectric>(1.5)), Vec3(190.0, 0.0, 190.0)));
    s.lights = std::make_shared<ObjectList>();  // scenes.rs:664-685: separate copies of the light rect and the glass sphere
    s.lights->add(mk<RectXZ>(213.0, 343.0, 227.0, 332.0, 554.0, -1.0, emitter(rgb(15.0, 15.0, 15.0))));
    s.lights->add(mk<Translate>(mk<Sphere>(Vec3(0.0, 90.0, 0.0), 90.0, mk<Dielectric>(1.5)), Vec3(190.0, 0.0, 190.0)));
    s.background = Vec3::zero();
    s.cam = look(Vec3(278.0, 278.0, -800.0), Vec3(278.0, 278.0, 0.0), 40.0, aspect_ratio, 0.0, 28.3);
    return s;
}

Scene cornell_smoke(double aspect_ratio) {  // scenes.rs:703-840
    Scene s;
    s.world = std::make_shared<ObjectList>();
    add_cornell_shell(*s.world, 113.0, 443.0, 127.0, 432.0, 7.0);
    s.world->add(mk<ConstantMedium>(placed_box(Vec3(165.0, 330.0, 165.0), 15.0, Vec3(265.0, 0.0, 295.0)), rgb(Vec3::zero()), 0.01));
    s.world->add(mk<ConstantMedium>(placed_box(Vec3(165.0, 165.0, 165.0), -18.0, Vec3(130.0, 0.0, 65.0)), rgb(Vec3::ones()), 0.01));
    s.background = Vec3::zero();
    s.cam = look(Vec3(278.0, 278.0, -800.0), Vec3(278.0, 278.0, 0.0), 40.0, aspect_ratio, 0.0, 28.3);
    return s;
}

Scene final_scene(double aspect_ratio, uint64_t seed, const std::string& image_path) {  // scenes.rs:841-1014
    SceneRng rng(seed, 0);
    Scene s;
    s.world = std::make_shared<ObjectList>();
    std::vector<ObjectPtr> ground;
    const MaterialPtr grass = matte(rgb(0.48, 0.83, 0.53));
    for (int i = 0; i < 20; i++) {
        for (int j = 0; j < 20; j++) {
            const double w = 100.0;
            const double x1 = -1000.0 + i * w, y1 = 0.0, z1 = -1000.0 + j * w;
            const double x2 = x1 + w;
            const double y2 = rng.gen_range(1.0, 101.0);
            const double z2 = z1 + w;
            ground.push_back(mk<Cuboid>(Vec3(x1, y1, z1), Vec3(x2, y2, z2), grass));
        }
    }
    s.world->add(BvhNode::build(ground, 0.0, 1.0, rng));
    s.world->add(mk<RectXZ>(123.0, 423.0, 147.0, 412.0, 554.0, -1.0, emitter(rgb(7.0, 7.0, 7.0))));
    s.world->add(mk<MovingSphere>(Vec3(400.0, 400.0, 200.0), Vec3(400.0, 400.0, 200.0) + Vec3(30.0, 0.0, 0.0), 0.0, 1.0, 50.0,
                                  matte(rgb(0.7, 0.3, 0.1))));
    s.world->add(mk<Sphere>(Vec3(260.0, 150.0, 45.0), 50.0, mk<Dielectric>(1.5)));
    s.world->add(mk<Sphere>(Vec3(0.0, 150.0, 145.0), 50.0, mk<Metal>(Vec3(0.8, 0.8, 0.9), 1.0)));
    s.world->add(mk<Sphere>(Vec3(360.0, 150.0, 145.0), 70.0, mk<Dielectric>(1.5)));
    s.world->add(mk<ConstantMedium>(mk<Sphere>(Vec3(360.0, 150.0, 145.0), 70.0, mk<Dielectric>(1.5)), rgb(0.2, 0.4, 0.9), 0.2));
    s.world->add(mk<ConstantMedium>(mk<Sphere>(Vec3(0.0, 0.0, 0.0), 5000.0, mk<Dielectric>(1.5)), rgb(Vec3::ones()), 0.0001));
    s.world->add(mk<Sphere>(Vec3(400.0, 200.0, 400.0), 100.0, matte(mk<ImageTexture>(image_path))));
    auto marble = mk<Perlin>(rng);
    s.world->add(mk<Sphere>(Vec3(220.0, 280.0, 300.0), 80.0, matte(mk<NoiseTexture>(marble, 0.1))));
    std::vector<ObjectPtr> cluster;
    const MaterialPtr white = matte(rgb(0.73, 0.73, 0.73));
    for (int i = 0; i < 1000; i++) cluster.push_back(mk<Sphere>(rng.random_vec3(0.0, 165.0), 10.0, white));
    s.world->add(mk<Translate>(mk<RotateY>(mk<Bvh>(cluster, 0.0, 1.0, rng), 15.0), Vec3(-100.0, 270.0, 395.0)));
    s.lights = std::make_shared<ObjectList>();  // scenes.rs:966-997
    s.lights->add(mk<RectXZ>(123.0, 423.0, 147.0, 412.0, 554.0, -1.0, emitter(rgb(7.0, 7.0, 7.0))));
    s.lights->add(mk<Sphere>(Vec3(260.0, 150.0, 45.0), 50.0, mk<Dielectric>(1.5)));
    s.lights->add(mk<Sphere>(Vec3(0.0, 150.0, 145.0), 50.0, mk<Metal>(Vec3(0.8, 0.8, 0.9), 1.0)));
    s.lights->add(mk<Sphere>(Vec3(360.0, 150.0, 145.0), 70.0, mk<Dielectric>(1.5)));
    s.background = Vec3::zero();
    s.cam = look(Vec3(478.0, 278.0, -600.0), Vec3(278.0, 278.0, 0.0), 40.0, aspect_ratio, 0.0, 28.3);
    return s;
}

}  // namespace rth
