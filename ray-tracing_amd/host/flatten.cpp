// flatten.cpp — see flatten.hpp for the rules.
#include "flatten.hpp"

#include <cstring>

namespace rth {

void FlatScene::seal() {
    desc.abi_version = RT_ABI_VERSION;
    desc.n_prims = (uint32_t)prims.size();         desc.prims = prims.data();
    desc.n_materials = (uint32_t)materials.size(); desc.materials = materials.data();
    desc.n_textures = (uint32_t)textures.size();   desc.textures = textures.data();
    desc.n_perlins = (uint32_t)perlins.size();     desc.perlins = perlins.data();
    desc.n_images = (uint32_t)images.size();       desc.images = images.data();
    desc.n_xforms = (uint32_t)xforms.size();       desc.xforms = xforms.data();
    desc.n_world = (uint32_t)world.size();         desc.world = world.data();
    desc.n_lights = (uint32_t)lights.size();       desc.lights = lights.data();
    desc.bvh.n_nodes = (uint32_t)left.size();
    desc.bvh.min_x = min_x.data(); desc.bvh.min_y = min_y.data(); desc.bvh.min_z = min_z.data();
    desc.bvh.max_x = max_x.data(); desc.bvh.max_y = max_y.data(); desc.bvh.max_z = max_z.data();
    desc.bvh.left = left.data();   desc.bvh.right = right.data();
}

namespace {

struct Flattener {
    FlatScene& f;
    std::map<const Texture*, uint32_t> tex_ids;
    std::map<const Material*, uint32_t> mat_ids;
    std::map<const Perlin*, uint32_t> perlin_ids;
    std::map<const RgbImage*, uint32_t> image_ids;

    uint32_t texture(const TexturePtr& t) {
        auto it = tex_ids.find(t.get());
        if (it != tex_ids.end()) return it->second;
        const uint32_t id = (uint32_t)f.textures.size();
        tex_ids[t.get()] = id;
        f.textures.push_back(RtTexture{});
        RtTexture r{};
        switch (t->kind) {
            case Texture::SOLID: {
                auto& s = static_cast<const SolidColor&>(*t);
                r.kind = RT_TEX_SOLID; r.color[0] = s.color.x; r.color[1] = s.color.y; r.color[2] = s.color.z;
                break;
            }
            case Texture::CHECKER: {
                auto& c = static_cast<const CheckerTexture&>(*t);
                r.kind = RT_TEX_CHECKER_3D; r.odd = texture(c.odd); r.even = texture(c.even);
                break;
            }
            case Texture::CHECKER_UV: {
                auto& c = static_cast<const CheckerTextureUV&>(*t);
                r.kind = RT_TEX_CHECKER_UV; r.odd = texture(c.odd); r.even = texture(c.even);
                break;
            }
            case Texture::NOISE: {
                auto& n = static_cast<const NoiseTexture&>(*t);
                r.kind = RT_TEX_NOISE; r.scale = n.scale;
                auto pit = perlin_ids.find(n.noise.get());
                if (pit == perlin_ids.end()) {
                    RtPerlin p{};
                    for (int i = 0; i < 256; i++) {
                        p.perm_x[i] = n.noise->perm_x[i]; p.perm_y[i] = n.noise->perm_y[i]; p.perm_z[i] = n.noise->perm_z[i];
                        p.ranvec[i][0] = n.noise->ranvec[i].x; p.ranvec[i][1] = n.noise->ranvec[i].y; p.ranvec[i][2] = n.noise->ranvec[i].z;
                    }
                    pit = perlin_ids.emplace(n.noise.get(), (uint32_t)f.perlins.size()).first;
                    f.perlins.push_back(p);
                }
                r.table = pit->second;
                break;
            }
            case Texture::IMAGE: {
                auto& im = static_cast<const ImageTexture&>(*t);
                r.kind = RT_TEX_IMAGE;
                auto iit = image_ids.find(im.img.get());
                if (iit == image_ids.end()) {
                    iit = image_ids.emplace(im.img.get(), (uint32_t)f.images.size()).first;
                    f.images.push_back(RtImage{im.img->width, im.img->height, im.img->rgb.data()});
                    f.image_owners.push_back(im.img);
                }
                r.table = iit->second;
                break;
            }
        }
        f.textures[id] = r;
        return id;
    }

    uint32_t material(const MaterialPtr& m) {
        auto it = mat_ids.find(m.get());
        if (it != mat_ids.end()) return it->second;
        RtMaterial r{};
        switch (m->kind) {
            case Material::LAMBERTIAN: r.kind = RT_MAT_LAMBERTIAN; r.texture = texture(static_cast<const Lambertian&>(*m).albedo); break;
            case Material::METAL: {
                auto& x = static_cast<const Metal&>(*m);
                r.kind = RT_MAT_METAL; r.albedo[0] = x.albedo.x; r.albedo[1] = x.albedo.y; r.albedo[2] = x.albedo.z; r.fuzz = x.fuzz;
                break;
            }
            case Material::DIELECTRIC: r.kind = RT_MAT_DIELECTRIC; r.ref_idx = static_cast<const Dielectric&>(*m).ref_idx; break;
            case Material::FROSTED_DIELECTRIC: {
                auto& x = static_cast<const FrostedDielectric&>(*m);
                r.kind = RT_MAT_FROSTED_DIELECTRIC; r.ref_idx = x.ref_idx; r.fuzz = x.fuzz;
                break;
            }
            case Material::DIFFUSE_LIGHT: r.kind = RT_MAT_DIFFUSE_LIGHT; r.texture = texture(static_cast<const DiffuseLight&>(*m).emit); break;
            case Material::ISOTROPIC: r.kind = RT_MAT_ISOTROPIC; r.texture = texture(static_cast<const Isotropic&>(*m).albedo); break;
        }
        const uint32_t id = (uint32_t)f.materials.size();
        f.materials.push_back(r);
        mat_ids[m.get()] = id;
        return id;
    }

    static bool is_primitive(const Object& o) {
        return o.kind == Object::SPHERE || o.kind == Object::MOVING_SPHERE || o.kind == Object::RECT_XY || o.kind == Object::RECT_XZ ||
               o.kind == Object::RECT_YZ || o.kind == Object::CUBOID;
    }

    uint32_t primitive(const Object& o) {
        RtPrim p{};
        switch (o.kind) {
            case Object::SPHERE: {
                auto& s = static_cast<const Sphere&>(o);
                p.kind = RT_PRIM_SPHERE; p.material = material(s.material);
                p.d[0] = s.center.x; p.d[1] = s.center.y; p.d[2] = s.center.z; p.d[3] = s.radius;
                break;
            }
            case Object::MOVING_SPHERE: {
                auto& s = static_cast<const MovingSphere&>(o);
                p.kind = RT_PRIM_MOVING_SPHERE; p.material = material(s.material);
                p.d[0] = s.center1.x; p.d[1] = s.center1.y; p.d[2] = s.center1.z;
                p.d[3] = s.center2.x; p.d[4] = s.center2.y; p.d[5] = s.center2.z;
                p.d[6] = s.t1; p.d[7] = s.t2; p.d[8] = s.radius;
                break;
            }
            case Object::RECT_XY: case Object::RECT_XZ: case Object::RECT_YZ: {
                auto& r = static_cast<const Rect&>(o);
                p.kind = o.kind == Object::RECT_XY ? RT_PRIM_RECT_XY : o.kind == Object::RECT_XZ ? RT_PRIM_RECT_XZ : RT_PRIM_RECT_YZ;
                p.material = material(r.material);
                p.d[0] = r.a1; p.d[1] = r.a2; p.d[2] = r.b1; p.d[3] = r.b2; p.d[4] = r.k; p.d[5] = r.face;
                break;
            }
            case Object::CUBOID: {
                auto& c = static_cast<const Cuboid&>(o);
                p.kind = RT_PRIM_CUBOID; p.material = material(c.material);
                p.d[0] = c.box_min.x; p.d[1] = c.box_min.y; p.d[2] = c.box_min.z;
                p.d[3] = c.box_max.x; p.d[4] = c.box_max.y; p.d[5] = c.box_max.z;
                break;
            }
            default: throw std::runtime_error("unsupported: not a primitive");
        }
        f.prims.push_back(p);
        return (uint32_t)f.prims.size() - 1;
    }

    int32_t bvh_child(const ObjectPtr& c) {
        if (c->kind == Object::BVH_NODE) return (int32_t)bvh_node(static_cast<const BvhNode&>(*c));
        if (c->kind == Object::BVH) return bvh_child(static_cast<const Bvh&>(*c).root);
        if (is_primitive(*c)) return ~(int32_t)primitive(*c);
        throw std::runtime_error("unsupported: a BVH leaf must be a primitive (Sphere/MovingSphere/Rect*/Cuboid)");
    }
    uint32_t bvh_node(const BvhNode& n) {  // pre-order: node, left subtree, right subtree
        const uint32_t id = (uint32_t)f.left.size();
        f.min_x.push_back(n.boxx.min.x); f.min_y.push_back(n.boxx.min.y); f.min_z.push_back(n.boxx.min.z);
        f.max_x.push_back(n.boxx.max.x); f.max_y.push_back(n.boxx.max.y); f.max_z.push_back(n.boxx.max.z);
        f.left.push_back(0); f.right.push_back(0);
        const int32_t l = bvh_child(n.left);
        f.left[id] = l;
        const int32_t r = bvh_child(n.right);
        f.right[id] = r;
        return id;
    }

    RtInstance instance(const ObjectPtr& entry) {
        RtInstance in{};
        const Object* cur = entry.get();
        if (cur->kind == Object::CONSTANT_MEDIUM) {
            auto& m = static_cast<const ConstantMedium&>(*cur);
            in.is_medium = 1; in.density = m.density; in.phase_material = material(m.phase_function);
            cur = m.boundary.get();
        }
        in.xform_begin = (uint32_t)f.xforms.size();
        for (;;) {
            if (cur->kind == Object::TRANSLATE) {
                auto& t = static_cast<const Translate&>(*cur);
                f.xforms.push_back(RtXform{RT_XFORM_TRANSLATE, 0, {t.offset.x, t.offset.y, t.offset.z}});
                cur = t.object.get();
            } else if (cur->kind == Object::ROTATE_Y) {
                auto& r = static_cast<const RotateY&>(*cur);
                f.xforms.push_back(RtXform{RT_XFORM_ROTATE_Y, 0, {r.sin_theta, r.cos_theta, 0.0}});
                cur = r.object.get();
            } else {
                break;
            }
        }
        in.xform_count = (uint32_t)f.xforms.size() - in.xform_begin;
        if (cur->kind == Object::BVH) cur = static_cast<const Bvh&>(*cur).root.get();
        if (cur->kind == Object::BVH_NODE) {
            in.geom_kind = RT_GEOM_BVH; in.geom_index = bvh_node(static_cast<const BvhNode&>(*cur));
        } else if (is_primitive(*cur)) {
            in.geom_kind = RT_GEOM_PRIM; in.geom_index = primitive(*cur);
        } else if (cur->kind == Object::CONSTANT_MEDIUM) {
            throw std::runtime_error("unsupported: ConstantMedium under a Translate/RotateY/ConstantMedium");
        } else {
            throw std::runtime_error("unsupported: nested ObjectList");
        }
        return in;
    }
};

void put3(double* dst, Vec3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }

}  // namespace

void flatten(const Scene& scene, FlatScene& out) {
    out = FlatScene{};
    Flattener fl{out, {}, {}, {}, {}};
    if (!scene.world || scene.world->objects.empty()) throw std::runtime_error("world is empty");
    for (auto& o : scene.world->objects) out.world.push_back(fl.instance(o));
    out.n_world_prims = (uint32_t)out.prims.size();
    if (scene.lights) {
        if (scene.lights->objects.empty()) throw std::runtime_error("unsupported: lights = Some(empty list)");
        for (auto& o : scene.lights->objects) out.lights.push_back(fl.instance(o));
    }
    out.seal();
    put3(out.desc.background, scene.background);
    RtCamera& c = out.desc.camera;
    put3(c.origin, scene.cam.origin); put3(c.lower_left_corner, scene.cam.lower_left_corner);
    put3(c.horizontal, scene.cam.horizontal); put3(c.vertical, scene.cam.vertical);
    put3(c.u, scene.cam.u); put3(c.v, scene.cam.v); put3(c.w, scene.cam.w);
    c.lens_radius = scene.cam.lens_radius; c.t1 = scene.cam.t1; c.t2 = scene.cam.t2;
}

// ---------------------------------------------------------------------------------------------
// unflatten + compare (round trip)
// ---------------------------------------------------------------------------------------------
namespace {

struct Unflattener {
    const RtSceneDesc& d;
    std::vector<TexturePtr> tex_cache;
    std::vector<MaterialPtr> mat_cache;

    explicit Unflattener(const RtSceneDesc& desc) : d(desc), tex_cache(desc.n_textures), mat_cache(desc.n_materials) {}

    TexturePtr texture(uint32_t id) {
        if (tex_cache[id]) return tex_cache[id];
        const RtTexture& t = d.textures[id];
        TexturePtr r;
        switch (t.kind) {
            case RT_TEX_SOLID: r = std::make_shared<SolidColor>(Vec3(t.color[0], t.color[1], t.color[2])); break;
            case RT_TEX_CHECKER_3D: r = std::make_shared<CheckerTexture>(texture(t.odd), texture(t.even)); break;
            case RT_TEX_CHECKER_UV: r = std::make_shared<CheckerTextureUV>(texture(t.odd), texture(t.even)); break;
            case RT_TEX_NOISE: {
                SceneRng scratch(0, 0);
                auto p = std::make_shared<Perlin>(scratch);
                const RtPerlin& src = d.perlins[t.table];
                for (int i = 0; i < 256; i++) {
                    p->perm_x[i] = src.perm_x[i]; p->perm_y[i] = src.perm_y[i]; p->perm_z[i] = src.perm_z[i];
                    p->ranvec[i] = Vec3(src.ranvec[i][0], src.ranvec[i][1], src.ranvec[i][2]);
                }
                r = std::make_shared<NoiseTexture>(p, t.scale);
                break;
            }
            default: {
                const RtImage& im = d.images[t.table];
                auto raster = std::make_shared<RgbImage>();
                raster->width = im.width; raster->height = im.height;
                raster->rgb.assign(im.rgb, im.rgb + (size_t)im.width * im.height * 3);
                r = std::make_shared<ImageTexture>(raster);
            }
        }
        return tex_cache[id] = r;
    }
    MaterialPtr material(uint32_t id) {
        if (mat_cache[id]) return mat_cache[id];
        const RtMaterial& m = d.materials[id];
        MaterialPtr r;
        switch (m.kind) {
            case RT_MAT_LAMBERTIAN: r = std::make_shared<Lambertian>(texture(m.texture)); break;
            case RT_MAT_METAL: r = std::make_shared<Metal>(Vec3(m.albedo[0], m.albedo[1], m.albedo[2]), m.fuzz); break;
            case RT_MAT_DIELECTRIC: r = std::make_shared<Dielectric>(m.ref_idx); break;
            case RT_MAT_FROSTED_DIELECTRIC: r = std::make_shared<FrostedDielectric>(m.ref_idx, m.fuzz); break;
            case RT_MAT_DIFFUSE_LIGHT: r = std::make_shared<DiffuseLight>(texture(m.texture)); break;
            default: r = std::make_shared<Isotropic>(texture(m.texture));
        }
        return mat_cache[id] = r;
    }
    ObjectPtr primitive(uint32_t id) {
        const RtPrim& p = d.prims[id];
        MaterialPtr m = material(p.material);
        switch (p.kind) {
            case RT_PRIM_SPHERE: return std::make_shared<Sphere>(Vec3(p.d[0], p.d[1], p.d[2]), p.d[3], m);
            case RT_PRIM_MOVING_SPHERE:
                return std::make_shared<MovingSphere>(Vec3(p.d[0], p.d[1], p.d[2]), Vec3(p.d[3], p.d[4], p.d[5]), p.d[6], p.d[7], p.d[8], m);
            case RT_PRIM_RECT_XY: return std::make_shared<RectXY>(p.d[0], p.d[1], p.d[2], p.d[3], p.d[4], p.d[5], m);
            case RT_PRIM_RECT_XZ: return std::make_shared<RectXZ>(p.d[0], p.d[1], p.d[2], p.d[3], p.d[4], p.d[5], m);
            case RT_PRIM_RECT_YZ: return std::make_shared<RectYZ>(p.d[0], p.d[1], p.d[2], p.d[3], p.d[4], p.d[5], m);
            default: return std::make_shared<Cuboid>(Vec3(p.d[0], p.d[1], p.d[2]), Vec3(p.d[3], p.d[4], p.d[5]), m);
        }
    }
    ObjectPtr bvh(int32_t child) {
        if (child < 0) return primitive((uint32_t)~child);
        auto n = std::make_shared<BvhNode>();
        const RtBvh& b = d.bvh;
        n->boxx = Aabb{{b.min_x[child], b.min_y[child], b.min_z[child]}, {b.max_x[child], b.max_y[child], b.max_z[child]}};
        n->left = bvh(b.left[child]);
        n->right = bvh(b.right[child]);
        return n;
    }
    ObjectPtr instance(const RtInstance& in) {
        ObjectPtr o = in.geom_kind == RT_GEOM_BVH ? bvh((int32_t)in.geom_index) : primitive(in.geom_index);
        for (int j = (int)in.xform_count - 1; j >= 0; j--) {
            const RtXform& x = d.xforms[in.xform_begin + j];
            if (x.kind == RT_XFORM_TRANSLATE) {
                o = std::make_shared<Translate>(o, Vec3(x.v[0], x.v[1], x.v[2]));
            } else {
                auto r = std::make_shared<RotateY>(o, 0.0);
                r->sin_theta = x.v[0];
                r->cos_theta = x.v[1];
                o = r;
            }
        }
        if (in.is_medium) {
            auto iso = std::static_pointer_cast<const Isotropic>(material(in.phase_material));
            o = std::make_shared<ConstantMedium>(o, iso->albedo, in.density);
        }
        return o;
    }
};

struct Comparer {
    int diffs = 0;
    std::string first;
    void note(const std::string& what) { if (diffs++ == 0) first = what; }
    void num(double a, double b, const char* what) { if (std::memcmp(&a, &b, sizeof a) != 0) note(what); }  // bit-exact
    void vec(Vec3 a, Vec3 b, const char* what) { num(a.x, b.x, what); num(a.y, b.y, what); num(a.z, b.z, what); }
    void texture(const Texture& a, const Texture& b) {
        if (a.kind != b.kind) return note("texture kind");
        switch (a.kind) {
            case Texture::SOLID: vec(static_cast<const SolidColor&>(a).color, static_cast<const SolidColor&>(b).color, "SolidColor.color"); break;
            case Texture::CHECKER: {
                auto &x = static_cast<const CheckerTexture&>(a), &y = static_cast<const CheckerTexture&>(b);
                texture(*x.odd, *y.odd); texture(*x.even, *y.even);
                break;
            }
            case Texture::CHECKER_UV: {
                auto &x = static_cast<const CheckerTextureUV&>(a), &y = static_cast<const CheckerTextureUV&>(b);
                texture(*x.odd, *y.odd); texture(*x.even, *y.even);
                break;
            }
            case Texture::NOISE: {
                auto &x = static_cast<const NoiseTexture&>(a), &y = static_cast<const NoiseTexture&>(b);
                num(x.scale, y.scale, "NoiseTexture.scale");
                if (x.noise->perm_x != y.noise->perm_x || x.noise->perm_y != y.noise->perm_y || x.noise->perm_z != y.noise->perm_z) note("Perlin.perm");
                for (int i = 0; i < 256; i++) vec(x.noise->ranvec[i], y.noise->ranvec[i], "Perlin.ranvec");
                break;
            }
            case Texture::IMAGE: {
                auto &x = static_cast<const ImageTexture&>(a), &y = static_cast<const ImageTexture&>(b);
                if (x.img->width != y.img->width || x.img->height != y.img->height || x.img->rgb != y.img->rgb) note("ImageTexture.img");
                break;
            }
        }
    }
    void material(const Material& a, const Material& b) {
        if (a.kind != b.kind) return note("material kind");
        switch (a.kind) {
            case Material::LAMBERTIAN: texture(*static_cast<const Lambertian&>(a).albedo, *static_cast<const Lambertian&>(b).albedo); break;
            case Material::METAL:
                vec(static_cast<const Metal&>(a).albedo, static_cast<const Metal&>(b).albedo, "Metal.albedo");
                num(static_cast<const Metal&>(a).fuzz, static_cast<const Metal&>(b).fuzz, "Metal.fuzz");
                break;
            case Material::DIELECTRIC: num(static_cast<const Dielectric&>(a).ref_idx, static_cast<const Dielectric&>(b).ref_idx, "Dielectric.ref_idx"); break;
            case Material::FROSTED_DIELECTRIC:
                num(static_cast<const FrostedDielectric&>(a).ref_idx, static_cast<const FrostedDielectric&>(b).ref_idx, "FrostedDielectric.ref_idx");
                num(static_cast<const FrostedDielectric&>(a).fuzz, static_cast<const FrostedDielectric&>(b).fuzz, "FrostedDielectric.fuzz");
                break;
            case Material::DIFFUSE_LIGHT: texture(*static_cast<const DiffuseLight&>(a).emit, *static_cast<const DiffuseLight&>(b).emit); break;
            case Material::ISOTROPIC: texture(*static_cast<const Isotropic&>(a).albedo, *static_cast<const Isotropic&>(b).albedo); break;
        }
    }
    static const Object& skip_bvh_wrapper(const Object& o) { return o.kind == Object::BVH ? *static_cast<const Bvh&>(o).root : o; }
    void object(const Object& a0, const Object& b0) {
        const Object &a = skip_bvh_wrapper(a0), &b = skip_bvh_wrapper(b0);  // Bvh is a transparent wrapper (bvh.rs:71-78)
        if (a.kind != b.kind) return note("object kind");
        switch (a.kind) {
            case Object::LIST: {
                auto &x = static_cast<const ObjectList&>(a), &y = static_cast<const ObjectList&>(b);
                if (x.objects.size() != y.objects.size()) return note("ObjectList length");
                for (size_t i = 0; i < x.objects.size(); i++) object(*x.objects[i], *y.objects[i]);
                break;
            }
            case Object::SPHERE: {
                auto &x = static_cast<const Sphere&>(a), &y = static_cast<const Sphere&>(b);
                vec(x.center, y.center, "Sphere.center"); num(x.radius, y.radius, "Sphere.radius"); material(*x.material, *y.material);
                break;
            }
            case Object::MOVING_SPHERE: {
                auto &x = static_cast<const MovingSphere&>(a), &y = static_cast<const MovingSphere&>(b);
                vec(x.center1, y.center1, "MovingSphere.center1"); vec(x.center2, y.center2, "MovingSphere.center2");
                num(x.t1, y.t1, "MovingSphere.t1"); num(x.t2, y.t2, "MovingSphere.t2"); num(x.radius, y.radius, "MovingSphere.radius");
                material(*x.material, *y.material);
                break;
            }
            case Object::RECT_XY: case Object::RECT_XZ: case Object::RECT_YZ: {
                auto &x = static_cast<const Rect&>(a), &y = static_cast<const Rect&>(b);
                num(x.a1, y.a1, "Rect.a1"); num(x.a2, y.a2, "Rect.a2"); num(x.b1, y.b1, "Rect.b1"); num(x.b2, y.b2, "Rect.b2");
                num(x.k, y.k, "Rect.k"); num(x.face, y.face, "Rect.face"); material(*x.material, *y.material);
                break;
            }
            case Object::CUBOID: {
                auto &x = static_cast<const Cuboid&>(a), &y = static_cast<const Cuboid&>(b);
                vec(x.box_min, y.box_min, "Cuboid.box_min"); vec(x.box_max, y.box_max, "Cuboid.box_max"); material(*x.material, *y.material);
                break;
            }
            case Object::CONSTANT_MEDIUM: {
                auto &x = static_cast<const ConstantMedium&>(a), &y = static_cast<const ConstantMedium&>(b);
                num(x.density, y.density, "ConstantMedium.density"); material(*x.phase_function, *y.phase_function);
                object(*x.boundary, *y.boundary);
                break;
            }
            case Object::TRANSLATE: {
                auto &x = static_cast<const Translate&>(a), &y = static_cast<const Translate&>(b);
                vec(x.offset, y.offset, "Translate.offset"); object(*x.object, *y.object);
                break;
            }
            case Object::ROTATE_Y: {
                auto &x = static_cast<const RotateY&>(a), &y = static_cast<const RotateY&>(b);
                num(x.sin_theta, y.sin_theta, "RotateY.sin_theta"); num(x.cos_theta, y.cos_theta, "RotateY.cos_theta");
                object(*x.object, *y.object);
                break;
            }
            case Object::BVH_NODE: {
                auto &x = static_cast<const BvhNode&>(a), &y = static_cast<const BvhNode&>(b);
                vec(x.boxx.min, y.boxx.min, "BvhNode.boxx.min"); vec(x.boxx.max, y.boxx.max, "BvhNode.boxx.max");
                object(*x.left, *y.left); object(*x.right, *y.right);
                break;
            }
            case Object::BVH: break;  // unreachable (skipped above)
        }
    }
};

}  // namespace

Scene unflatten(const RtSceneDesc& d) {
    Unflattener u(d);
    Scene s;
    s.world = std::make_shared<ObjectList>();
    for (uint32_t i = 0; i < d.n_world; i++) s.world->add(u.instance(d.world[i]));
    if (d.n_lights) {
        s.lights = std::make_shared<ObjectList>();
        for (uint32_t i = 0; i < d.n_lights; i++) s.lights->add(u.instance(d.lights[i]));
    }
    s.background = Vec3(d.background[0], d.background[1], d.background[2]);
    const RtCamera& c = d.camera;
    auto v3 = [](const double* p) { return Vec3(p[0], p[1], p[2]); };
    s.cam.origin = v3(c.origin); s.cam.lower_left_corner = v3(c.lower_left_corner);
    s.cam.horizontal = v3(c.horizontal); s.cam.vertical = v3(c.vertical);
    s.cam.u = v3(c.u); s.cam.v = v3(c.v); s.cam.w = v3(c.w);
    s.cam.lens_radius = c.lens_radius; s.cam.t1 = c.t1; s.cam.t2 = c.t2;
    return s;
}

int compare_scenes(const Scene& a, const Scene& b, std::string& first) {
    Comparer c;
    c.object(*a.world, *b.world);
    if ((a.lights != nullptr) != (b.lights != nullptr)) c.note("lights Some/None");
    else if (a.lights) c.object(*a.lights, *b.lights);
    c.vec(a.background, b.background, "background");
    c.vec(a.cam.origin, b.cam.origin, "Camera.origin"); c.vec(a.cam.lower_left_corner, b.cam.lower_left_corner, "Camera.lower_left_corner");
    c.vec(a.cam.horizontal, b.cam.horizontal, "Camera.horizontal"); c.vec(a.cam.vertical, b.cam.vertical, "Camera.vertical");
    c.vec(a.cam.u, b.cam.u, "Camera.u"); c.vec(a.cam.v, b.cam.v, "Camera.v"); c.vec(a.cam.w, b.cam.w, "Camera.w");
    c.num(a.cam.lens_radius, b.cam.lens_radius, "Camera.lens_radius"); c.num(a.cam.t1, b.cam.t1, "Camera.t1"); c.num(a.cam.t2, b.cam.t2, "Camera.t2");
    first = c.first;
    return c.diffs;
}

}  // namespace rth
