"""Random flat scenes for the parity tests (shared by the CPU host-simulation test and the GPU test)."""
import math
import random

from ray_tracing_amd import flat as F


def random_scene(seed, with_bvh=True, with_media=True, with_lights=True):
    rng = random.Random(seed)
    fs = F.FlatScene()
    fs.background = [rng.choice([0.0, 0.7]), rng.choice([0.0, 0.8]), rng.choice([0.05, 1.0])]
    U = rng.uniform

    def tex():
        c = rng.random()
        if c < 0.6:
            return fs.solid(U(0.1, 0.9), U(0.1, 0.9), U(0.1, 0.9))
        if c < 0.8:
            return fs.texture(F.CHECKER_3D, odd=fs.solid(U(0, 0.3), U(0, 0.3), U(0, 0.3)), even=fs.solid(0.9, 0.9, 0.9))
        return fs.texture(F.CHECKER_UV, odd=fs.solid(0.2, 0.3, 0.1), even=fs.texture(F.CHECKER_3D, odd=fs.solid(0.9, 0.1, 0.1), even=fs.solid(0.8, 0.8, 0.9)))

    def mat(allow_light=True):
        c = rng.random()
        if c < 0.45:
            return fs.material(F.LAMBERTIAN, texture=tex())
        if c < 0.6:
            return fs.material(F.METAL, albedo=(U(0.5, 1), U(0.5, 1), U(0.5, 1)), fuzz=U(0, 0.6))
        if c < 0.75:
            return fs.material(F.DIELECTRIC, ref_idx=U(1.2, 1.8))
        if c < 0.82:
            return fs.material(F.FROSTED, ref_idx=1.5, fuzz=U(0.05, 0.4))
        if allow_light:
            return fs.material(F.DIFFUSE_LIGHT, texture=fs.solid(U(2, 8), U(2, 8), U(2, 8)))
        return fs.material(F.LAMBERTIAN, texture=tex())

    def prim(scale=1.0):
        c = rng.random()
        p = [U(-6, 6) * scale, U(-2, 4) * scale, U(-6, 6) * scale]
        if c < 0.4:
            return fs.sphere(p, U(0.3, 1.5) * scale, mat())
        if c < 0.55:
            p2 = [p[0] + U(-0.5, 0.5), p[1] + U(0, 1), p[2] + U(-0.5, 0.5)]
            return fs.moving_sphere(p, p2, 0.0, 1.0, U(0.3, 1.0) * scale, mat())
        if c < 0.8:
            a1, b1 = U(-6, 4) * scale, U(-6, 4) * scale
            return fs.rect(rng.randrange(3), a1, a1 + U(0.5, 4) * scale, b1, b1 + U(0.5, 4) * scale, U(-5, 5) * scale, rng.choice([-1.0, 1.0]), mat())
        mx = [p[k] + U(0.4, 2.5) * scale for k in range(3)]
        return fs.cuboid(p, mx, mat())

    def chain():
        c = rng.random()
        if c < 0.5:
            return ()
        if c < 0.65:
            return (("t", (U(-2, 2), U(-1, 1), U(-2, 2))),)
        if c < 0.8:
            return (("r", U(-60, 60)),)
        if c < 0.95:
            return (("t", (U(-2, 2), U(-1, 1), U(-2, 2))), ("r", U(-60, 60)))
        return (("r", U(-30, 30)), ("t", (U(-1, 1), 0.0, U(-1, 1))), ("r", U(-30, 30)))  # a chain outside the reference's usual forms

    # a big diffuse floor so paths bounce
    fs.add_world(fs.rect(1, -30, 30, -30, 30, -2.5, 1.0, fs.material(F.LAMBERTIAN, texture=tex())))
    for _ in range(rng.randrange(3, 8)):
        fs.add_world(prim(), chain=chain())
    if with_bvh:
        for _ in range(rng.randrange(1, 3)):
            ids = [prim(scale=0.6) for _ in range(rng.randrange(2, 24))]
            fs.add_world(fs.build_bvh(ids, rng), bvh=True, chain=chain() if rng.random() < 0.5 else ())
    if with_media and rng.random() < 0.7:
        iso = fs.material(F.ISOTROPIC, texture=fs.solid(U(0.2, 1), U(0.2, 1), U(0.2, 1)))
        b = fs.sphere([U(-3, 3), U(0, 2), U(-3, 3)], U(1, 2.5), fs.material(F.DIELECTRIC, ref_idx=1.5)) if rng.random() < 0.5 else \
            fs.cuboid([-1, -1, -1], [U(0.5, 2), U(0.5, 2), U(0.5, 2)], fs.material(F.LAMBERTIAN, texture=fs.solid(1, 1, 1)))
        fs.add_world(b, chain=chain(), medium=(U(0.05, 0.8), iso))
    if with_lights and rng.random() < 0.7:
        lm = fs.material(F.DIFFUSE_LIGHT, texture=fs.solid(6, 6, 6))
        lr = fs.rect(1, -1.5, 1.5, -1.0, 1.0, 6.0, -1.0, lm)
        fs.add_world(lr)
        fs.add_light(lr)
        if rng.random() < 0.6:
            gs = fs.sphere([U(-2, 2), U(0, 1), U(-2, 2)], U(0.5, 1.0), fs.material(F.DIELECTRIC, ref_idx=1.5))
            ch = chain()
            fs.add_world(gs, chain=ch)
            fs.add_light(gs, chain=ch)
        if rng.random() < 0.3:
            cb = fs.cuboid([3, -2, 3], [4, 0, 4.5], lm)
            fs.add_world(cb)
            fs.add_light(cb)
    fs.camera([U(8, 14), U(1, 5), U(8, 14)], [0, 0.5, 0], [0, 1, 0], U(25, 50), 1.0, rng.choice([0.0, 0.2]), 12.0, 0.0, 1.0)
    return fs
