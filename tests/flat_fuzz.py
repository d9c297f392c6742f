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


def tie_scene(seed):
    """Scenes built to hit what the device's own tree order could get wrong (rt_device.hpp: walk_fast): EQUAL hit distances
    inside one BVH (the same sphere / cuboid / rectangle twice with different materials: spheres keep the first in the
    reference's order, rectangles and cuboids take the last), a grid of cuboids with shared faces as in final_scene's ground,
    spheres tangent to their neighbours' boxes, integer coordinates and an axis-aligned camera on a grid line, so that
    many rays run exactly in a coordinate plane (a zero direction component: 0/0 and x/0 in the slab tests) or hit
    exactly on box faces and edges."""
    rng = random.Random(1000 + seed)
    fs = F.FlatScene()
    fs.background = [0.7, 0.8, 1.0]

    def m(i):
        return fs.material(F.LAMBERTIAN, texture=fs.solid(0.1 + 0.07 * (i % 12), 0.9 - 0.05 * (i % 15), 0.3 + 0.04 * (i % 9)))

    ids = []
    k = 0
    # ground grid of cuboids with shared side faces and a few equal heights (shared top edges)
    n = rng.randrange(3, 6)
    for ix in range(n):
        for iz in range(n):
            h = float(rng.choice([1, 1, 2, 3]))
            ids.append(fs.cuboid([2.0 * ix - n, -1.0, 2.0 * iz - n], [2.0 * ix - n + 2.0, h, 2.0 * iz - n + 2.0], m(k))); k += 1
    # the same primitives twice (other material), appended in random places of the build order
    dup = []
    for _ in range(rng.randrange(2, 5)):
        c = [float(rng.randrange(-n, n)), float(rng.randrange(4, 7)), float(rng.randrange(-n, n))]
        r = float(rng.choice([1, 1, 2]))
        dup += [fs.sphere(c, r, m(k)), fs.sphere(c, r, m(k + 1))]; k += 2
    for _ in range(rng.randrange(1, 3)):
        lo = [float(rng.randrange(-n, n)), 4.0, float(rng.randrange(-n, n))]
        hi = [lo[0] + 1.0, 5.0, lo[2] + 2.0]
        dup += [fs.cuboid(lo, hi, m(k)), fs.cuboid(lo, hi, m(k + 1))]; k += 2
    a1 = float(rng.randrange(-n, 0))
    dup += [fs.rect(1, a1, a1 + 3.0, -2.0, 2.0, 7.0, -1.0, m(k)), fs.rect(1, a1, a1 + 3.0, -2.0, 2.0, 7.0, 1.0, m(k + 1))]; k += 2
    # spheres resting on the grid (tangent to the cuboid tops) and touching each other
    for _ in range(rng.randrange(2, 5)):
        ix, iz = rng.randrange(n), rng.randrange(n)
        dup.append(fs.sphere([2.0 * ix - n + 1.0, 4.0, 2.0 * iz - n + 1.0], 1.0, m(k))); k += 1
    ids += dup
    rng.shuffle(ids)
    chain = () if seed % 3 else (("t", (1.0, 0.0, -2.0)),)
    fs.add_world(fs.build_bvh(ids, rng), bvh=True, chain=chain)
    # something in front of and something behind the tree in the world list, one of them at a distance equal to a tree hit
    fs.add_world(fs.rect(1, -30, 30, -30, 30, -1.0, 1.0, m(k)))  # the plane of the cuboids' bottoms
    if seed % 2:
        fs.add_world(fs.sphere([0.0, 4.0, 0.0], 1.0, m(k + 1)))
    # on a grid line, looking along -z: the centre column of an odd-width probe has d.x == 0 exactly
    fs.camera([0.0, 5.0, 3.0 * n + 6.0], [0.0, 2.0, 0.0] if seed % 2 else [0.0, 5.0, 0.0], [0, 1, 0], 40.0, 1.0, 0.0, 10.0, 0.0, 1.0)
    return fs


def cross_tie_scene(seed):
    """Equal hit distances ACROSS world-list entries (rt_device.hpp: world_hit scans the leading run of the list with the plain
    primitives first and the trees last, and must notice when that order could decide): the leaves of a tree again as plain
    world objects — the same sphere (strict comparison: the EARLIER list entry keeps the hit), the same cuboid, a rectangle in
    a cuboid's face (inclusive: the LATER entry takes it) — placed BEFORE and AFTER the tree in the list, with other materials
    so that the winner shows in the id and in the radiance.  Odd seeds put a second tree and a medium behind the first run."""
    rng = random.Random(7000 + seed)
    fs = F.FlatScene()
    fs.background = [0.6, 0.7, 0.9]

    def m(i):
        return fs.material(F.LAMBERTIAN, texture=fs.solid(0.15 + 0.06 * (i % 13), 0.85 - 0.05 * (i % 11), 0.25 + 0.05 * (i % 7)))

    k = 0
    spheres, boxes = [], []
    for _ in range(rng.randrange(4, 8)):
        c = [float(rng.randrange(-6, 7)), float(rng.randrange(1, 6)), float(rng.randrange(-6, 7))]
        spheres.append((c, float(rng.choice([1, 1, 2]))))
    for _ in range(rng.randrange(3, 6)):
        lo = [float(rng.randrange(-7, 6)), 0.0, float(rng.randrange(-7, 6))]
        boxes.append((lo, [lo[0] + float(rng.choice([1, 2])), float(rng.randrange(1, 4)), lo[2] + float(rng.choice([1, 2]))]))
    leaves = []
    for c, r in spheres:
        leaves.append(fs.sphere(c, r, m(k))); k += 1
    for lo, hi in boxes:
        leaves.append(fs.cuboid(lo, hi, m(k))); k += 1
    rng.shuffle(leaves)
    before, after = [], []
    for c, r in spheres:
        if rng.random() < 0.7:
            (before if rng.random() < 0.5 else after).append(fs.sphere(c, r, m(k))); k += 1
    for lo, hi in boxes:
        roll = rng.random()
        dst = before if rng.random() < 0.5 else after
        if roll < 0.4:
            dst.append(fs.cuboid(lo, hi, m(k))); k += 1
        elif roll < 0.8:  # its top face as a rectangle of its own (XZ at y = hi.y)
            dst.append(fs.rect(1, lo[0], hi[0], lo[2], hi[2], hi[1], 1.0, m(k))); k += 1
    for p in before:
        fs.add_world(p)
    fs.add_world(fs.build_bvh(leaves, rng), bvh=True)
    for p in after:
        fs.add_world(p)
    fs.add_world(fs.rect(1, -30, 30, -30, 30, 0.0, 1.0, m(k)))  # the plane of the cuboids' bottoms (equal to their -y faces)
    k += 1
    if seed % 2:
        fog = fs.material(F.ISOTROPIC, texture=fs.solid(0.8, 0.8, 0.8))
        fs.add_world(fs.sphere([0.0, 3.0, 0.0], 9.0, m(k)), medium=(0.05, fog))
        more = [fs.sphere([float(rng.randrange(-6, 7)), float(rng.randrange(6, 9)), float(rng.randrange(-6, 7))], 1.0, m(k + 1 + i)) for i in range(4)]
        fs.add_world(fs.build_bvh(more, rng), bvh=True)
        fs.add_world(fs.sphere(spheres[0][0], spheres[0][1], m(k + 6)))  # the first tree's sphere once more, behind the medium
    fs.camera([0.0, 4.0, 24.0], [0.0, 2.0, 0.0], [0, 1, 0], 40.0, 1.0, 0.0, 10.0, 0.0, 1.0)
    return fs


def nan_scene():
    """A scene whose diffuse bounces all come out NaN: the light LIST holds a sphere that encloses the whole scene, and
    Sphere::pdf_value from inside a sphere takes the square root of a negative number (src/objects.rs:126-137).  Samples that
    see the floor are retried until the cap and dropped; samples that see the lamp or the sky succeed.  No BVH: the
    block-sorted kernel renders it."""
    fs = F.FlatScene()
    grey = fs.material(F.LAMBERTIAN, fs.solid(0.7, 0.7, 0.7))
    lamp = fs.material(F.DIFFUSE_LIGHT, fs.solid(4.0, 3.0, 2.0))
    fs.add_world(fs.rect(1, -3.0, 3.0, -3.0, 3.0, 0.0, 1.0, grey))       # floor (XZ at y = 0)
    fs.add_world(fs.sphere((0.0, 2.5, 0.0), 0.6, lamp))
    fs.add_light(fs.sphere((0.0, 1.0, 0.0), 50.0, lamp))                # encloses camera and floor
    fs.background = [0.1, 0.2, 0.3]
    fs.camera((0.0, 2.0, -6.0), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0), 50.0, 1.0, 0.0, 6.0)
    return fs


def medium_scene(seed):
    """ConstantMedium with every kind of boundary the flat form can express: a sphere, a MOVING sphere, a cuboid, a single
    rectangle (never scatters: the second boundary search finds nothing), and a BVH of spheres and cuboids — some behind
    wrapper chains — over a lit floor.  The device takes two routes through a medium (rt_device.hpp: boundary_two for one
    sphere / cuboid, two searches through geom_hit for everything else); this scene set walks both."""
    rng = random.Random(5000 + seed)
    U = rng.uniform
    fs = F.FlatScene()
    grey = fs.material(F.LAMBERTIAN, fs.solid(0.6, 0.6, 0.6))
    lamp = fs.material(F.DIFFUSE_LIGHT, fs.solid(5.0, 5.0, 5.0))
    glass = fs.material(F.DIELECTRIC, ref_idx=1.5)
    fs.add_world(fs.rect(1, -20, 20, -20, 20, -1.0, 1.0, grey))
    lr = fs.rect(1, -2.0, 2.0, -2.0, 2.0, 7.0, -1.0, lamp)
    fs.add_world(lr)
    if seed % 2 == 0:
        fs.add_light(lr)
    fs.add_world(fs.sphere([U(-4, 4), 0.5, U(-4, 4)], 1.0, fs.material(F.METAL, albedo=(0.8, 0.7, 0.6), fuzz=0.1)))

    def chain():
        c = rng.random()
        if c < 0.4:
            return ()
        if c < 0.7:
            return (("t", (U(-2, 2), U(0, 1), U(-2, 2))),)
        return (("t", (U(-2, 2), U(0, 1), U(-2, 2))), ("r", U(-45, 45)))

    kinds = ["sphere", "moving", "cuboid", "rect", "bvh"]
    rng.shuffle(kinds)
    for kind in kinds[:3 + seed % 3]:
        iso = fs.material(F.ISOTROPIC, texture=fs.solid(U(0.2, 1), U(0.2, 1), U(0.2, 1)))
        c = [U(-4, 4), U(0.5, 2.5), U(-4, 4)]
        if kind == "sphere":
            fs.add_world(fs.sphere(c, U(0.8, 2.0), glass), chain=chain(), medium=(U(0.1, 1.5), iso))
        elif kind == "moving":
            c2 = [c[0] + U(-1, 1), c[1] + U(-0.3, 0.3), c[2] + U(-1, 1)]
            fs.add_world(fs.moving_sphere(c, c2, 0.0, 1.0, U(0.8, 1.8), glass), chain=chain(), medium=(U(0.1, 1.5), iso))
        elif kind == "cuboid":
            fs.add_world(fs.cuboid([c[0] - 1, c[1] - 1, c[2] - 1], [c[0] + U(0.5, 1.5), c[1] + U(0.5, 1.5), c[2] + U(0.5, 1.5)], grey),
                         chain=chain(), medium=(U(0.1, 1.5), iso))
        elif kind == "rect":
            fs.add_world(fs.rect(rng.randrange(3), -1.5, 1.5, -1.5, 1.5, U(0, 2), 1.0, grey), chain=chain(), medium=(U(0.1, 1.5), iso))
        else:
            ids = []
            for _ in range(rng.randrange(2, 7)):
                q = [c[0] + U(-1.5, 1.5), c[1] + U(-0.5, 1.0), c[2] + U(-1.5, 1.5)]
                ids.append(fs.sphere(q, U(0.4, 1.0), glass) if rng.random() < 0.6 else
                           fs.cuboid([q[0] - 0.5, q[1] - 0.5, q[2] - 0.5], [q[0] + 0.5, q[1] + 0.6, q[2] + 0.7], grey))
            fs.add_world(fs.build_bvh(ids, rng), bvh=True, chain=chain() if rng.random() < 0.5 else (), medium=(U(0.1, 1.5), iso))
    fs.background = [0.3, 0.4, 0.6]
    fs.camera([U(9, 13), U(2, 5), U(9, 13)], [0, 1.0, 0], [0, 1, 0], U(30, 45), 1.0, rng.choice([0.0, 0.15]), 13.0, 0.0, 1.0)
    return fs


def odd_lights_scene(seed):
    """A light LIST with members that keep the Object trait's defaults (src/objects.rs:22-27: pdf_value = 0, random = (1,0,0)):
    a BvhNode, a ConstantMedium and a MovingSphere, beside an ordinary rectangle and a translated sphere."""
    rng = random.Random(7000 + seed)
    U = rng.uniform
    fs = F.FlatScene()
    grey = fs.material(F.LAMBERTIAN, fs.solid(0.7, 0.6, 0.5))
    lamp = fs.material(F.DIFFUSE_LIGHT, fs.solid(6.0, 5.0, 4.0))
    glass = fs.material(F.DIELECTRIC, ref_idx=1.5)
    iso = fs.material(F.ISOTROPIC, texture=fs.solid(0.9, 0.9, 0.9))
    fs.add_world(fs.rect(1, -15, 15, -15, 15, 0.0, 1.0, grey))
    lr = fs.rect(1, -1.5, 1.5, -1.5, 1.5, 6.0, -1.0, lamp)
    fs.add_world(lr); fs.add_light(lr)
    gs = fs.sphere([0.0, 1.0, 0.0], 1.0, glass)
    ch = (("t", (U(-3, 3), 0.0, U(-3, 3))),)
    fs.add_world(gs, chain=ch); fs.add_light(gs, chain=ch)
    ids = [fs.sphere([U(-5, 5), U(0.5, 1.5), U(-5, 5)], U(0.4, 0.9), grey) for _ in range(rng.randrange(3, 9))]
    root = fs.build_bvh(ids, rng)
    fs.add_world(root, bvh=True); fs.add_light(root, bvh=True)
    fog = fs.sphere([U(-3, 3), 1.5, U(-3, 3)], 1.5, glass)
    fs.add_world(fog, medium=(0.4, iso)); fs.add_light(fog, medium=(0.4, iso))
    ms = fs.moving_sphere([3.0, 1.0, -3.0], [3.5, 1.2, -3.0], 0.0, 1.0, 0.8, lamp)
    fs.add_world(ms); fs.add_light(ms)
    fs.background = [0.05, 0.05, 0.1]
    fs.camera([U(8, 12), U(3, 6), U(8, 12)], [0, 1.0, 0], [0, 1, 0], 40.0, 1.0, 0.0, 12.0, 0.0, 1.0)
    return fs


def far_scene(seed, offset=(1.0e6, -3.0e5, 2.0e6)):
    """Everything far from the origin (coordinates ~1e6 with features of size ~1): the f32 copies of the boxes resolve 0.06 - 0.25
    there and the f32 node classification's margins are units wide, so almost every decision goes to the second (f64) level, the
    exact test or the reference-order walk.  Spheres (some moving), cuboids and rectangles in a BVH, a floor, a light."""
    rng = random.Random(11000 + seed)
    U = rng.uniform
    ox, oy, oz = offset
    fs = F.FlatScene()
    grey = fs.material(F.LAMBERTIAN, fs.solid(0.6, 0.6, 0.6))
    lamp = fs.material(F.DIFFUSE_LIGHT, fs.solid(5.0, 5.0, 5.0))
    glass = fs.material(F.DIELECTRIC, ref_idx=1.5)
    metal = fs.material(F.METAL, albedo=(0.8, 0.8, 0.7), fuzz=0.05)
    fs.add_world(fs.rect(1, ox - 30, ox + 30, oz - 30, oz + 30, oy - 1.0, 1.0, grey))
    lr = fs.rect(1, ox - 2, ox + 2, oz - 2, oz + 2, oy + 8.0, -1.0, lamp)
    fs.add_world(lr)
    if seed % 2:
        fs.add_light(lr)
    ids = []
    for _ in range(rng.randrange(6, 40)):
        c = [ox + U(-6, 6), oy + U(0, 4), oz + U(-6, 6)]
        k = rng.random()
        if k < 0.5:
            ids.append(fs.sphere(c, U(0.3, 1.2), rng.choice([grey, glass, metal])))
        elif k < 0.7:
            ids.append(fs.moving_sphere(c, [c[0] + U(-1, 1), c[1] + U(0, 0.5), c[2] + U(-1, 1)], 0.0, 1.0, U(0.3, 0.9), grey))
        elif k < 0.9:
            ids.append(fs.cuboid([c[0] - 0.5, c[1] - 0.5, c[2] - 0.5], [c[0] + U(0.3, 1), c[1] + U(0.3, 1), c[2] + U(0.3, 1)], grey))
        else:
            ids.append(fs.rect(rng.randrange(3), c[0] - 1, c[0] + 1, c[1] - 1, c[1] + 1, c[2], 1.0, grey))
    fs.add_world(fs.build_bvh(ids, rng), bvh=True)
    fs.add_world(fs.sphere([ox + U(-3, 3), oy + 1.0, oz + U(-3, 3)], 1.0, glass))
    fs.background = [0.4, 0.5, 0.7]
    fs.camera([ox + U(9, 13), oy + U(2, 6), oz + U(9, 13)], [ox, oy + 1.0, oz], [0, 1, 0], U(30, 45), 1.0, rng.choice([0.0, 0.1]), 14.0, 0.0, 1.0)
    return fs


def pile_scene(seed):
    """Many primitives in the SAME place inside one BVH: copies of one sphere, of one cuboid and of one rectangle (different
    materials), plus concentric spheres.  Every centroid coincides (the device's tree builder cannot split by position) and nearly
    every hit is an exact tie, which the reference resolves by its visiting order."""
    rng = random.Random(13000 + seed)
    fs = F.FlatScene()
    mats = [fs.material(F.LAMBERTIAN, fs.solid(rng.random(), rng.random(), rng.random())) for _ in range(6)]
    mats.append(fs.material(F.METAL, albedo=(0.9, 0.8, 0.7), fuzz=0.0))
    mats.append(fs.material(F.DIELECTRIC, ref_idx=1.5))
    fs.add_world(fs.rect(1, -20, 20, -20, 20, -1.0, 1.0, mats[0]))
    ids = []
    n = 20 + 15 * (seed % 4)
    for i in range(n):
        ids.append(fs.sphere([0.0, 1.0, 0.0], 1.0, rng.choice(mats)))
    for i in range(n // 2):
        ids.append(fs.cuboid([2.0, -1.0, -1.0], [3.5, 1.0, 1.0], rng.choice(mats)))
    for i in range(n // 3):
        ids.append(fs.rect(0, -4.0, -2.0, -1.0, 2.0, 0.5, 1.0, rng.choice(mats)))
    for i in range(8):
        ids.append(fs.sphere([0.0, 1.0, 3.0], 0.2 + 0.1 * i, rng.choice(mats)))
    rng.shuffle(ids)
    fs.add_world(fs.build_bvh(ids, rng), bvh=True)
    fs.background = [0.6, 0.7, 0.9]
    fs.camera([7.0, 3.0, 9.0], [0.5, 0.5, 0.5], [0, 1, 0], 40.0, 1.0, 0.0, 10.0, 0.0, 1.0)
    return fs


def texture_scene(seed):
    """NoiseTexture and ImageTexture in every position the flat form allows: Perlin tables with scales from tiny to huge (and
    negative), images of awkward sizes (1x1, 3x2, 17x5, 64x33) on spheres, moving spheres, rectangles and cuboids, as emitters, as
    the phase texture of a medium, and as the branches of nested checker textures; several tables and images per scene."""
    rng = random.Random(17000 + seed)
    U = rng.uniform
    fs = F.FlatScene()
    for _ in range(1 + seed % 3):  # Perlin tables: permutations + unit vectors (texture.rs:100-130)
        perms = []
        for _k in range(3):
            p = list(range(256)); rng.shuffle(p); perms.append(p)
        vecs = []
        for _k in range(256):
            v = [U(-1, 1), U(-1, 1), U(-1, 1)]
            n = math.sqrt(sum(c * c for c in v)) or 1.0
            vecs.append([c / n for c in v])
        fs.perlins.append((perms[0], perms[1], perms[2], vecs))
    for (w, h) in ((1, 1), (3, 2), (17, 5), (64, 33))[: 2 + seed % 3]:
        fs.images.append((w, h, bytes(rng.randrange(256) for _ in range(w * h * 3))))

    def noise():
        return fs.texture(F.NOISE, table=rng.randrange(len(fs.perlins)), scale=rng.choice([0.1, 1.0, 4.0, 37.5, -2.0, 1e-3, 250.0]))

    def image():
        return fs.texture(F.IMAGE, table=rng.randrange(len(fs.images)))

    def tex(depth=0):
        c = rng.random()
        if c < 0.35:
            return noise()
        if c < 0.7:
            return image()
        if c < 0.8 or depth >= 2:
            return fs.solid(U(0.1, 0.9), U(0.1, 0.9), U(0.1, 0.9))
        return fs.texture(rng.choice([F.CHECKER_3D, F.CHECKER_UV]), odd=tex(depth + 1), even=tex(depth + 1))

    def mat():
        c = rng.random()
        if c < 0.7:
            return fs.material(F.LAMBERTIAN, texture=tex())
        if c < 0.85:
            return fs.material(F.DIFFUSE_LIGHT, texture=tex())
        return fs.material(F.METAL, albedo=(0.8, 0.8, 0.8), fuzz=0.1)

    fs.add_world(fs.rect(1, -20, 20, -20, 20, -1.0, 1.0, fs.material(F.LAMBERTIAN, texture=tex())))
    for _ in range(rng.randrange(5, 10)):
        p = [U(-5, 5), U(0, 3), U(-5, 5)]
        k = rng.random()
        if k < 0.35:
            prim = fs.sphere(p, U(0.5, 1.5), mat())
        elif k < 0.5:
            prim = fs.moving_sphere(p, [p[0] + U(-0.5, 0.5), p[1] + U(0, 0.6), p[2]], 0.0, 1.0, U(0.4, 1.0), mat())
        elif k < 0.75:
            prim = fs.rect(rng.randrange(3), p[0] - 1.5, p[0] + 1.5, p[1] - 1.0, p[1] + 2.0, p[2], rng.choice([-1.0, 1.0]), mat())
        else:
            prim = fs.cuboid(p, [p[0] + U(0.5, 2), p[1] + U(0.5, 2), p[2] + U(0.5, 2)], mat())
        c = rng.random()
        chain = () if c < 0.5 else ((("t", (U(-1, 1), 0.0, U(-1, 1))),) if c < 0.75 else (("t", (U(-1, 1), 0.0, U(-1, 1))), ("r", U(-40, 40))))
        fs.add_world(prim, chain=chain)
    ids = [fs.sphere([U(-5, 5), U(3, 5), U(-5, 5)], U(0.3, 0.8), mat()) for _ in range(rng.randrange(2, 10))]
    fs.add_world(fs.build_bvh(ids, rng), bvh=True)
    iso = fs.material(F.ISOTROPIC, texture=tex())
    fs.add_world(fs.sphere([U(-3, 3), 1.5, U(-3, 3)], 1.8, fs.material(F.DIELECTRIC, ref_idx=1.5)), medium=(U(0.2, 1.0), iso))
    lr = fs.rect(1, -2, 2, -2, 2, 8.0, -1.0, fs.material(F.DIFFUSE_LIGHT, texture=fs.solid(5, 5, 5)))
    fs.add_world(lr)
    if seed % 2:
        fs.add_light(lr)
    fs.background = [0.5, 0.6, 0.8]
    fs.camera([U(9, 13), U(2, 6), U(9, 13)], [0, 1.0, 0], [0, 1, 0], U(30, 45), 1.0, 0.0, 14.0, 0.0, 1.0)
    return fs


def extreme_scene(seed):
    """Parameters at and beyond the edge of what the reference's scenes use: negative and zero sphere radii (the hollow-glass trick
    of the book uses a negative one), zero-area and inverted rectangles, flat and inverted cuboids, Metal fuzz 0 / 1 / 3, Dielectric
    ref_idx 1, 0.5, 0, -1.5, 2.4, FrostedDielectric fuzz 2, emitters with negative and huge power, albedos above 1, a medium of
    density 1e-6 and one of 1e6 — mostly NaN / inf territory, which has to flow through the device exactly as through the oracle."""
    rng = random.Random(19000 + seed)
    U = rng.uniform
    fs = F.FlatScene()

    def mat():
        c = rng.randrange(9)
        if c == 0:
            return fs.material(F.METAL, albedo=(U(0.5, 1.5), U(0.5, 1.5), U(0.5, 1.5)), fuzz=rng.choice([0.0, 1.0, 3.0]))
        if c == 1:
            return fs.material(F.DIELECTRIC, ref_idx=rng.choice([1.0, 0.5, 0.0, -1.5, 2.4, 1e-9, 1e9]))
        if c == 2:
            return fs.material(F.FROSTED, ref_idx=rng.choice([1.5, 0.7]), fuzz=rng.choice([0.0, 2.0]))
        if c == 3:
            return fs.material(F.DIFFUSE_LIGHT, texture=fs.solid(*rng.choice([(-3.0, 2.0, 1.0), (1e30, 1e30, 1e30), (0.0, 0.0, 0.0), (4.0, 4.0, 4.0)])))
        return fs.material(F.LAMBERTIAN, texture=fs.solid(U(0.0, 1.6), U(0.0, 1.6), U(0.0, 1.6)))

    fs.add_world(fs.rect(1, -20, 20, -20, 20, -1.0, 1.0, fs.material(F.LAMBERTIAN, texture=fs.solid(0.7, 0.7, 0.7))))
    ids = []
    for _ in range(rng.randrange(8, 16)):
        p = [U(-5, 5), U(0, 3), U(-5, 5)]
        k = rng.randrange(8)
        if k == 0:
            prim = fs.sphere(p, rng.choice([-0.8, 0.0, 1e-12, 1.0]), mat())
        elif k == 1:   # hollow sphere: outer positive, inner negative radius, same centre
            fs.add_world(fs.sphere(p, 1.0, fs.material(F.DIELECTRIC, ref_idx=1.5)))
            prim = fs.sphere(p, -0.9, fs.material(F.DIELECTRIC, ref_idx=1.5))
        elif k == 2:
            prim = fs.moving_sphere(p, p, 0.0, 1.0, rng.choice([-0.5, 0.7]), mat())   # "moving" without motion
        elif k == 3:
            prim = fs.rect(rng.randrange(3), p[0], p[0], p[1] - 1, p[1] + 1, p[2], 1.0, mat())        # zero width
        elif k == 4:
            prim = fs.rect(rng.randrange(3), p[0] + 1, p[0] - 1, p[1] - 1, p[1] + 1, p[2], -1.0, mat())  # inverted range
        elif k == 5:
            prim = fs.cuboid(p, [p[0] + 1.0, p[1], p[2] + 1.0], mat())                                  # flat
        elif k == 6:
            prim = fs.cuboid([p[0] + 1, p[1] + 1, p[2] + 1], p, mat())                                  # min > max
        else:
            prim = fs.sphere(p, U(0.4, 1.2), mat())
        if rng.random() < 0.5:
            ids.append(prim)
        else:
            fs.add_world(prim, chain=() if rng.random() < 0.6 else (("t", (U(-1, 1), 0.0, U(-1, 1))), ("r", U(-50, 50))))
    while len(ids) < 2:
        ids.append(fs.sphere([U(-5, 5), 4.0, U(-5, 5)], 0.5, mat()))
    fs.add_world(fs.build_bvh(ids, rng), bvh=True)
    iso = fs.material(F.ISOTROPIC, texture=fs.solid(1.3, 0.9, 0.2))
    fs.add_world(fs.sphere([U(-3, 3), 1.0, U(-3, 3)], 1.5, fs.material(F.DIELECTRIC, ref_idx=1.5)), medium=(rng.choice([1e-6, 1e6, 0.5]), iso))
    lr = fs.rect(1, -2, 2, -2, 2, 8.0, -1.0, fs.material(F.DIFFUSE_LIGHT, texture=fs.solid(5, 5, 5)))
    fs.add_world(lr)
    if seed % 2:
        fs.add_light(lr)
        gs = fs.sphere([U(-2, 2), 1.0, U(-2, 2)], rng.choice([1.0, -1.0, 0.0]), fs.material(F.DIELECTRIC, ref_idx=1.5))
        fs.add_world(gs); fs.add_light(gs)
    fs.background = [0.5, 0.6, 0.8]
    fs.camera([U(9, 13), U(2, 6), U(9, 13)], [0, 1.0, 0], [0, 1, 0], U(30, 45), 1.0, rng.choice([0.0, 0.3]), 14.0, 0.0, 1.0)
    return fs


def instancing_scene(seed):
    """One BVH referenced by several instances (as it is, behind Translate + RotateY, and as the boundary of a ConstantMedium), a second
    BVH used twice with different chains, and rectangle / cuboid lights behind rotating wrapper chains (the light list repeats the
    world's chain, as the reference's scenes do for their glass sphere)."""
    rng = random.Random(23000 + seed)
    U = rng.uniform
    fs = F.FlatScene()
    grey = fs.material(F.LAMBERTIAN, fs.solid(0.6, 0.6, 0.6))
    red = fs.material(F.LAMBERTIAN, fs.solid(0.7, 0.2, 0.2))
    glass = fs.material(F.DIELECTRIC, ref_idx=1.5)
    metal = fs.material(F.METAL, albedo=(0.8, 0.8, 0.7), fuzz=0.2)
    lamp = fs.material(F.DIFFUSE_LIGHT, fs.solid(6.0, 6.0, 6.0))
    iso = fs.material(F.ISOTROPIC, texture=fs.solid(0.8, 0.8, 0.9))
    fs.add_world(fs.rect(1, -25, 25, -25, 25, -1.0, 1.0, grey))

    def cluster(n):
        ids = []
        for _ in range(n):
            c = [U(-1.5, 1.5), U(0, 2), U(-1.5, 1.5)]
            k = rng.random()
            if k < 0.5:
                ids.append(fs.sphere(c, U(0.3, 0.7), rng.choice([grey, red, glass, metal])))
            elif k < 0.7:
                ids.append(fs.moving_sphere(c, [c[0] + U(-0.4, 0.4), c[1] + U(0, 0.4), c[2]], 0.0, 1.0, U(0.3, 0.6), red))
            else:
                ids.append(fs.cuboid([c[0] - 0.4, c[1] - 0.4, c[2] - 0.4], [c[0] + U(0.2, 0.6), c[1] + U(0.2, 0.6), c[2] + U(0.2, 0.6)], grey))
        return fs.build_bvh(ids, rng)

    a = cluster(rng.randrange(3, 12))
    fs.add_world(a, bvh=True)
    fs.add_world(a, bvh=True, chain=(("t", (U(4, 6), 0.0, U(-2, 2))), ("r", U(-60, 60))))
    fs.add_world(a, bvh=True, chain=(("t", (U(-6, -4), 0.5, U(-2, 2))),), medium=(U(0.3, 1.5), iso))
    b = cluster(rng.randrange(2, 6))
    fs.add_world(b, bvh=True, chain=(("r", U(-30, 30)), ("t", (0.0, 0.0, U(4, 6)))))
    fs.add_world(b, bvh=True, chain=(("t", (U(-2, 2), 2.5, U(-6, -4))), ("r", 90.0)))
    lch = (("t", (U(-2, 2), 7.0, U(-2, 2))), ("r", U(-45, 45)))
    lr = fs.rect(1, -1.5, 1.5, -1.0, 1.0, 0.0, -1.0, lamp)
    fs.add_world(lr, chain=lch); fs.add_light(lr, chain=lch)
    if seed % 2:
        cch = (("r", U(-80, 80)), ("t", (U(5, 7), 0.0, U(5, 7))))
        cb = fs.cuboid([-0.5, 0.0, -0.5], [0.5, 1.5, 0.7], lamp)
        fs.add_world(cb, chain=cch); fs.add_light(cb, chain=cch)
    fs.background = [0.1, 0.15, 0.25]
    fs.camera([U(10, 14), U(3, 7), U(10, 14)], [0, 1.0, 0], [0, 1, 0], U(35, 50), 1.0, 0.0, 15.0, 0.0, 1.0)
    return fs
