"""Known-answer tests that pin the CPU oracle (oracle/).

1. The reference's own unit tests — the 20 Vec3 tests of src/vec3.rs:259-395 — restated against
   the oracle's Vec3 (the ONLY golden vectors the reference holds; nothing there pins intersection,
   materials, pdfs or textures).
2. Analytic known answers derived from the formulas the reference implements: hit
   distances for axis-aligned rays, inclusive/strict range rules, pdf normalisation by Monte Carlo,
   reflect/refract identities, rotate round trip, sphere uv at poles/seam, Perlin with fixed tables.
"""
import ctypes as C
import math

import numpy as np
import pytest

from oracle_lib import lib


def _d(*v):
    return (C.c_double * len(v))(*[float(x) for x in v])


@pytest.fixture(scope="module")
def L():
    l = lib()
    dp = C.POINTER(C.c_double)
    l.orc_kat_vec3.argtypes = [C.c_int, dp, dp, C.c_double, dp]
    for name in ("orc_kat_sphere_hit", "orc_kat_moving_sphere_hit"):
        getattr(l, name).argtypes = [dp, dp, C.c_double, C.c_double, dp]
    l.orc_kat_rect_hit.argtypes = [C.c_int, dp, dp, C.c_double, C.c_double, dp]
    l.orc_kat_cuboid_hit.argtypes = [dp, dp, C.c_double, C.c_double, dp, C.POINTER(C.c_int)]
    l.orc_kat_aabb_hit.argtypes = [dp, dp, C.c_double, C.c_double]
    l.orc_kat_aabb_hit.restype = C.c_int
    l.orc_kat_rect_pdf.argtypes = [C.c_int, dp, dp, dp]
    l.orc_kat_rect_pdf.restype = C.c_double
    l.orc_kat_sphere_pdf.argtypes = [dp, dp, dp]
    l.orc_kat_sphere_pdf.restype = C.c_double
    l.orc_kat_rect_random.argtypes = [C.c_int, dp, dp, C.c_uint64, C.c_int, C.c_void_p]
    l.orc_kat_sphere_random.argtypes = [dp, dp, C.c_uint64, C.c_int, C.c_void_p]
    l.orc_kat_cosine_generate.argtypes = [dp, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
    l.orc_kat_cosine_value.argtypes = [dp, dp]
    l.orc_kat_cosine_value.restype = C.c_double
    l.orc_kat_onb.argtypes = [dp, dp]
    l.orc_kat_reflect.argtypes = [dp, dp, dp]
    l.orc_kat_refract.argtypes = [dp, dp, C.c_double, dp]
    l.orc_kat_schlick.argtypes = [C.c_double, C.c_double]
    l.orc_kat_schlick.restype = C.c_double
    l.orc_kat_sphere_uv.argtypes = [dp, dp]
    l.orc_kat_rotate_translate_hit.argtypes = [C.c_double, dp, dp, dp, C.c_double, C.c_double, dp]
    l.orc_kat_perlin_noise.argtypes = [C.c_void_p, C.c_void_p, dp]
    l.orc_kat_perlin_noise.restype = C.c_double
    l.orc_kat_perlin_turb.argtypes = [C.c_void_p, C.c_void_p, dp, C.c_int]
    l.orc_kat_perlin_turb.restype = C.c_double
    l.orc_kat_checker.argtypes = [dp, dp]
    l.orc_kat_checker_uv.argtypes = [C.c_double, C.c_double, dp]
    l.orc_kat_rng.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    return l


def vec3(L, op, a, b=(0, 0, 0), s=0.0):
    out = _d(0, 0, 0)
    L.orc_kat_vec3(op, _d(*a), _d(*b), float(s), out)
    return tuple(out)


# ---- 1. the reference's Vec3 unit tests (src/vec3.rs:259-395), exact equality as there ----------
def test_ref_vec3_add(L):  # test_add, test_add_assign :269-281
    assert vec3(L, 0, (1, 0, -1), (2, 4, 6)) == (3.0, 4.0, 5.0)


def test_ref_vec3_add_f64(L):  # test_add_f64, test_add_assign_f64 :284-296
    assert vec3(L, 1, (1, 0, -1), s=233.0) == (234.0, 233.0, 232.0)


def test_ref_vec3_sub(L):  # test_sub, test_sub_assign :299-311
    assert vec3(L, 2, (1, 0, -1), (2, 4, 6)) == (-1.0, -4.0, -7.0)


def test_ref_vec3_sub_f64(L):  # test_sub_f64, test_sub_assign_f64 :314-323
    assert vec3(L, 3, (1, 0, -1), s=1.0) == (0.0, -1.0, -2.0)


def test_ref_vec3_mul_is_dot(L):  # test_mul :326-328
    assert vec3(L, 4, (1, 0, -1), (1, 1, 1))[0] == 0.0


def test_ref_vec3_mul_f64(L):  # test_mul_assign, test_mul_f64 :331-340
    assert vec3(L, 5, (1, 0, -1), s=2.0) == (2.0, 0.0, -2.0)
    assert vec3(L, 5, (1, 0, -1), s=1.0) == (1.0, 0.0, -1.0)


def test_ref_vec3_div(L):  # test_div :343-345
    assert vec3(L, 6, (1, -2, 0), s=2.0) == (0.5, -1.0, 0.0)


def test_ref_vec3_elemul(L):  # test_elemul :348-353
    assert vec3(L, 7, (1, 2, 3), (1, 2, 3)) == (1.0, 4.0, 9.0)


def test_ref_vec3_cross(L):  # test_cross :356-361
    assert vec3(L, 8, (1, 2, 3), (2, 3, 4)) == (8.0 - 9.0, 6.0 - 4.0, 3.0 - 4.0)


def test_ref_vec3_neg(L):  # test_neg :364-366
    assert vec3(L, 9, (1, -2, 3)) == (-1.0, 2.0, -3.0)


def test_ref_vec3_squared_length(L):  # :369-371
    assert vec3(L, 10, (1, 2, 3))[0] == 14.0


def test_ref_vec3_length(L):  # :374-379
    assert vec3(L, 11, (3, 4, 5))[0] == math.sqrt(3.0 * 3.0 + 4.0 * 4.0 + 5.0 * 5.0)


def test_ref_vec3_unit(L):  # :382-388
    assert vec3(L, 12, (233, 0, 0)) == (1.0, 0.0, 0.0)
    assert vec3(L, 12, (-233, 0, 0)) == (-1.0, 0.0, 0.0)


def test_ref_vec3_unit_zero_is_nan(L):
    # :390-394 panics in the reference; the oracle (and the device) yield NaN so that the sample
    # is retried by the NaN loop of ray_tracing.rs:102-110 — documented deviation (oracle/README.md)
    assert all(math.isnan(c) for c in vec3(L, 12, (0, 0, 0)))


# ---- 2. analytic known answers -------------------------------------------------------------------
def sphere_hit(L, c_r, ray, tmin=0.001, tmax=1e300):
    out = _d(*([0] * 10))
    L.orc_kat_sphere_hit(_d(*c_r), _d(*ray), tmin, tmax, out)
    return list(out)


def rect_hit(L, axis, rect, ray, tmin=0.001, tmax=1e300):
    out = _d(*([0] * 10))
    L.orc_kat_rect_hit(axis, _d(*rect), _d(*ray), tmin, tmax, out)
    return list(out)


def test_sphere_axis_ray(L):
    # objects.rs:87-119: sphere at origin r=1, ray from z=-5 along +z (|d| = 2 -> t in units of |d|)
    h = sphere_hit(L, (0, 0, 0, 1), (0, 0, -5, 0, 0, 2, 0))
    assert h[0] == 1.0 and h[1] == 2.0
    assert h[2:5] == [0.0, 0.0, -1.0] and h[5:8] == [0.0, 0.0, -1.0]  # p and OUTWARD normal (no front-face flip)
    # inside the sphere: first root negative, second root returned, normal still outward
    h = sphere_hit(L, (0, 0, 0, 1), (0, 0, 0, 0, 0, 1, 0))
    assert h[0] == 1.0 and h[1] == 1.0 and h[5:8] == [0.0, 0.0, 1.0]


def test_sphere_strict_range_and_tangent(L):
    # strict comparisons (objects.rs:96,107): a root exactly at t_max / t_min is rejected
    assert sphere_hit(L, (0, 0, 0, 1), (0, 0, -5, 0, 0, 1, 0), 0.001, 4.0)[0] == 0.0  # first root t = 4 == t_max; second 6 > t_max
    assert sphere_hit(L, (0, 0, 0, 1), (0, 0, -5, 0, 0, 1, 0), 4.0, 7.0)[1] == 6.0     # first root == t_min rejected -> second
    # tangent ray: discriminant == 0 is a miss (strict `> 0.0`, objects.rs:93)
    assert sphere_hit(L, (0, 0, 0, 1), (1, 0, -5, 0, 0, 1, 0))[0] == 0.0


def test_sphere_uv_uses_first_root(L):
    # objects.rs:94-95,106-115: uv is computed from the FIRST root even when the second is returned
    h = sphere_hit(L, (0, 0, 0, 1), (0, 0, 0, 0, 0, 1, 0))  # returns t=+1, p=(0,0,1); first root t=-1 -> (0,0,-1)
    uv = _d(0, 0)
    L.orc_kat_sphere_uv(_d(0, 0, -1), uv)
    assert h[8] == uv[0] and h[9] == uv[1]


def test_sphere_uv_formula(L):
    # objects.rs:208-215: phi = atan2(z,x), theta = asin(y), u = 1-(phi+pi)/2pi, v = (theta+pi/2)/pi
    uv = _d(0, 0)
    for p, eu, ev in (((0, 1, 0), 0.5, 1.0), ((0, -1, 0), 0.5, 0.0), ((1, 0, 0), 0.5, 0.5), ((0, 0, 1), 0.25, 0.5),
                      ((0, 0, -1), 0.75, 0.5), ((-1, 0, 0), 0.0, 0.5)):
        L.orc_kat_sphere_uv(_d(*p), uv)
        assert uv[0] == pytest.approx(eu, abs=1e-15) and uv[1] == pytest.approx(ev, abs=1e-15), p


def test_moving_sphere_center_lerp(L):
    # objects.rs:154-158: centre moves linearly with ray.time
    out = _d(*([0] * 10))
    ms = _d(0, 0, 0, 0, 2, 0, 0, 1, 1)  # c1, c2, t1, t2, r
    L.orc_kat_moving_sphere_hit(ms, _d(0, 1, -5, 0, 0, 1, 0.5), 0.001, 1e300, out)  # at time 0.5 centre = (0,1,0)
    assert out[0] == 1.0 and out[1] == 4.0 and list(out[5:8]) == [0.0, 0.0, -1.0]
    L.orc_kat_moving_sphere_hit(ms, _d(0, 1, -5, 0, 0, 1, 0.0), 0.001, 1e300, out)  # at time 0 centre = (0,0,0): tangent -> miss
    assert out[0] == 0.0


def test_rect_inclusive_and_face(L):
    # objects.rs:227-244: RectXY x,y in [0,1], k=2, face -1; inclusive range and bounds tests
    r = (0, 1, 0, 1, 2, -1)
    h = rect_hit(L, 0, r, (0.25, 0.75, 0, 0, 0, 1, 0))
    assert h[0] == 1.0 and h[1] == 2.0 and h[5:8] == [0.0, 0.0, -1.0] and h[8:10] == [0.25, 0.75]
    assert rect_hit(L, 0, r, (0.25, 0.75, 0, 0, 0, 1, 0), 0.001, 2.0)[0] == 1.0   # t == t_max accepted
    assert rect_hit(L, 0, r, (0.25, 0.75, 0, 0, 0, 1, 0), 2.0, 9.0)[0] == 1.0     # t == t_min accepted
    assert rect_hit(L, 0, r, (1.0, 0.0, 0, 0, 0, 1, 0))[0] == 1.0                 # corner accepted
    assert rect_hit(L, 0, r, (1.0000001, 0.5, 0, 0, 0, 1, 0))[0] == 0.0
    # XZ (k on y) and YZ (k on x)
    h = rect_hit(L, 1, (0, 2, 0, 4, 5, 1), (1, 0, 1, 0, 2.5, 0, 0))
    assert h[0] == 1.0 and h[1] == 2.0 and h[5:8] == [0.0, 1.0, 0.0] and h[8:10] == [0.5, 0.25]
    h = rect_hit(L, 2, (0, 2, 0, 4, -3, 1), (0, 1, 3, -1, 0, 0, 0))
    assert h[0] == 1.0 and h[1] == 3.0 and h[5:8] == [1.0, 0.0, 0.0] and h[8:10] == [0.5, 0.75]


def test_rect_parallel_ray_is_miss(L):
    # d_k == 0 -> t = +-inf or NaN; every comparison fails
    assert rect_hit(L, 0, (0, 1, 0, 1, 2, 1), (0.5, 0.5, 0, 1, 0, 0, 0))[0] == 0.0
    assert rect_hit(L, 0, (0, 1, 0, 1, 2, 1), (0.5, 0.5, 2, 1, 0, 0, 0))[0] == 0.0  # 0/0


def test_cuboid_sides(L):
    # objects.rs:399-452 side order -z,+z,-y,+y,-x,+x ; :458-485 sequential closest
    out = _d(*([0] * 10))
    side = C.c_int()
    box = _d(0, 0, 0, 1, 2, 3)
    for ray, es, et, en in (((0.5, 1, -1, 0, 0, 1, 0), 0, 1.0, (0, 0, -1)), ((0.5, 1, 5, 0, 0, -1, 0), 1, 2.0, (0, 0, 1)),
                            ((0.5, -2, 1, 0, 1, 0, 0), 2, 2.0, (0, -1, 0)), ((0.5, 4, 1, 0, -1, 0, 0), 3, 2.0, (0, 1, 0)),
                            ((-1, 1, 1, 1, 0, 0, 0), 4, 1.0, (-1, 0, 0)), ((3, 1, 1, -1, 0, 0, 0), 5, 2.0, (1, 0, 0))):
        L.orc_kat_cuboid_hit(box, _d(*ray), 0.001, 1e300, out, C.byref(side))
        assert out[0] == 1.0 and side.value == es and out[1] == et and tuple(out[5:8]) == tuple(float(x) for x in en), ray
    # from inside: the exit face
    L.orc_kat_cuboid_hit(box, _d(0.5, 1, 1, 0, 0, 1, 0), 0.001, 1e300, out, C.byref(side))
    assert side.value == 1 and out[1] == 2.0


def test_cuboid_edge_tie_later_side_wins(L):
    # a ray through an edge hits two sides at the same t; rect tests are inclusive (`t <= closest`),
    # so the LATER side in tuple order replaces the earlier one
    out = _d(*([0] * 10))
    side = C.c_int()
    L.orc_kat_cuboid_hit(_d(0, 0, 0, 1, 1, 1), _d(-1, 0.5, -1, 1, 0, 1, 0), 0.001, 1e300, out, C.byref(side))
    assert out[0] == 1.0 and out[1] == 1.0 and side.value == 4  # sides 0 (-z) and 4 (-x) tie at t=1 -> 4


def test_aabb_loose_slab(L):
    # bvh.rs:26-58: each axis is tested against the ORIGINAL [t_min,t_max]; intervals are not intersected
    box = _d(0, 0, 0, 1, 1, 1)
    assert L.orc_kat_aabb_hit(box, _d(0.5, 0.5, -1, 0, 0, 1, 0), 0.001, 1e300) == 1
    assert L.orc_kat_aabb_hit(box, _d(2, 0.5, -1, 0, 0, 1, 0), 0.001, 1e300) == 0
    assert L.orc_kat_aabb_hit(box, _d(0.5, 0.5, -1, 0, 0, 1, 0), 0.001, 0.5) == 0
    # a diagonal ray whose per-axis intervals are disjoint (x: [2,3], z: [0.5,1.5]) still passes the loose test
    assert L.orc_kat_aabb_hit(box, _d(-2, 0.5, -0.5, 1, 0, 1, 0), 0.001, 1e300) == 1
    # 0/0 on an axis: NaN operands are ignored by f64::min/max -> the axis passes
    assert L.orc_kat_aabb_hit(box, _d(0.0, 0.5, -1, 0, 0, 1, 0), 0.001, 1e300) == 1


def test_rotate_translate_roundtrip(L):
    # transforms.rs: Translate(RotateY(Cuboid, a), off): hit point maps back onto the rotated box
    out = _d(*([0] * 10))
    ang, off = 15.0, (265.0, 0.0, 295.0)
    L.orc_kat_rotate_translate_hit(ang, _d(*off), _d(0, 0, 0, 165, 330, 165), _d(278, 278, -800, 0.05, -0.1, 1, 0), 0.001, 1e300, out)
    assert out[0] == 1.0
    p = np.array(out[2:5]) - np.array(off)
    th = math.radians(ang)
    # inverse rotation (world -> object): x' = cos x - sin z, z' = sin x + cos z  (transforms.rs:90-103)
    x = math.cos(th) * p[0] - math.sin(th) * p[2]
    z = math.sin(th) * p[0] + math.cos(th) * p[2]
    on_face = min(abs(x), abs(x - 165), abs(z), abs(z - 165), abs(p[1]), abs(p[1] - 330))
    assert on_face < 1e-9 and -1e-9 <= x <= 165 + 1e-9 and -1e-9 <= z <= 165 + 1e-9
    n = np.array(out[5:8])
    assert abs(np.linalg.norm(n) - 1.0) < 1e-12
    # angle 0 and zero offset = the bare cuboid
    a, b = _d(*([0] * 10)), _d(*([0] * 10))
    side = C.c_int()
    ray = _d(50, 60, -10, 0.1, 0.2, 1, 0)
    L.orc_kat_rotate_translate_hit(0.0, _d(0, 0, 0), _d(0, 0, 0, 165, 330, 165), ray, 0.001, 1e300, a)
    L.orc_kat_cuboid_hit(_d(0, 0, 0, 165, 330, 165), ray, 0.001, 1e300, b, C.byref(side))
    assert list(a)[:8] == list(b)[:8]


def test_reflect_refract_schlick(L):
    out = _d(0, 0, 0)
    L.orc_kat_reflect(_d(1, -1, 0), _d(0, 1, 0), out)  # vec3.rs:249-251
    assert tuple(out) == (1.0, 1.0, 0.0)
    # refract with eta = 1 leaves a unit direction unchanged (vec3.rs:252-257)
    v = np.array([0.6, -0.8, 0.0])
    L.orc_kat_refract(_d(*v), _d(0, 1, 0), 1.0, out)
    assert np.allclose(tuple(out), v, atol=1e-15)
    # Snell: sin(theta_t) = eta * sin(theta_i)
    L.orc_kat_refract(_d(*v), _d(0, 1, 0), 1.0 / 1.5, out)
    assert abs(out[0] - 0.6 / 1.5) < 1e-15 and abs(np.linalg.norm(tuple(out)) - 1.0) < 1e-15
    # schlick (materials.rs:6-9): r0 at normal incidence, 1 at grazing
    assert L.orc_kat_schlick(1.0, 1.5) == pytest.approx(0.04, abs=1e-15)
    assert L.orc_kat_schlick(0.0, 1.5) == pytest.approx(1.0, abs=1e-15)


def test_onb_is_orthonormal(L):
    out = _d(*([0] * 9))
    for w in ((0, 0, 1), (1, 0, 0), (0.95, 0.1, 0.1), (-3, 4, 12)):
        L.orc_kat_onb(_d(*w), out)  # pdf.rs:11-24
        m = np.array(out).reshape(3, 3)
        assert np.allclose(m @ m.T, np.eye(3), atol=1e-14)
        assert np.allclose(m[2], np.array(w) / np.linalg.norm(w), atol=1e-15)


def test_cosine_pdf_normalised_and_matches_generate(L):
    # pdf.rs:43-53: E_{d~generate}[1/pdf(d)] over the hemisphere = 2*pi  (importance-sampling identity)
    n = 200000
    dirs = np.zeros((n, 3))
    pdf = np.zeros(n)
    L.orc_kat_cosine_generate(_d(0.3, -0.5, 0.8), 7, n, dirs.ctypes.data, pdf.ctypes.data)
    w = np.array([0.3, -0.5, 0.8]) / np.linalg.norm([0.3, -0.5, 0.8])
    cosv = dirs @ w / np.linalg.norm(dirs, axis=1)
    assert (cosv >= -1e-12).all()
    assert np.allclose(pdf, np.maximum(cosv, 0) / math.pi, atol=1e-12)
    assert cosv.mean() == pytest.approx(2.0 / 3.0, abs=3e-3)  # E[cos] under cos/pi
    # value() normalises internally: scaling d does not change it (pdf.rs:44)
    assert L.orc_kat_cosine_value(_d(0, 0, 1), _d(0, 0, 5)) == pytest.approx(1 / math.pi, abs=1e-15)
    assert L.orc_kat_cosine_value(_d(0, 0, 1), _d(0, 0, -5)) == 0.0


def test_rect_light_pdf_integrates_to_one(L):
    # objects.rs:305-323 (RectXZ light of the Cornell box): directions from `random` have
    # E[1/pdf] = solid angle; and pdf = d^2 / (cos * A) analytically
    rect = (213, 343, 227, 332, 554, -1)
    origin = (278.0, 100.0, 278.0)
    n = 100000
    dirs = np.zeros((n, 3))
    L.orc_kat_rect_random(1, _d(*rect), _d(*origin), 11, n, dirs.ctypes.data)
    pts = dirs + np.array(origin)
    assert np.all(pts[:, 1] == 554.0) and pts[:, 0].min() >= 213 and pts[:, 0].max() < 343 and pts[:, 2].min() >= 227 and pts[:, 2].max() < 332
    area = (343 - 213) * (332 - 227)
    inv = []
    for d in dirs[:2000]:
        p = L.orc_kat_rect_pdf(1, _d(*rect), _d(*origin), _d(*d))
        d2 = float(d @ d)
        cos = abs(d[1]) / math.sqrt(d2)
        assert p == pytest.approx(d2 / (cos * area), rel=1e-12)
        inv.append(1.0 / p)
    # solid angle of the rectangle seen from origin, by direct quadrature
    xs = np.linspace(213, 343, 401)
    zs = np.linspace(227, 332, 401)
    xm, zm = np.meshgrid((xs[:-1] + xs[1:]) / 2, (zs[:-1] + zs[1:]) / 2)
    r2 = (xm - origin[0]) ** 2 + (554 - origin[1]) ** 2 + (zm - origin[2]) ** 2
    omega = ((554 - origin[1]) / r2 ** 1.5).sum() * (xs[1] - xs[0]) * (zs[1] - zs[0])
    assert np.mean(inv) == pytest.approx(omega, rel=0.02)
    # a direction that misses the rect has pdf 0
    assert L.orc_kat_rect_pdf(1, _d(*rect), _d(*origin), _d(1, 0, 0)) == 0.0


def test_sphere_light_pdf(L):
    # objects.rs:126-143: pdf = 1/(2 pi (1 - cos_max)) inside the cone, 0 outside; samples stay in the cone
    c_r = (190.0, 90.0, 190.0, 90.0)
    origin = (400.0, 300.0, 100.0)
    dist2 = sum((c_r[i] - origin[i]) ** 2 for i in range(3))
    cos_max = math.sqrt(1 - 90.0 ** 2 / dist2)
    n = 20000
    dirs = np.zeros((n, 3))
    L.orc_kat_sphere_random(_d(*c_r), _d(*origin), 5, n, dirs.ctypes.data)
    axis = (np.array(c_r[:3]) - np.array(origin)) / math.sqrt(dist2)
    cosv = dirs @ axis / np.linalg.norm(dirs, axis=1)
    assert cosv.min() >= cos_max - 1e-12
    assert cosv.mean() == pytest.approx((1 + cos_max) / 2, abs=2e-4)  # uniform in cos over [cos_max, 1]
    p = L.orc_kat_sphere_pdf(_d(*c_r), _d(*origin), _d(*axis))
    assert p == pytest.approx(1 / (2 * math.pi * (1 - cos_max)), rel=1e-12)
    assert L.orc_kat_sphere_pdf(_d(*c_r), _d(*origin), _d(*(-axis))) == 0.0


def _perlin_tables(seed=3):
    rs = np.random.RandomState(seed)
    perm = np.stack([rs.permutation(256) for _ in range(3)]).astype(np.int32)
    v = rs.uniform(-1, 1, (256, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return np.ascontiguousarray(perm), np.ascontiguousarray(v)


def _perlin_noise_np(perm, ranvec, p):
    # texture.rs:139-165 restated independently in numpy
    f = np.floor(p)
    u, v, w = p - f
    i, j, k = (int(x) for x in f)
    uu, vv, ww = (t * t * (3 - 2 * t) for t in (u, v, w))
    acc = 0.0
    for di in range(2):
        for dj in range(2):
            for dk in range(2):
                g = ranvec[perm[0][(i + di) & 255] ^ perm[1][(j + dj) & 255] ^ perm[2][(k + dk) & 255]]
                acc += (di * uu + (1 - di) * (1 - uu)) * (dj * vv + (1 - dj) * (1 - vv)) * (dk * ww + (1 - dk) * (1 - ww)) * \
                    float(g @ np.array([u - di, v - dj, w - dk]))
    return acc


def test_perlin_fixed_tables(L):
    perm, ranvec = _perlin_tables()
    for p in ((0.3, 1.7, -2.4), (-5.5, 0.25, 100.125), (255.9, -255.9, 3.0), (0.0, 0.0, 0.0)):
        got = L.orc_kat_perlin_noise(perm.ctypes.data, ranvec.ctypes.data, _d(*p))
        assert got == pytest.approx(_perlin_noise_np(perm, ranvec, np.array(p, dtype=np.float64)), abs=1e-14), p
    # lattice points: every weight multiplies a zero offset or a zero blend -> exactly the (0,0,0) corner term = 0
    assert L.orc_kat_perlin_noise(perm.ctypes.data, ranvec.ctypes.data, _d(4.0, -7.0, 9.0)) == 0.0
    # turb (texture.rs:166-176): |sum_{i<7} 2^-i noise(2^i p)|
    p = np.array([0.3, 1.7, -2.4])
    exp = abs(sum(0.5 ** i * _perlin_noise_np(perm, ranvec, p * 2.0 ** i) for i in range(7)))
    assert L.orc_kat_perlin_turb(perm.ctypes.data, ranvec.ctypes.data, _d(*p), 7) == pytest.approx(exp, abs=1e-13)


def test_checker_textures(L):
    out = _d(0)
    # texture.rs:42-49: sines < 0 -> odd (0), else even (1)
    for p in ((0.1, 0.1, 0.1), (0.4, 0.1, 0.1), (-0.1, 0.1, 0.1), (0.0, 0.3, 0.3)):
        L.orc_kat_checker(_d(*p), out)
        s = math.sin(10 * p[0]) * math.sin(10 * p[1]) * math.sin(10 * p[2])
        assert out[0] == (0.0 if s < 0 else 1.0), p
    # texture.rs:27-33: (u/0.01) as i32 % 2 == (v/0.01) as i32 % 2 -> odd (0) else even (1)
    for u, v in ((0.005, 0.005), (0.015, 0.005), (0.015, 0.015), (0.995, 0.505)):
        L.orc_kat_checker_uv(u, v, out)
        assert out[0] == (0.0 if int(u / 0.01) % 2 == int(v / 0.01) % 2 else 1.0), (u, v)


def test_rng_stream_spec(L):
    # The generator both the oracle and the device use (DESIGN.md "RNG"): xoshiro128++ seeded from
    # splitmix64(seed + key * golden), key = pixel << 32 | sample.  Independent restatement here.
    M = (1 << 64) - 1

    def splitmix(z):
        z = (z + 0x9E3779B97F4A7C15) & M
        r = z
        r = ((r ^ (r >> 30)) * 0xBF58476D1CE4E5B9) & M
        r = ((r ^ (r >> 27)) * 0x94D049BB133111EB) & M
        return z, r ^ (r >> 31)

    def stream(seed, pixel, sample, n):
        key = (pixel << 32) | sample
        z = (seed + key * 0x9E3779B97F4A7C15) & M
        z, a = splitmix(z)
        z, b = splitmix(z)
        s = [a & 0xFFFFFFFF, a >> 32, b & 0xFFFFFFFF, b >> 32]
        if not any(s):
            s[0] = 1
        rotl = lambda x, k: ((x << k) | (x >> (32 - k))) & 0xFFFFFFFF
        out = []
        for _ in range(n):
            out.append((rotl((s[0] + s[3]) & 0xFFFFFFFF, 7) + s[0]) & 0xFFFFFFFF)
            t = (s[1] << 9) & 0xFFFFFFFF
            s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t
            s[3] = rotl(s[3], 11)
        return out

    for seed, pixel, sample in ((0x5EED0001, 0, 0), (0x5EED0001, 1048575, 1023), (0, 7, 9), ((1 << 64) - 1, 4194303, 4095)):
        n = 16
        u32 = np.zeros(n, dtype=np.uint32)
        f64 = np.zeros(n)
        L.orc_kat_rng(seed, pixel, sample, n, u32.ctypes.data, f64.ctypes.data)
        exp = stream(seed, pixel, sample, 2 * n)
        assert u32.tolist() == exp[:n]
        # gen::<f64>() = 53 high bits of a u64 (hi word drawn first) * 2^-53
        expf = [(((exp[2 * i] << 32) | exp[2 * i + 1]) >> 11) * 2.0 ** -53 for i in range(n)]
        assert f64.tolist() == expf
