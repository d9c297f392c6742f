"""The DEVICE code on the CPU (tests/hostsim: rt_device.hpp / rt_path.hpp / rt_pack.hpp compiled by g++
with RT_HOST_SIM) against the oracle and the golden fixtures.  With T = double, primitive ids and hit
distances agree BIT FOR BIT and every path takes the same decisions (same arithmetic order under
-ffp-contract=off, same keyed RNG, same libm: identical cast counts); radiance agrees to 1e-12 relative,
not bitwise, because the iterative throughput form multiplies the attenuations front to back while the
reference's recursion (ray_tracing.rs:38-52) multiplies them back to front.  This pins the device
templates — including every feature-mask instantiation — without a GPU; the -m gpu tests then only
have to establish that the HIP build of the same templates behaves the same on the hardware."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import ray_tracing_amd as rt
from ray_tracing_amd import _ffi
from oracle_lib import DEFAULT_SEED, OracleScene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM_DIR = os.path.join(ROOT, "tests", "hostsim")
SIM_LIB = os.path.join(SIM_DIR, "libhostsim.so")
CSRC = os.path.join(ROOT, "ray-tracing_amd", "csrc")
SCENES = rt.SCENES
RAD_RTOL = 1e-12  # association of the throughput product only; decisions are identical


def _build():
    deps = [os.path.join(SIM_DIR, f) for f in ("sim.cpp", "rt_hostsim_shim.hpp")] + \
           [os.path.join(CSRC, f) for f in ("rt_device.hpp", "rt_path.hpp", "rt_pack.hpp", "rt_variants.hpp")] + \
           [os.path.join(ROOT, "include", "rt_hip.h")]
    if os.path.exists(SIM_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(SIM_LIB) for d in deps):
        return
    cmd = ["g++", "-std=c++17", "-O2", "-march=x86-64-v3", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-DRT_HOST_SIM",
           "-Wall", "-Wextra", "-Wno-unknown-pragmas", "-I" + SIM_DIR, "-I" + CSRC, "-I" + os.path.join(ROOT, "include"),
           "-o", SIM_LIB, os.path.join(SIM_DIR, "sim.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.fixture(scope="module")
def sim():
    _build()
    L = C.CDLL(SIM_LIB)
    vp, u32, dbl = C.c_void_p, C.c_uint32, C.c_double
    L.sim_last_error.restype = C.c_char_p
    L.sim_scene_create.argtypes, L.sim_scene_create.restype = [vp, C.POINTER(C.c_int)], vp
    L.sim_scene_destroy.argtypes = [vp]
    L.sim_scene_features.argtypes, L.sim_scene_features.restype = [vp], u32
    L.sim_trace_primary.argtypes = [vp, u32, u32, dbl, dbl, u32, C.c_int, vp, vp]
    L.sim_sample_pixel.argtypes = [vp, C.POINTER(_ffi.RtRenderParams), u32, u32, C.c_int, vp, vp]
    L.sim_render.argtypes = [vp, C.POINTER(_ffi.RtRenderParams), C.c_int, vp, vp]
    return L


class SimScene:
    def __init__(self, L, name, aspect=1.0):
        self.L = L
        self.host = rt.HostScene(name, aspect)
        rc = C.c_int()
        self.h = L.sim_scene_create(self.host.desc_ptr, C.byref(rc))
        assert self.h, L.sim_last_error()

    @classmethod
    def from_host(cls, L, host):
        self = cls.__new__(cls)
        self.L, self.host = L, host
        rc = C.c_int()
        self.h = L.sim_scene_create(host.desc_ptr, C.byref(rc))
        assert self.h, L.sim_last_error()
        return self

    def primary(self, W, H, time, xi, precision="f64", generic=False):
        ids = np.zeros((H, W), dtype=np.int32)
        ts = np.zeros((H, W))
        self.L.sim_trace_primary(self.h, W, H, time, xi, 1 if precision == "f64" else 0, int(generic), ids.ctypes.data, ts.ctypes.data)
        return ids, ts

    def _params(self, W, H, s0, n, precision, max_depth=50, seed=DEFAULT_SEED, max_nan_retries=64):
        return _ffi.RtRenderParams(W, H, s0, n, max_depth, max_nan_retries, seed, 1 if precision == "f64" else 0, 0)

    def samples(self, W, H, x, y, s0, n, precision="f64", generic=False):
        out = np.zeros((n, 3))
        casts = C.c_uint64(0)
        p = self._params(W, H, s0, n, precision)
        self.L.sim_sample_pixel(self.h, C.byref(p), x, y, int(generic), out.ctypes.data, C.byref(casts))
        return out, casts.value

    def render(self, W, H, s0, n, precision="f64", generic=False, max_depth=50, max_nan_retries=64):
        acc = np.zeros((H, W, 3))
        casts = C.c_uint64(0)
        p = self._params(W, H, s0, n, precision, max_depth, max_nan_retries=max_nan_retries)
        self.L.sim_render(self.h, C.byref(p), int(generic), acc.ctypes.data, C.byref(casts))
        return acc, casts.value

    def __del__(self):
        try:
            self.L.sim_scene_destroy(self.h)
        except Exception:
            pass


def test_feature_masks(sim):
    F = dict(BVH=1, MEDIUM=2, MOVING=4, XFORM=8, SPHERE=16, RECT=32, CUBOID=64, METAL=128, DIELECTRIC=256, FROSTED=512, LIGHTS=1024,
             TEXTURED=2048, UV=4096)
    def f(name):
        sc = SimScene(sim, name)  # keep the scene alive across the call
        return sim.sim_scene_features(sc.h)

    assert f("cornell_box") == F["XFORM"] | F["SPHERE"] | F["RECT"] | F["CUBOID"] | F["DIELECTRIC"] | F["LIGHTS"]
    assert f("cornell_smoke") == F["XFORM"] | F["RECT"] | F["CUBOID"] | F["MEDIUM"]
    assert f("random_scene") == F["BVH"] | F["SPHERE"] | F["MOVING"] | F["METAL"] | F["DIELECTRIC"] | F["TEXTURED"]
    assert f("earth") == F["SPHERE"] | F["TEXTURED"] | F["UV"]
    fs = f("final_scene")
    assert fs & F["BVH"] and fs & F["MEDIUM"] and fs & F["LIGHTS"] and fs & F["UV"] and fs & F["MOVING"] and fs & F["CUBOID"]
    assert f("random_scene_light") & F["FROSTED"]


@pytest.mark.parametrize("name", SCENES)
def test_sim_primary_bit_exact(sim, name, golden_dir):
    g = np.load(os.path.join(golden_dir, f"primary_{name}.npz"))
    sc = SimScene(sim, name)
    for generic in (False, True):
        for tag, tm in (("t0", 0.0), ("t05", 0.5)):
            ids, ts = sc.primary(64, 64, tm, float(g["medium_xi"]), "f64", generic)
            assert np.array_equal(ids, g[f"ids_{tag}"]), (name, generic, tag)
            assert np.array_equal(ts, g[f"t_{tag}"]), (name, generic, tag)


@pytest.mark.parametrize("name", SCENES)
def test_sim_samples_bit_exact(sim, name, golden_dir):
    g = np.load(os.path.join(golden_dir, f"samples_{name}.npz"))["rgb"]
    sc = SimScene(sim, name)
    for generic in (False, True):
        got, casts = sc.samples(48, 48, 20, 13, 0, 64, "f64", generic)
        assert np.allclose(got, g, rtol=RAD_RTOL, atol=0), (name, generic, np.abs(got - g).max())
        assert casts >= 64


@pytest.mark.parametrize("name", ("cornell_box", "cornell_smoke", "random_scene", "final_scene", "simple_light", "random_scene_light"))
def test_sim_frame_equals_oracle(sim, name):
    W, H, s0, n = 28, 20, 5, 3
    sc = SimScene(sim, name, W / H)
    got, casts = sc.render(W, H, s0, n)
    ref, cnt = OracleScene(name, W / H).render(W, H, s0, n, threads=8)
    assert np.allclose(got, ref, rtol=RAD_RTOL, atol=0), np.abs(got - ref).max()
    assert casts == cnt["casts"]


def test_sim_f32_is_close(sim, golden_dir):
    # the f32 instantiation on the CPU: same ids almost everywhere, t within 5e-4, sane radiance
    for name in ("cornell_box", "final_scene"):
        g = np.load(os.path.join(golden_dir, f"primary_{name}.npz"))
        sc = SimScene(sim, name)
        ids, ts = sc.primary(64, 64, 0.0, 0.5, "f32")
        same = ids == g["ids_t0"]
        assert same.mean() > 0.995
        hit = same & (ids >= 0)
        assert (np.abs(ts[hit] - g["t_t0"][hit]) / g["t_t0"][hit]).max() < 5e-4
        a64, _ = sc.render(16, 16, 0, 64, "f64")
        a32, _ = sc.render(16, 16, 0, 64, "f32")
        assert np.isfinite(a32).all() and abs(a32.mean() - a64.mean()) < 0.25 * a64.mean()


def test_sim_depth_limits(sim):
    sc = SimScene(sim, "cornell_box")
    assert sc.render(8, 8, 0, 4, max_depth=0)[0].max() == 0.0
    a, _ = sc.render(16, 16, 0, 4, max_depth=1)
    assert set(np.unique(a).tolist()) <= {0.0, 15.0, 30.0, 45.0, 60.0}


def test_packer_intervals_of_the_named_scenes(sim):
    # rt_pack.hpp: the reference's own trees are nested and contain their primitives, so every named scene keeps the fast walk over
    # its whole shutter hull, and geometric node culling for (at least) the (0, 1) its boxes were built for
    sim.sim_scene_intervals.argtypes = [C.c_void_p, C.c_void_p]
    for name in rt.SCENES:
        sc = SimScene(sim, name)
        v = np.zeros(6)
        sim.sim_scene_intervals(sc.h, v.ctypes.data)
        assert (v[0], v[1]) == (0.0, 1.0) and (v[2], v[3]) == (0.0, 1.0), (name, v)
        assert v[4] <= 0.0 and v[5] >= 1.0, (name, v)
        if name == "random_scene":  # the only named scene with moving spheres INSIDE a BVH: its boxes hold them for (0, 1) and no longer
            assert -0.01 < v[4] and v[5] < 1.01, (name, v)
        else:
            assert v[4] < -1e300 and v[5] > 1e300, (name, v)


def test_sim_nan_retry_cap(sim):
    # ray_tracing.rs:102-110 retries a NaN sample for ever; this build gives up after max_nan_retries and takes the sample as
    # zero (stats[3]).  final_scene produces NaN samples, and with a cap of 1 some samples NaN twice in a row: same frames,
    # same casts as the oracle with the same cap — and fewer casts than with the default cap
    sc = SimScene(sim, "final_scene")
    oc = OracleScene("final_scene", 1.0)
    got, casts = sc.render(24, 24, 0, 24, max_nan_retries=1)
    ref, cnt = oc.render(24, 24, 0, 24, threads=8, max_nan_retries=1)
    assert cnt["nan_retries"] > 0 and cnt["dropped"] > 0, cnt
    assert casts == cnt["casts"]
    assert np.allclose(got, ref, rtol=RAD_RTOL, atol=0, equal_nan=True)
    ref64, cnt64 = oc.render(24, 24, 0, 24, threads=8)
    assert cnt64["dropped"] == 0 and cnt64["casts"] > cnt["casts"] and cnt64["nan_retries"] > cnt["nan_retries"]


def test_light_with_empty_range_is_rejected(sim):
    # gen_range(low, high) panics in the reference on low >= high and on a non-finite width; the device would redraw for ever.
    # Rejected at scene creation, for both arithmetic modes (a range that is empty only after rounding to f32 counts)
    hs = rt.HostScene("cornell_box", 1.0)
    li = hs.desc.lights[0].geom_index

    def create(mutate):
        bad = _ffi.RtSceneDesc.from_buffer_copy(hs.desc)
        prims = (_ffi.RtPrim * bad.n_prims)(*[bad.prims[i] for i in range(bad.n_prims)])
        bad.prims = C.cast(prims, C.POINTER(_ffi.RtPrim))
        mutate(bad, prims)
        rc = C.c_int()
        h = sim.sim_scene_create(C.byref(bad), C.byref(rc))
        return h, rc.value, sim.sim_last_error()

    def set_d(k, v):
        def f(bad, prims):
            prims[li].d[k] = v
        return f

    x1 = hs.desc.prims[li].d[0]
    for mutate in (set_d(1, x1), set_d(1, x1 - 1.0), set_d(1, float("nan")), set_d(0, float("-inf")), set_d(1, float("inf")),
                   set_d(1, x1 * (1.0 + 1e-12)),           # non-empty in f64, empty in f32
                   set_d(3, 1e300)):                        # finite in f64, infinite in f32
        h, rc, msg = create(mutate)
        assert not h and rc == 1 and b"range" in msg, msg

    def shutter(t1, t2):
        def f(bad, prims):
            bad.camera.t1, bad.camera.t2 = t1, t2
        return f

    for mutate in (shutter(0.5, 0.5), shutter(1.0, 0.0), shutter(0.0, float("inf")), shutter(float("nan"), 1.0), shutter(0.0, 1e-60)):
        h, rc, msg = create(mutate)
        assert not h and rc == 1 and b"shutter" in msg, msg
    h, rc, msg = create(shutter(0.25, 0.75))
    assert h and rc == 0
    sim.sim_scene_destroy(h)


@pytest.mark.parametrize("name", ("random_scene", "final_scene", "random_scene_light"))
def test_sim_bvh_cull_is_invisible(sim, name):
    # The device walks BVHs with a division-free node classification + conservative culling (rt_device.hpp
    # aabb_classify) instead of the reference's loose test on every node: ids and t must still be bit-exact
    # on a dense probe (many grazing rays along silhouettes), for two times and a non-square aspect.
    W, H = 224, 126
    sc = SimScene(sim, name, W / H)
    oc = OracleScene(name, W / H)
    for tm in (0.0, 0.73):
        ids, ts = sc.primary(W, H, tm, 0.5)
        oi, ot = oc.trace_primary(W, H, tm, 0.5)
        assert np.array_equal(ids, oi) and np.array_equal(ts, ot)


def test_checker_sign_decision_is_the_sine_product(sim):
    # rt_device.hpp: checker_is_odd<double> decides CheckerTexture's branch (texture.rs:42-49) from the SIGNS of the three sines
    # (one range-reduction step each; the sines themselves where that step is not conclusive).  Against the plain product of
    # libm sines: uniform arguments, arguments next to multiples of pi (where the sign flips), zeros, huge and non-finite ones.
    rng = np.random.default_rng(20260311)
    n = 400_000
    a = rng.uniform(-1.2e4, 1.2e4, size=(n, 3))
    k = rng.integers(-300_000, 300_000, size=200_000).astype(np.float64)
    near = k * np.pi
    ulps = rng.integers(-3, 4, size=near.shape)
    for _ in range(3):
        near = np.where(ulps > 0, np.nextafter(near, np.inf), np.where(ulps < 0, np.nextafter(near, -np.inf), near))
        ulps = ulps - np.sign(ulps)
    b = rng.uniform(-50.0, 50.0, size=(near.size, 3))
    b[:, 0] = near
    b[::2, 1] = near[::-1][::2] * 0.5  # odd multiples of pi/2 as well
    special = np.array([[0.0, 1.0, 1.0], [-0.0, 1.0, 1.0], [1.0, 0.0, -1.0], [1e300, 1.0, 1.0], [np.inf, 1.0, 1.0], [np.nan, 1.0, 1.0],
                        [1048575.9, 1.0, -1.0], [1048576.0, -1.0, 1.0], [-1048577.0, 2.0, 4.0], [5e-324, -1.0, 1.0], [-5e-324, 1.0, 1.0],
                        [4.0, 4.0, 4.0], [-4.0, 4.0, 4.0], [np.pi, np.pi, np.pi], [-np.pi, 1.0, 1.0], [2 * np.pi, 1.0, 1.0]])
    xyz = np.ascontiguousarray(np.concatenate([a, b, special]))
    m = xyz.shape[0]
    fast, plain, sure = (np.zeros(m, np.uint8) for _ in range(3))
    sim.sim_checker.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    sim.sim_checker(xyz.ctypes.data, m, fast.ctypes.data, plain.ctypes.data, sure.ctypes.data)
    assert np.array_equal(fast, plain), xyz[fast != plain][:5]
    assert sure[:n].mean() > 0.999999 and 0.2 < plain[:n].mean() < 0.8  # the quick route decides practically every ordinary argument
    assert sure[n:n + near.size].mean() < 0.9                           # and declines next to the multiples of pi (both routes ran)
