#!/usr/bin/env python3
"""Traversal statistics of the DEVICE code, from the CPU build of the device templates with event counters
(tests/hostsim, -DRT_SIM_COUNTERS): per cast, how many node classifications, undecided nodes, exact tests,
leaf tests ... a ray makes.   python tools/sim_counters.py random_scene [W H spp]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import ray_tracing_amd as rt  # noqa: E402
from ray_tracing_amd import _ffi  # noqa: E402

SIM_DIR = os.path.join(ROOT, "tests", "hostsim")
CSRC = os.path.join(ROOT, "ray-tracing_amd", "csrc")
LIB = os.path.join(SIM_DIR, "libhostsim_counters.so")
subprocess.run(["g++", "-std=c++17", "-O2", "-march=x86-64-v3", "-ffp-contract=off", "-fPIC", "-shared", "-DRT_HOST_SIM", "-DRT_SIM_COUNTERS",
                "-Wno-unknown-pragmas", "-I" + SIM_DIR, "-I" + CSRC, "-I" + os.path.join(ROOT, "include"), "-o", LIB,
                os.path.join(SIM_DIR, "sim.cpp")], check=True)
L = C.CDLL(LIB)
vp, u32 = C.c_void_p, C.c_uint32
L.sim_scene_create.argtypes, L.sim_scene_create.restype = [vp, C.POINTER(C.c_int)], vp
L.sim_render.argtypes = [vp, C.POINTER(_ffi.RtRenderParams), C.c_int, vp, vp]
name = sys.argv[1] if len(sys.argv) > 1 else "random_scene"
W, H, spp = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (96, 54, 4)
for precision in ("f64", "f32"):
    hs = rt.HostScene(name, W / H)
    rc = C.c_int()
    h = L.sim_scene_create(hs.desc_ptr, C.byref(rc))
    ctr = (C.c_ulonglong * 8).in_dll(L, "g_sim_counters")
    for i in range(8):
        ctr[i] = 0
    acc = np.zeros((H, W, 3))
    casts = C.c_uint64(0)
    p = _ffi.RtRenderParams(W, H, 0, spp, 50, 64, rt.DEFAULT_SEED, _ffi.RT_F64 if precision == "f64" else _ffi.RT_F32, 0)
    L.sim_render(h, C.byref(p), 0, acc.ctypes.data, C.byref(casts))
    n = max(casts.value, 1)
    names = ["node classifications (level 1)", "exact box tests", "leaf primitive tests", "leaf hits", "leaves culled by their box",
             "level-2 classifications", "reference-order walks (fallback)", "hit validations (parent box, exact)"]
    print(f"{name} {W}x{H}@{spp} {precision}: {casts.value} casts")
    for i, nm in enumerate(names):
        print(f"  {nm:34s} {ctr[i] / n:9.3f} per cast")
    hist = np.array(list((C.c_ulonglong * 64).in_dll(L, "g_sim_walk_hist")), dtype=np.float64)
    for i in range(64):
        (C.c_ulonglong * 64).in_dll(L, "g_sim_walk_hist")[i] = 0
    if hist.sum() > 0:
        cdf = np.cumsum(hist) / hist.sum()
        mean = (hist * (np.arange(64) * 8 + 4)).sum() / hist.sum()
        pct = {q: int(np.searchsorted(cdf, q) * 8 + 8) for q in (0.5, 0.75, 0.9, 0.98, 0.999)}
        print(f"  BVH walks: {int(hist.sum())}, node classifications per walk: mean {mean:.1f}, percentiles (upper bin edge) {pct}")
