#!/usr/bin/env python3
"""Robustness fuzz of the boundary's scene validation (CPU; rt_pack.hpp through tests/hostsim): randomly CORRUPTED flat
scenes — indices out of range, cyclic / shared BVH children, unknown kinds, NaN / zero / negative parameters, wrapper ranges
past the array — must be rejected with a message or, if the validator lets them through, be packed, probed and rendered
without a crash or a hang.  Every case runs in a forked child with an alarm.

  python tools/fuzz_validate.py FIRST LAST        # seeds; prints the seeds that crashed or hung, exit 1 if any"""
import ctypes as C
import math
import os
import random
import signal
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import ray_tracing_amd as rt  # noqa: E402
from ray_tracing_amd import _ffi  # noqa: E402
from flat_fuzz import random_scene, tie_scene  # noqa: E402
import test_hostsim  # noqa: E402

BAD_FLOATS = (float("nan"), float("inf"), -float("inf"), 0.0, -0.0, -1.0, 1e308, -1e308, 5e-324, 1e-300)


def weird_int(rng, n):
    return rng.choice([n, n + 1, 2 * n + 7, 0xFFFFFFFF, 0x7FFFFFFF, 0x80000000, rng.randrange(0, max(1, n)), 10 ** 6])


def corrupt(fs, rng):
    """1-3 random corruptions of a FlatScene (ray_tracing_amd/flat.py holds plain Python lists)."""
    what = []
    for _ in range(rng.randint(1, 3)):
        k = rng.randrange(14)
        if k == 0 and fs.prims:
            i = rng.randrange(len(fs.prims)); kind, m, d = fs.prims[i]
            fs.prims[i] = (rng.choice([6, 7, 255, 0xFFFFFFFF, rng.randrange(6)]), m, d); what.append("prim kind")
        elif k == 1 and fs.prims:
            i = rng.randrange(len(fs.prims)); kind, m, d = fs.prims[i]
            fs.prims[i] = (kind, weird_int(rng, len(fs.materials)), d); what.append("prim material")
        elif k == 2 and fs.prims:
            i = rng.randrange(len(fs.prims)); kind, m, d = fs.prims[i]; d = list(d)
            d[rng.randrange(10)] = rng.choice(BAD_FLOATS); fs.prims[i] = (kind, m, d); what.append("prim value")
        elif k == 3 and fs.materials:
            i = rng.randrange(len(fs.materials)); kind, t, al, fz, ri = fs.materials[i]
            fs.materials[i] = (rng.choice([6, 99, 0xFFFFFFFF, rng.randrange(6)]), weird_int(rng, len(fs.textures)), al, rng.choice(BAD_FLOATS + (fz,)),
                               rng.choice(BAD_FLOATS + (ri,))); what.append("material")
        elif k == 4 and fs.textures:
            i = rng.randrange(len(fs.textures)); kind, o, e, tb, col, sc = fs.textures[i]
            fs.textures[i] = (rng.choice([5, 77, 0xFFFFFFFF, rng.randrange(5)]), weird_int(rng, len(fs.textures)), weird_int(rng, len(fs.textures)),
                              weird_int(rng, max(len(fs.perlins), len(fs.images))), col, rng.choice(BAD_FLOATS + (sc,))); what.append("texture")
        elif k in (5, 6) and (fs.world if k == 5 else fs.lights):
            lst = fs.world if k == 5 else fs.lights
            i = rng.randrange(len(lst)); t = list(lst[i]); f = rng.randrange(7)
            if f == 0: t[0] = rng.choice([0, 1, 2, 3, 0xFFFFFFFF])
            elif f == 1: t[1] = weird_int(rng, max(len(fs.prims), len(fs.nodes)))
            elif f == 2: t[2] = weird_int(rng, len(fs.xforms))
            elif f == 3: t[3] = weird_int(rng, len(fs.xforms))
            elif f == 4: t[4] = rng.choice([0, 1, 2, 0xFFFFFFFF])
            elif f == 5: t[5] = weird_int(rng, len(fs.materials))
            else: t[6] = rng.choice(BAD_FLOATS)
            lst[i] = tuple(t); what.append("instance")
        elif k in (7, 8, 9) and fs.nodes:
            i = rng.randrange(len(fs.nodes)); side = rng.choice([1, 2]); n = len(fs.nodes)
            fs.nodes[i][side] = rng.choice([i, 0, rng.randrange(n), n, n + 5, -(len(fs.prims) + 1), -(len(fs.prims) + 9), -0x80000000, 0x7FFFFFFF,
                                            ~rng.randrange(max(1, len(fs.prims)))]); what.append("node child")
        elif k == 10 and fs.nodes:
            i = rng.randrange(len(fs.nodes)); fs.nodes[i][0][rng.randrange(6)] = rng.choice(BAD_FLOATS); what.append("node box")
        elif k == 11 and fs.xforms:
            i = rng.randrange(len(fs.xforms)); kind, v = fs.xforms[i]; v = list(v); v[rng.randrange(3)] = rng.choice(BAD_FLOATS)
            fs.xforms[i] = (rng.choice([kind, 2, 9, 0xFFFFFFFF]), v); what.append("xform")
        elif k == 12:
            cam = list(fs.cam); j = rng.randrange(len(cam))
            if isinstance(cam[j], list):
                cam[j] = list(cam[j]); cam[j][rng.randrange(3)] = rng.choice(BAD_FLOATS)
            else:
                cam[j] = rng.choice(BAD_FLOATS)
            fs.cam = tuple(cam); what.append("camera")
        elif k == 13:
            fs.background[rng.randrange(3)] = rng.choice(BAD_FLOATS); what.append("background")
    return what


def one_case(L, seed):
    rng = random.Random(seed)
    fs = tie_scene(seed % 12) if seed % 5 == 0 else random_scene(seed % 97)
    what = corrupt(fs, rng)
    try:
        host = rt.FlatHostScene(fs)
    except (OverflowError, TypeError, ValueError, C.ArgumentError):
        return "python", what  # the ctypes mirror already refuses the value (e.g. an index that is not a uint32)
    rc = C.c_int()
    h = L.sim_scene_create(host.desc_ptr, C.byref(rc))
    if not h:
        assert rc.value != 0 and L.sim_last_error(), "a rejection must carry a code and a message"
        return "rejected", what
    W = H = 6
    ids = np.zeros((H, W), dtype=np.int32); ts = np.zeros((H, W))
    for generic in (0, 1):
        L.sim_trace_primary(h, W, H, 0.3, 0.5, 1, generic, ids.ctypes.data, ts.ctypes.data)
    acc = np.zeros((H, W, 3)); casts = C.c_uint64(0)
    for precision in (1, 0):
        p = _ffi.RtRenderParams(W, H, 0, 1, 6, 2, 0x5EED0001, precision, 0)
        L.sim_render(h, C.byref(p), 0, acc.ctypes.data, C.byref(casts))
    L.sim_scene_destroy(h)
    return "accepted", what


def main():
    first, last = int(sys.argv[1]), int(sys.argv[2])
    test_hostsim._build()
    bad, tally = [], {"python": 0, "rejected": 0, "accepted": 0}
    import tempfile
    log = tempfile.NamedTemporaryFile(prefix="fuzz_validate_", suffix=".log", delete=False)  # (children append one line each)
    w = log.fileno()
    for seed in range(first, last):
        pid = os.fork()
        if pid == 0:
            signal.alarm(20)
            L = C.CDLL(test_hostsim.SIM_LIB)
            vp, u32, dbl = C.c_void_p, C.c_uint32, C.c_double
            L.sim_last_error.restype = C.c_char_p
            L.sim_scene_create.argtypes, L.sim_scene_create.restype = [vp, C.POINTER(C.c_int)], vp
            L.sim_scene_destroy.argtypes = [vp]
            L.sim_trace_primary.argtypes = [vp, u32, u32, dbl, dbl, u32, C.c_int, vp, vp]
            L.sim_render.argtypes = [vp, C.POINTER(_ffi.RtRenderParams), C.c_int, vp, vp]
            try:
                kind, what = one_case(L, seed)
                os.write(w, f"{seed} {kind} {','.join(what)}\n".encode())
                os._exit(0)
            except BaseException as e:  # noqa: BLE001
                os.write(w, f"{seed} exception {type(e).__name__}: {e}\n".encode())
                os._exit(3)
        _, status = os.waitpid(pid, 0)
        if status != 0:
            sig = status & 0x7F
            bad.append((seed, f"signal {sig}" if sig else f"exit {status >> 8}"))
    log.close()
    lines = open(log.name).read().splitlines()
    os.unlink(log.name)
    for line in lines:
        parts = line.split(" ", 2)
        if parts[1] in tally:
            tally[parts[1]] += 1
        else:
            print(line)
    print(f"{last - first} corrupted scenes: {tally}; crashed or hung: {len(bad)}")
    for seed, how in bad:
        rng = random.Random(seed)
        fs = tie_scene(seed % 12) if seed % 5 == 0 else random_scene(seed % 97)
        print(f"  seed {seed}: {how}  ({', '.join(corrupt(fs, rng))})")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
