#!/usr/bin/env python3
"""Robustness fuzz of the in-repo JPEG / PPM loader (host/jpeg.cpp through rth_image_load): truncated and bit-flipped files must be
refused with a message or decoded — never crash or hang.  Every case runs in a forked child with an alarm.
  python tools/fuzz_jpeg.py N [seed]"""
import ctypes as C
import io
import os
import random
import signal
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402

from ray_tracing_amd import _ffi  # noqa: E402


def sources(rng):
    out = []
    for (w, h, sub, kw) in ((64, 48, 2, {}), (37, 21, 1, {}), (50, 33, 0, {}), (96, 80, 2, {"restart_marker_blocks": 5}), (40, 40, 0, {"restart_marker_rows": 1})):
        a = (np.random.RandomState(w).rand(h, w, 3) * 255).astype(np.uint8)
        b = io.BytesIO()
        Image.fromarray(a).save(b, "JPEG", subsampling=sub, quality=rng.choice([30, 75, 95]), **kw)
        out.append(b.getvalue())
    b = io.BytesIO()
    Image.fromarray((np.random.RandomState(1).rand(20, 30) * 255).astype(np.uint8)).save(b, "JPEG")
    out.append(b.getvalue())
    out.append(b"P6\n7 5\n255\n" + bytes(range(105)))
    return out


def main():
    n = int(sys.argv[1])
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    src = sources(rng)
    lib = _ffi.host()
    lib.rth_image_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.rth_image_free.argtypes = [C.POINTER(C.c_uint8)]
    tally = {"decoded": 0, "refused": 0}
    bad = []
    with tempfile.TemporaryDirectory() as td:
        for i in range(n):
            data = bytearray(rng.choice(src))
            k = rng.randrange(5)
            if k == 0:
                data = data[: rng.randrange(len(data))]
            elif k == 1:
                for _ in range(rng.randint(1, 8)):
                    data[rng.randrange(len(data))] = rng.randrange(256)
            elif k == 2:  # corrupt the headers (first 700 bytes: SOI, DQT, SOF, DHT, SOS)
                for _ in range(rng.randint(1, 4)):
                    data[rng.randrange(min(len(data), 700))] = rng.choice([0x00, 0xFF, 0x7F, 0x80, rng.randrange(256)])
            elif k == 3:
                at = rng.randrange(len(data)); data[at:at] = bytes(rng.randrange(256) for _ in range(rng.randint(1, 40)))
            else:
                at = rng.randrange(len(data)); del data[at:at + rng.randint(1, 60)]
            path = os.path.join(td, f"c{i}.jpg")
            open(path, "wb").write(bytes(data))
            pid = os.fork()
            if pid == 0:
                signal.alarm(10)
                p = C.POINTER(C.c_uint8)(); w = C.c_uint32(); h = C.c_uint32()
                rc = lib.rth_image_load(path.encode(), C.byref(p), C.byref(w), C.byref(h))
                if rc == 0:
                    assert 0 < w.value <= 65535 and 0 < h.value <= 65535
                    _ = bytes(C.cast(p, C.POINTER(C.c_uint8 * (w.value * h.value * 3))).contents)  # every byte readable
                    lib.rth_image_free(p)
                    os._exit(10)
                os._exit(11 if lib.rth_last_error() else 12)
            _, status = os.waitpid(pid, 0)
            code = status >> 8 if (status & 0x7F) == 0 else -(status & 0x7F)
            if code == 10:
                tally["decoded"] += 1
            elif code == 11:
                tally["refused"] += 1
            else:
                bad.append((i, k, code))
            os.unlink(path)
    print(f"{n} corrupted files: {tally}; crashed / hung / refused without a message: {len(bad)} {bad[:10]}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
