"""In-repo baseline JPEG decoder (ray-tracing_amd/host/jpeg.cpp) against Pillow's libjpeg on files
made here, and on the reference's images/earthmap.jpg when the reference tree is present (data file, read only).
Decoders differ legitimately by an LSB or two (IDCT arithmetic, colour rounding; chroma upsampling filter for
subsampled files); the reference's decoder (jpeg-decoder 0.1.20) is not available: texel parity is unpinned."""
import ctypes as C
import io
import os

import numpy as np
import pytest

import ray_tracing_amd as rt
from ray_tracing_amd import _ffi

PIL = pytest.importorskip("PIL.Image")
REF_JPG = "/root/reference/images/earthmap.jpg"


def host():
    lib = C.CDLL(_ffi.LIB_HOST)
    lib.rth_image_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.rth_image_load.restype = C.c_int
    lib.rth_image_free.argtypes = [C.POINTER(C.c_uint8)]
    lib.rth_last_error.restype = C.c_char_p
    return lib


def decode(path):
    lib = host()
    p, w, h = C.POINTER(C.c_uint8)(), C.c_uint32(), C.c_uint32()
    if lib.rth_image_load(str(path).encode(), C.byref(p), C.byref(w), C.byref(h)) != 0:
        raise RuntimeError(lib.rth_last_error().decode())
    arr = np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    lib.rth_image_free(p)
    return arr


def picture(w, h, seed, smooth=False):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    if smooth:  # slowly varying colour: the choice of chroma upsampling filter hardly matters
        img = np.stack([127 + 100 * np.sin(x / 40.0 + seed), 127 + 100 * np.cos(y / 35.0), 60 + (x + 2 * y) * 0.7], axis=2)
    else:
        img = np.stack([127 + 120 * np.sin(x / 9.0 + seed), 127 + 120 * np.cos(y / 7.0), (x * 3 + y * 5) % 256], axis=2)
        img += rng.normal(0, 6, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("w,h,sub,kw", [
    (64, 48, 0, {}), (37, 21, 0, {}), (8, 8, 0, {}), (1, 1, 0, {}), (129, 65, 0, {"quality": 98}), (100, 60, 0, {"optimize": True}),
    (64, 48, 2, {}), (37, 21, 2, {}), (50, 33, 1, {}), (96, 80, 0, {"restart_marker_blocks": 5}), (70, 50, 2, {"restart_marker_rows": 1}),
])
def test_matches_libjpeg(tmp_path, w, h, sub, kw):
    src = picture(w, h, w + h, smooth=sub != 0)
    path = tmp_path / "t.jpg"
    try:
        PIL.fromarray(src).save(path, "JPEG", subsampling=sub, **dict({"quality": 90}, **kw))
    except TypeError:
        pytest.skip("this Pillow cannot write that variant")
    ref = np.asarray(PIL.open(path).convert("RGB")).astype(int)
    got = decode(path).astype(int)
    assert got.shape == ref.shape
    d = np.abs(got - ref)
    if sub == 0:
        assert d.max() <= 4 and d.mean() < 0.5, (d.max(), d.mean())
    else:  # libjpeg smooths upsampled chroma ("fancy upsampling"); this decoder replicates it
        assert d.mean() < 1.5 and d.max() <= 12, (d.mean(), d.max())


def test_greyscale(tmp_path):
    src = picture(40, 30, 1)[:, :, 0]
    path = tmp_path / "g.jpg"
    PIL.fromarray(src, "L").save(path, "JPEG", quality=92)
    ref = np.asarray(PIL.open(path).convert("RGB")).astype(int)
    d = np.abs(decode(path).astype(int) - ref)
    assert d.max() <= 3


def test_rejects_progressive_and_garbage(tmp_path):
    path = tmp_path / "p.jpg"
    PIL.fromarray(picture(32, 32, 2)).save(path, "JPEG", progressive=True)
    with pytest.raises(RuntimeError, match="baseline"):
        decode(path)
    raw = open(path, "rb").read()
    open(path, "wb").write(b"\xff\xd8" + b"\x00" * 64)
    with pytest.raises(RuntimeError):
        decode(path)
    ok = tmp_path / "ok.jpg"
    PIL.fromarray(picture(32, 32, 2)).save(ok, "JPEG")
    data = open(ok, "rb").read()
    open(path, "wb").write(data[: len(data) // 2])   # truncated scan: zeros are fed, no crash, an image or an error
    try:
        decode(path)
    except RuntimeError:
        pass
    assert raw[:2] == b"\xff\xd8"


def test_ppm_still_loads():
    arr = decode(rt.EARTHMAP)
    assert arr.shape == (512, 1024, 3)


@pytest.mark.skipif(not os.path.exists(REF_JPG), reason="reference tree not present")
def test_reference_earthmap_jpeg_equals_committed_raster():
    # the committed asset was decoded from this file with Pillow (tools/decode_earthmap.py)
    got = decode(REF_JPG).astype(int)
    ref = decode(rt.EARTHMAP).astype(int)
    d = np.abs(got - ref)
    assert got.shape == ref.shape and d.max() <= 4 and d.mean() < 0.3, (d.max(), d.mean())
    # and the scene constructor accepts the JPEG directly
    hs = rt.HostScene("earth", 1.0, image_path=REF_JPG)
    assert hs.desc.n_images == 1 and hs.desc.images[0].width == 1024 and hs.desc.images[0].height == 512
    hs.close()
