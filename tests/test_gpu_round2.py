"""GPU tests added in round 2 (all through the C-ABI): the exchange step, the one-call rgb8 entry point, progressive
rendering with checkpoint/resume on the device, an ImageTexture fed through the in-repo JPEG decoder, the job-counter
ring under more launches in flight than it has slots, and property runs at BASELINE's full frame sizes."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

import ray_tracing_amd as rt
from frame_check import frame_close
from oracle_lib import OracleScene, resolve as oracle_resolve
from ray_tracing_amd import _ffi, flat

pytestmark = pytest.mark.gpu


def _where():
    """name of the running test (the tag frame_close logs under)"""
    return os.environ.get("PYTEST_CURRENT_TEST", "?").split("::")[-1].split(" ")[0]


@pytest.fixture(scope="module")
def scenes(gpu):
    cache = {}

    def get(name, aspect=1.0):
        key = (name, aspect)
        if key not in cache:
            hs = rt.HostScene(name, aspect)
            cache[key] = (hs, rt.DeviceScene(hs, 0))
        return cache[key][1]

    yield get
    for hs, ds in cache.values():
        ds.close()
        hs.close()


# ---- exchange step (SURVEY §8 b-4 / e): rt_comm_* + rt_allreduce_* ------------------------------------------
def test_exchange_single_rank_is_identity_and_errors_are_codes(scenes):
    lib = _ffi.hip()
    comm = rt.Comm(0, 1)  # rt_comm_unique_id + rt_comm_create on device 0: a 1-rank RCCL communicator
    ds = scenes("cornell_box")
    acc = torch.zeros((64, 64, 3), dtype=torch.float32, device="cuda:0")
    st = torch.zeros(4, dtype=torch.int64, device="cuda:0")
    ds.render_into(acc, 0, 4, precision="f64", stats=st)
    before, st_before = acc.clone(), st.clone()
    comm.all_reduce(acc)           # sum over one rank: unchanged, bit for bit
    comm.all_reduce_stats(st)
    torch.cuda.synchronize()
    assert torch.equal(acc, before) and torch.equal(st, st_before)
    # a full-size buffer (the 12.6 MB of BASELINE's frame)
    big = torch.rand((1024, 1024, 3), dtype=torch.float32, device="cuda:0")
    ref = big.clone()
    comm.all_reduce(big)
    torch.cuda.synchronize()
    assert torch.equal(big, ref)
    # errors come back as codes
    assert lib.rt_allreduce_accum(None, C.c_void_p(acc.data_ptr()), 12, None) == 1
    assert lib.rt_allreduce_accum(comm._h, None, 12, None) == 1
    assert lib.rt_allreduce_accum(comm._h, None, 0, None) == 0
    assert lib.rt_allreduce_stats(comm._h, None, None) == 1
    h = C.c_void_p()
    buf = (C.c_uint8 * 128)()
    assert lib.rt_comm_create(buf, 2, 2, C.byref(h)) == 1 and b"rank" in lib.rt_last_error()
    assert lib.rt_comm_create(None, 1, 0, C.byref(h)) == 1
    assert lib.rt_comm_adopt(None, C.byref(h)) == 1
    assert lib.rt_comm_destroy(None) == 0
    comm.close()


def test_render_shards_plus_exchange_equal_one_render(scenes):
    # what N GPUs do, on one: N disjoint sample shards rendered into separate buffers, summed, == one render of all samples.
    # (the sum over buffers is what the all-reduce computes; the communicator here has one rank)
    ds = scenes("cornell_box")
    W = H = 96
    whole = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    ds.render_into(whole, 0, 64, precision="f64")
    parts = []
    for r in range(8):
        b, c = rt.spp_shard(64, r, 8)
        p = torch.zeros_like(whole)
        ds.render_into(p, b, c, precision="f64")
        parts.append(p)
    total = torch.stack(parts).sum(0)
    assert torch.allclose(total, whole, rtol=2e-6, atol=1e-6)


# ---- rt_render_rgb8: ray_tracing.rs:95-118 in one call -------------------------------------------------------
@pytest.mark.parametrize("name,W,H,spp", (("cornell_box", 64, 48, 32), ("random_scene", 72, 40, 4), ("earth", 33, 21, 3)))
def test_render_rgb8_equals_render_then_resolve(name, W, H, spp, scenes):
    ds = scenes(name, W / H)
    acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    st_a = torch.zeros(4, dtype=torch.int64, device="cuda:0")
    st_b = torch.zeros_like(st_a)
    ds.render_into(acc, 5, spp, precision="f64", stats=st_a)
    want = rt.resolve(acc, spp)
    got = ds.render_rgb8(W, H, 5, spp, precision="f64", stats=st_b)
    # sample chunks added by float atomics can differ in the last bit of a sum: allow one grey level on a few pixels
    d = (got.int() - want.int()).abs()
    assert d.max().item() <= 1 and (d > 0).float().mean().item() <= 1e-3
    assert st_a.tolist() == st_b.tolist()
    # and against the oracle's image of the same samples
    ref, _ = OracleScene(name, W / H).render(W, H, 5, spp, threads=8)
    d = np.abs(got.cpu().numpy().astype(int) - oracle_resolve(ref, spp).astype(int))
    assert d.max() <= 1 and (d > 0).mean() <= 2e-3
    # a second, larger frame regrows the work buffer; then the small one again
    got2 = ds.render_rgb8(2 * W, 2 * H, 0, 2, precision="f64")
    assert got2.shape == (2 * H, 2 * W, 3)
    again = ds.render_rgb8(W, H, 5, spp, precision="f64")
    assert (again.int() - got.int()).abs().max().item() <= 1
    with pytest.raises(rt.RtError):
        ds.render_rgb8(W, H, 0, 0)


# ---- f4 on the device: progressive passes, checkpoint, resume ------------------------------------------------
def test_progressive_checkpoint_resume_on_device(scenes, tmp_path):
    ds = scenes("cornell_smoke")
    W, H, SPP = 56, 40, 24
    whole = rt.render(ds, W, H, SPP, precision="f64")
    seen = []
    acc, done = rt.render_progressive(ds, W, H, SPP, 8, precision="f64", spp_per_launch=5, on_pass=lambda d, t: seen.append(d))
    assert done == SPP and seen == [8, 16, 24]
    assert torch.allclose(acc, whole, rtol=2e-6, atol=1e-6)
    # stop after 16 samples (2 passes), continue from the file in 3 passes of other sizes
    path = str(tmp_path / "smoke.rtck")
    part, d16 = rt.render_progressive(ds, W, H, 16, 8, precision="f64", checkpoint=path, scene_name="cornell_smoke")
    arr, meta = rt.load_checkpoint(path)
    assert d16 == 16 and meta["spp_done"] == 16 and meta["precision"] == "f64" and np.array_equal(arr, part.cpu().numpy())
    full, d24 = rt.render_progressive(ds, W, H, SPP, 3, precision="f64", checkpoint=path, resume=True, scene_name="cornell_smoke")
    assert d24 == SPP and rt.load_checkpoint(path)[1]["spp_done"] == SPP
    assert torch.allclose(full, whole, rtol=2e-6, atol=1e-6)
    # the resumed render's 8-bit image is the one-go render's
    assert (rt.resolve(full, SPP).int() - rt.resolve(whole, SPP).int()).abs().max().item() <= 1
    # a checkpoint of another precision / with more samples than asked for is refused
    with pytest.raises(ValueError):
        rt.render_progressive(ds, W, H, SPP, 3, precision="f32", checkpoint=path, resume=True, scene_name="cornell_smoke")
    with pytest.raises(ValueError):
        rt.render_progressive(ds, W, H, 8, 3, precision="f64", checkpoint=path, resume=True, scene_name="cornell_smoke")


# ---- ImageTexture through the in-repo JPEG decoder (src/texture.rs:68-74) ------------------------------------
def test_earth_frame_from_jpeg_equals_frame_from_its_raster(gpu, golden_dir, tmp_path):
    jpg = os.path.join(golden_dir, "earth_256x128.jpg")  # committed baseline JPEG (tools/make_small_jpeg.py)
    lib = C.CDLL(_ffi.LIB_HOST)
    lib.rth_image_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.rth_image_free.argtypes = [C.POINTER(C.c_uint8)]
    p, w, h = C.POINTER(C.c_uint8)(), C.c_uint32(), C.c_uint32()
    assert lib.rth_image_load(jpg.encode(), C.byref(p), C.byref(w), C.byref(h)) == 0
    raster = np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    lib.rth_image_free(p)
    assert raster.shape == (128, 256, 3) and raster.std() > 10
    ppm = str(tmp_path / "earth_small.ppm")
    with open(ppm, "wb") as f:
        f.write(b"P6\n256 128\n255\n" + raster.tobytes())
    W, H, spp = 96, 64, 8
    frames = []
    for path in (jpg, ppm):
        hs = rt.HostScene("earth", W / H, image_path=path)
        assert hs.desc.n_images == 1 and (hs.desc.images[0].width, hs.desc.images[0].height) == (256, 128)
        ds = rt.DeviceScene(hs, 0)
        acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
        ds.render_into(acc, 0, spp, precision="f64")
        frames.append(acc.cpu().numpy())
        ds.close(); hs.close()
    # same texels -> same frame, up to the order of the float additions of a launch that is cut into sample chunks
    assert np.allclose(frames[0], frames[1], rtol=1e-6, atol=0)
    # and it is the oracle's frame for that raster (the texture is really sampled: the frame differs from the default asset's)
    ref, _ = OracleScene("earth", W / H, image_path=ppm).render(W, H, 0, spp, threads=8)
    assert (np.abs(frames[0] - ref) <= 2e-6 * np.maximum(np.abs(ref), 1.0)).all(axis=2).mean() >= 0.9995
    ref_default, _ = OracleScene("earth", W / H).render(W, H, 0, spp, threads=8)
    assert np.abs(ref - ref_default).max() > 0.05


# ---- job-counter ring: more launches in flight than slots, on two streams ------------------------------------
def test_more_launches_in_flight_than_ring_slots(scenes):
    ds = scenes("random_scene", 1.5)  # persistent kernel: draws jobs from the ring's counters
    W, H, spp = 48, 32, 2
    n = 150  # > 2 x 64 slots
    want = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    ds.render_into(want, 0, spp, precision="f32")
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    bufs = [torch.zeros((n // 2 + 1, H, W, 3), dtype=torch.float32, device="cuda:0") for _ in range(2)]
    torch.cuda.synchronize()
    for i in range(n):
        k = i & 1
        with torch.cuda.stream(streams[k]):
            ds.render_into(bufs[k][i // 2], 0, spp, precision="f32")
    torch.cuda.synchronize()
    for k in range(2):
        for j in range(n // 2):
            assert torch.allclose(bufs[k][j], want, rtol=2e-6, atol=1e-6), (k, j)


# ---- BASELINE frame sizes: properties that need no oracle frame ----------------------------------------------
@pytest.mark.parametrize("name,W,H,spp", (("random_scene", 800, 450, 6), ("cornell_smoke", 1024, 1024, 6), ("final_scene", 2048, 2048, 2)))
def test_full_size_frames_properties(name, W, H, spp, scenes):
    # C1 800x450, C4 1024^2, C5 2048^2: job/grid arithmetic at the real sizes.  (1) the probe equals the oracle's on a strided
    # subset of rows (the whole grid for C1); (2) sample ranges are additive; (3) every sample counted once; (4) finite sums.
    aspect = W / H
    ds = scenes(name, aspect)
    oc = OracleScene(name, aspect)
    ids, ts = ds.trace_primary(W, H, 0.5, 0.4, "f64")
    if W * H <= 800 * 450:
        oi, ot = oc.trace_primary(W, H, 0.5, 0.4)
        assert np.array_equal(ids.cpu().numpy(), oi) and np.array_equal(ts.cpu().numpy(), ot)
    else:
        # the oracle's probe is a full-grid call: compare on the sub-grid of every s-th pixel, s a common divisor of W-1
        # and H-1.  u = (s k)/(W-1) and k/((W-1)/s) are the same rational from exactly representable operands, and IEEE
        # division rounds correctly: the two grids' rays are bit-identical, so ids and t must be too.
        s_ = next(k for k in (3, 23, 5, 7, 11, 31) if (W - 1) % k == 0 and (H - 1) % k == 0)
        w3, h3 = (W - 1) // s_ + 1, (H - 1) // s_ + 1
        oi, ot = oc.trace_primary(w3, h3, 0.5, 0.4)
        gi, gt = ids.cpu().numpy()[::s_, ::s_], ts.cpu().numpy()[::s_, ::s_]
        assert gi.shape == oi.shape and np.array_equal(gi, oi) and np.array_equal(gt, ot)
    a = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    b = torch.zeros_like(a)
    sa = torch.zeros(4, dtype=torch.int64, device="cuda:0")
    sb = torch.zeros_like(sa)
    ds.render_into(a, 0, spp, precision="f64", stats=sa)
    ds.render_into(b, 0, spp // 2, precision="f64", stats=sb)
    ds.render_into(b, spp // 2, spp - spp // 2, precision="f64", stats=sb)
    assert torch.allclose(a, b, rtol=2e-6, atol=2e-6)
    assert sa.tolist() == sb.tolist() and sa[0].item() + sa[3].item() == W * H * spp
    assert torch.isfinite(a).all()


def test_json_scene_boxes_on_the_device_path(gpu, golden_dir):
    # the SoA node boxes handed to rt_scene_create for the reference's JSON scenes equal the boxes stored in those files
    # (tests/test_json_boxes.py has the CPU side), and the device probe over exactly those arrays equals the fixture
    for fixture in ("json_scene10", "json_scene_500"):
        fx = np.load(os.path.join(golden_dir, fixture + ".npz"))
        host = rt.FlatHostScene(flat.FlatScene.from_dict(json.loads(str(fx["flat_scene"]))))
        d = host.desc
        n = d.bvh.n_nodes
        soa = np.stack([np.ctypeslib.as_array(getattr(d.bvh, k), shape=(n,)) for k in ("min_x", "min_y", "min_z", "max_x", "max_y", "max_z")], axis=1)
        assert np.abs(soa - fx["json_boxes"]).max() <= 1e-6
        ds = rt.DeviceScene(host, 0)
        ids, ts = ds.trace_primary(64, 64, 0.0, 0.5, "f64")
        assert np.array_equal(ids.cpu().numpy(), fx["ids"]) and np.array_equal(ts.cpu().numpy(), fx["t"])
        ds.close()


@pytest.mark.parametrize("seed", range(6))
def test_tie_scene_gpu_equals_oracle(seed, gpu):
    # tests/flat_fuzz.py: tie_scene — equal hit distances inside one BVH, shared cuboid faces, rays in coordinate planes: the cases
    # where the walk over the device's own tree must fall back to the reference-order walk (rt_device.hpp: walk_fast)
    from flat_fuzz import tie_scene
    host = rt.FlatHostScene(tie_scene(seed))
    oc = OracleScene.from_desc(host.desc_ptr)
    ds = rt.DeviceScene(host, 0)
    try:
        for (W, H) in ((41, 31), (96, 72)):
            ids, ts = ds.trace_primary(W, H, 0.25, 0.5, "f64")
            oi, ot = oc.trace_primary(W, H, 0.25, 0.5)
            assert np.array_equal(ids.cpu().numpy(), oi) and np.array_equal(ts.cpu().numpy(), ot)
        W, H, spp = 40, 30, 4
        ref, cnt = oc.render(W, H, 0, spp, threads=8)
        acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
        st = torch.zeros(4, dtype=torch.int64, device="cuda:0")
        ds.render_into(acc, 0, spp, precision="f64", stats=st)
        frame_close(f"{_where()}", acc.cpu().numpy(), ref, ds, oc, 0, spp)
        assert abs(int(st[1]) - cnt["casts"]) <= 0.005 * cnt["casts"] + 2
    finally:
        ds.close()


@pytest.mark.parametrize("precision", ("f64", "f32"))
def test_empty_world_on_the_device(precision, gpu):
    # an ObjectList without objects (src/objects.rs:38-48): no hit anywhere, every sample is the background; ragged frame
    from test_flat_scenes import empty_world_scene
    host = rt.FlatHostScene(empty_world_scene())
    ds = rt.DeviceScene(host, 0)
    try:
        ids, ts = ds.trace_primary(13, 9, 0.0, 0.5, precision)
        assert (ids.cpu().numpy() == -1).all()
        for no_sort in (False, True):
            acc = torch.zeros((9, 13, 3), dtype=torch.float32, device="cuda:0")
            st = torch.zeros(4, dtype=torch.int64, device="cuda:0")
            ds.render_into(acc, 0, 5, precision=precision, stats=st, no_sort=no_sort)
            assert st.cpu().tolist() == [13 * 9 * 5, 13 * 9 * 5, 0, 0]
            assert np.allclose(acc.cpu().numpy() / 5, [0.25, 0.5, 0.75], rtol=1e-6)
        rgb = ds.render_rgb8(13, 9, 0, 5, precision=precision).cpu().numpy()
        want = (np.clip(np.sqrt(np.array([0.25, 0.5, 0.75])), 0, 0.99999) * 256).astype(np.uint8)
        assert (rgb == want).all()
    finally:
        ds.close()


@pytest.mark.parametrize("no_sort", (False, True))
def test_nan_retry_cap_and_dropped_samples(no_sort, gpu):
    # ray_tracing.rs:102-110 retries a NaN sample for ever; the device gives up after max_nan_retries (stats[3], sample = 0).
    # final_scene produces NaN samples; with a cap of 1 some NaN twice in a row.  Counters and frame against the oracle run
    # with the same cap.
    W, H, spp = 48, 48, 24
    oc = OracleScene("final_scene", 1.0)
    ref, cnt = oc.render(W, H, 0, spp, threads=8, max_nan_retries=1)
    assert cnt["dropped"] > 0
    ds = rt.DeviceScene(rt.HostScene("final_scene", 1.0), 0)
    try:
        acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
        st = torch.zeros(4, dtype=torch.int64, device="cuda:0")
        ds.render_into(acc, 0, spp, precision="f64", stats=st, max_nan_retries=1, no_sort=no_sort)
        accepted, casts, retries, dropped = st.cpu().tolist()
        assert accepted + dropped == W * H * spp
        assert abs(dropped - cnt["dropped"]) <= 2 and abs(retries - cnt["nan_retries"]) <= 0.01 * cnt["nan_retries"] + 2
        assert abs(casts - cnt["casts"]) <= 0.005 * cnt["casts"]
        frame_close(f"nan_retry_cap[no_sort={no_sort}]", acc.cpu().numpy(), ref, ds, oc, 0, spp, max_nan_retries=1)
        st.zero_()
        ds.render_into(torch.zeros_like(acc), 0, spp, precision="f64", stats=st)  # default cap (64): nothing dropped here
        assert st.cpu().tolist()[3] == 0 and st.cpu().tolist()[0] == W * H * spp
    finally:
        ds.close()


@pytest.mark.parametrize("no_sort", (False, True))
def test_nan_scene_retries_and_drops_on_device(no_sort, gpu):
    # tests/flat_fuzz.py: nan_scene — every diffuse bounce is NaN: floor samples are retried up to the cap and dropped, in the
    # block-sorted kernel (light list, no BVH) and in the persistent one; counters and frame equal the oracle's with the same cap
    from flat_fuzz import nan_scene
    host = rt.FlatHostScene(nan_scene())
    oc = OracleScene.from_desc(host.desc_ptr)
    ds = rt.DeviceScene(host, 0)
    W, H, spp = 40, 40, 6
    try:
        for cap in (1, 3):
            ref, cnt = oc.render(W, H, 0, spp, threads=8, max_nan_retries=cap)
            acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
            st = torch.zeros(4, dtype=torch.int64, device="cuda:0")
            ds.render_into(acc, 0, spp, precision="f64", stats=st, max_nan_retries=cap, no_sort=no_sort)
            accepted, casts, retries, dropped = st.cpu().tolist()
            assert dropped == cnt["dropped"] > 1000 and retries == cnt["nan_retries"] == (cap + 1) * dropped
            assert accepted == W * H * spp - dropped and casts == cnt["casts"]
            assert np.allclose(acc.cpu().numpy(), ref, rtol=2e-6, atol=0)
    finally:
        ds.close()


def _png_pixels(path):
    import struct
    import zlib
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    at, idat, w, h = 8, b"", 0, 0
    while at < len(data):
        n, tag = struct.unpack(">I", data[at:at + 4])[0], data[at + 4:at + 8]
        body = data[at + 8:at + 8 + n]
        assert zlib.crc32(tag + body) & 0xFFFFFFFF == struct.unpack(">I", data[at + 8 + n:at + 12 + n])[0]
        if tag == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10])
            assert (depth, ctype) == (8, 2)
        elif tag == b"IDAT":
            idat += body
        at += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + w * 3)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 3)


@pytest.mark.parametrize("name,W,H,spp", (("cornell_box", 72, 56, 12), ("random_scene", 61, 35, 5), ("earth", 40, 30, 4)))
def test_native_launcher_equals_python_path(name, W, H, spp, scenes, tmp_path):
    # ray-tracing_amd/rt_render (host/rt_render_main.cpp): C++ against the two C headers only — no Python, no torch in that
    # process.  Same scene, seed and sample ranges -> the same 8-bit image as renderer.render + resolve + write_png.
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ray-tracing_amd", "rt_render")
    assert os.path.exists(exe), "build the launcher: python -c 'import ray_tracing_amd as rt; rt.build_all()'"
    out = str(tmp_path / "native.png")
    r = subprocess.run([exe, "--scene", name, "--width", str(W), "--height", str(H), "--spp", str(spp), "--spp-per-launch", "4", "--out", out],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["scene"] == name and rec["n_gpus"] == 1 and rec["dropped_samples"] == 0
    ds = scenes(name, W / H)
    st = torch.zeros(4, dtype=torch.int64, device="cuda:0")
    acc = rt.render(ds, W, H, spp, precision="f64", spp_per_launch=4, stats=st)
    want = rt.resolve(acc, spp).cpu().numpy()
    assert np.array_equal(_png_pixels(out), want)
    assert abs(rec["casts_per_sample"] - st[1].item() / st[0].item()) < 1e-3
    ref = str(tmp_path / "python.png")
    rt.write_png(ref, rt.resolve(acc, spp))
    assert np.array_equal(_png_pixels(ref), want)
    # errors are messages and exit codes
    r = subprocess.run([exe, "--scene", "no_such_scene", "--out", out], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "scene" in r.stderr


@pytest.mark.parametrize("seed", range(6))
def test_medium_scene_gpu_equals_oracle(seed, gpu):
    # tests/flat_fuzz.py: medium_scene — ConstantMedium around a sphere, a moving sphere, a cuboid, a single rectangle and a BVH,
    # some behind wrapper chains: both routes of the device through a medium (boundary_two / two searches through geom_hit)
    from flat_fuzz import medium_scene
    host = rt.FlatHostScene(medium_scene(seed))
    oc = OracleScene.from_desc(host.desc_ptr)
    ds = rt.DeviceScene(host, 0)
    try:
        W, H = 56, 40
        for tm, xi in ((0.0, 0.5), (0.7, 0.03)):
            ids, ts = ds.trace_primary(W, H, tm, xi, "f64")
            oi, ot = oc.trace_primary(W, H, tm, xi)
            assert np.array_equal(ids.cpu().numpy(), oi) and np.array_equal(ts.cpu().numpy(), ot)
        spp = 4
        ref, cnt = oc.render(W, H, 0, spp, threads=8)
        assert cnt["isotropic"] > 0
        for no_sort in (False, True):
            acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
            st = torch.zeros(4, dtype=torch.int64, device="cuda:0")
            ds.render_into(acc, 0, spp, precision="f64", stats=st, no_sort=no_sort)
            frame_close(f"{_where()}[no_sort={no_sort}]", acc.cpu().numpy(), ref, ds, oc, 0, spp)
            assert abs(int(st[1]) - cnt["casts"]) <= 0.005 * cnt["casts"] + 2
    finally:
        ds.close()


@pytest.mark.parametrize("seed", range(3))
def test_odd_lights_scene_gpu_equals_oracle(seed, gpu):
    # tests/flat_fuzz.py: odd_lights_scene — a light list with a BvhNode, a ConstantMedium and a MovingSphere (the Object trait's
    # defaults: pdf_value 0, random (1,0,0)) beside a rectangle and a translated sphere
    from flat_fuzz import odd_lights_scene
    host = rt.FlatHostScene(odd_lights_scene(seed))
    oc = OracleScene.from_desc(host.desc_ptr)
    ds = rt.DeviceScene(host, 0)
    try:
        W, H, spp = 48, 36, 6
        ref, cnt = oc.render(W, H, 0, spp, threads=8)
        acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
        st = torch.zeros(4, dtype=torch.int64, device="cuda:0")
        ds.render_into(acc, 0, spp, precision="f64", stats=st)
        frame_close(f"{_where()}", acc.cpu().numpy(), ref, ds, oc, 0, spp)
        assert abs(int(st[1]) - cnt["casts"]) <= 0.005 * cnt["casts"] + 2
        assert int(st[2]) == cnt["nan_retries"] or abs(int(st[2]) - cnt["nan_retries"]) <= 0.02 * cnt["nan_retries"] + 2
    finally:
        ds.close()


@pytest.mark.parametrize("seed,shutter", ((3, (-0.7, 2.3)), (17, (1.5, 4.0)), (21, (-3.0, -1.0))))
def test_shutter_intervals_other_than_0_1_on_device(seed, shutter, gpu):
    # ray times outside the (0, 1) the reference builds its BVH boxes for: moving spheres leave their boxes, the reference's loose
    # per-axis box test (bvh.rs:26-58) still reaches some of them — the walk may not skip nodes on geometric grounds there
    from flat_fuzz import random_scene
    fs = random_scene(seed)
    fs.cam = tuple(fs.cam[:-2]) + (float(shutter[0]), float(shutter[1]))
    host = rt.FlatHostScene(fs)
    oc = OracleScene.from_desc(host.desc_ptr)
    ds = rt.DeviceScene(host, 0)
    try:
        W, H = 56, 40
        for tm in (shutter[0], 0.5 * (shutter[0] + shutter[1]), shutter[1]):
            ids, ts = ds.trace_primary(W, H, tm, 0.4, "f64")
            oi, ot = oc.trace_primary(W, H, tm, 0.4)
            assert np.array_equal(ids.cpu().numpy(), oi) and np.array_equal(ts.cpu().numpy(), ot), (seed, tm)
        spp = 4
        ref, cnt = oc.render(W, H, 0, spp, threads=8)
        acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
        st = torch.zeros(4, dtype=torch.int64, device="cuda:0")
        ds.render_into(acc, 0, spp, precision="f64", stats=st)
        frame_close(f"{_where()}", acc.cpu().numpy(), ref, ds, oc, 0, spp)
        assert abs(int(st[1]) - cnt["casts"]) <= 0.005 * cnt["casts"] + 2
    finally:
        ds.close()


@pytest.mark.parametrize("seed", (0, 3, 5, 9, 12, 14))
def test_bvh_boxes_that_do_not_fit_their_contents_on_device(seed, gpu):
    # node boxes that are too small, shifted or not nested (tests/test_flat_scenes.py has the CPU twin and the reasons): the device
    # follows the boxes it is given exactly as the reference's walk would
    import random as _random
    from flat_fuzz import random_scene
    rng = _random.Random(900 + seed)
    fs = next(c for c in (random_scene(s0) for s0 in range(seed * 5, seed * 5 + 40)) if len(c.nodes) >= 3)
    for _ in range(rng.randint(1, 3)):
        box = fs.nodes[rng.randrange(len(fs.nodes))][0]
        mode = rng.randrange(4)
        for k in range(3):
            w = box[3 + k] - box[k]
            if mode == 0:
                box[k] += 0.3 * w; box[3 + k] -= 0.3 * w
            elif mode == 1:
                box[k] += 0.4 * w; box[3 + k] += 0.4 * w
            elif mode == 2:
                box[k] -= 2.0 * w; box[3 + k] += 2.0 * w
            else:
                box[3 + k] -= 0.5 * w * (k == seed % 3)
    host = rt.FlatHostScene(fs)
    oc = OracleScene.from_desc(host.desc_ptr)
    ds = rt.DeviceScene(host, 0)
    try:
        W, H = 64, 48
        for tm in (0.0, 0.55):
            ids, ts = ds.trace_primary(W, H, tm, 0.5, "f64")
            oi, ot = oc.trace_primary(W, H, tm, 0.5)
            assert np.array_equal(ids.cpu().numpy(), oi) and np.array_equal(ts.cpu().numpy(), ot), (seed, tm)
        spp = 4
        ref, cnt = oc.render(W, H, 0, spp, threads=8)
        acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
        st = torch.zeros(4, dtype=torch.int64, device="cuda:0")
        ds.render_into(acc, 0, spp, precision="f64", stats=st)
        frame_close(f"{_where()}", acc.cpu().numpy(), ref, ds, oc, 0, spp)
        assert abs(int(st[1]) - cnt["casts"]) <= 0.005 * cnt["casts"] + 2
    finally:
        ds.close()


def _gpu_same_as_oracle(fs, W=56, H=40, spp=4, times=((0.0, 0.5), (0.6, 0.1))):
    host = rt.FlatHostScene(fs)
    oc = OracleScene.from_desc(host.desc_ptr)
    ds = rt.DeviceScene(host, 0)
    try:
        for tm, xi in times:
            ids, ts = ds.trace_primary(W, H, tm, xi, "f64")
            oi, ot = oc.trace_primary(W, H, tm, xi)
            assert np.array_equal(ids.cpu().numpy(), oi) and np.array_equal(ts.cpu().numpy(), ot), tm
        ref, cnt = oc.render(W, H, 0, spp, threads=8)
        acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
        st = torch.zeros(4, dtype=torch.int64, device="cuda:0")
        ds.render_into(acc, 0, spp, precision="f64", stats=st)
        frame_close(f"{_where()}", acc.cpu().numpy(), ref, ds, oc, 0, spp)
        assert abs(int(st[1]) - cnt["casts"]) <= 0.005 * cnt["casts"] + 2
    finally:
        ds.close()


@pytest.mark.parametrize("seed", (0, 1, 3, 6))
def test_far_scene_gpu_equals_oracle(seed, gpu):
    # tests/flat_fuzz.py: far_scene — coordinates ~1e6 (and 4e9), features ~1: the f32 node classification decides almost nothing
    from flat_fuzz import far_scene
    _gpu_same_as_oracle(far_scene(seed))
    if seed == 0:
        _gpu_same_as_oracle(far_scene(seed, offset=(-4.0e9, 1.0e9, 3.0e9)), times=((0.3, 0.5),))


@pytest.mark.parametrize("seed", (0, 3))
def test_pile_scene_gpu_equals_oracle(seed, gpu):
    # tests/flat_fuzz.py: pile_scene — dozens of primitives in the same place inside one BVH (coinciding centroids, exact ties)
    from flat_fuzz import pile_scene
    _gpu_same_as_oracle(pile_scene(seed), W=41, H=31, times=((0.0, 0.5),))


@pytest.mark.parametrize("seed", (0, 1, 2, 5, 8))
def test_texture_scene_gpu_equals_oracle(seed, gpu):
    # tests/flat_fuzz.py: texture_scene — noise (1-3 Perlin tables, scales 1e-3 .. 250 and negative) and image textures (1x1 .. 64x33)
    # on every primitive kind, as emitters, as a medium's phase texture and inside nested checkers
    from flat_fuzz import texture_scene
    _gpu_same_as_oracle(texture_scene(seed), times=((0.0, 0.5),), spp=6)


@pytest.mark.parametrize("seed", (0, 3, 5, 9))
def test_moving_spheres_with_other_time_bases_on_device(seed, gpu):
    from flat_fuzz import random_scene
    fs = random_scene(seed)
    n = 0
    for i, (kind, m, d) in enumerate(fs.prims):
        if kind == 1:
            d = list(d)
            d[6], d[7] = ((-1.0, 3.0), (0.5, 0.75), (2.0, 1.0))[n % 3]
            fs.prims[i] = (kind, m, d)
            n += 1
    assert n > 0
    _gpu_same_as_oracle(fs, times=((0.0, 0.5), (0.5, 0.3), (1.0, 0.7)))


@pytest.mark.parametrize("seed", (0, 1, 3, 4, 7, 10, 13))
def test_extreme_scene_gpu_equals_oracle(seed, gpu):
    # tests/flat_fuzz.py: extreme_scene — negative / zero radii, degenerate rectangles and cuboids, material parameters far outside the
    # reference's scenes: NaN / inf have to flow through the device as through the oracle (and NaN samples are retried alike)
    from flat_fuzz import extreme_scene
    _gpu_same_as_oracle(extreme_scene(seed), spp=6)


@pytest.mark.parametrize("seed", (0, 1, 4, 7))
def test_instancing_scene_gpu_equals_oracle(seed, gpu):
    # tests/flat_fuzz.py: instancing_scene — one BVH behind several instances (plain, Translate + RotateY, as a medium's boundary),
    # rectangle and cuboid lights behind rotating chains
    from flat_fuzz import instancing_scene
    _gpu_same_as_oracle(instancing_scene(seed), spp=6)


@pytest.mark.parametrize("name", ("cornell_box", "random_scene", "cornell_smoke"))
@pytest.mark.parametrize("W,H,spp,s0", ((2, 2, 5, 0), (3, 700, 2, 7), (2051, 3, 2, 0), (500, 2, 3, 4294967000), (9, 17, 257, 1), (64, 64, 1, 123456)))
def test_odd_frame_shapes_and_sample_ranges(name, W, H, spp, s0, scenes):
    # the launch arithmetic (tiles, job lists, sample chunks, the grid's z dimension, atomics vs plain adds) at shapes far from the
    # benchmark's: slivers, a frame smaller than one tile, sample indices next to 2^32, one sample, a prime number of samples
    ds = scenes(name, W / H)
    oc = OracleScene(name, W / H)
    ref, cnt = oc.render(W, H, s0, spp, threads=8)
    for no_sort in (False, True):
        acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
        st = torch.zeros(4, dtype=torch.int64, device="cuda:0")
        ds.render_into(acc, s0, spp, precision="f64", stats=st, no_sort=no_sort)
        frame_close(f"odd_shapes[{name},{W}x{H},{spp},{s0},no_sort={no_sort}]", acc.cpu().numpy(), ref, ds, oc, s0, spp)
        assert int(st[0]) == W * H * spp and abs(int(st[1]) - cnt["casts"]) <= 0.005 * cnt["casts"] + 2


# ---- jobs that stream into each other (both render kernels): many jobs per resident wave / block ----------------------------
@pytest.mark.parametrize("name,W,H,spp", (("cornell_box", 1024, 1024, 1), ("cornell_box", 1024, 1024, 7), ("cornell_smoke", 1100, 700, 3),
                                          ("cornell_box", 1536, 1024, 2), ("cornell_smoke", 1024, 1024, 5)))
def test_sorted_kernel_with_many_short_jobs_per_block(name, W, H, spp, scenes):
    # more tiles than resident blocks and job lists of 1-7 entries per lane: every block goes through many jobs, the next one's paths
    # start while the last one's still run (two pixel-sum buffers), job switches wait for the buffer of the job before.  A block that
    # left while the counter still had jobs (found while this was built: 27 % of a 1024^2 @ 7 spp frame missing) shows here.
    ds = scenes(name, W / H)
    a = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    b = torch.zeros_like(a)
    sa = torch.zeros(4, dtype=torch.int64, device="cuda:0")
    sb = torch.zeros_like(sa)
    ds.render_into(a, 5, spp, precision="f64", stats=sa)
    ds.render_into(b, 5, spp, precision="f64", stats=sb, no_sort=True)
    assert sa.tolist() == sb.tolist() and sa[0].item() == W * H * spp and sa[3].item() == 0
    assert torch.allclose(a, b, rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("name,W,H,spp", (("random_scene", 800, 450, 100), ("final_scene", 512, 512, 24), ("two_perlin_spheres", 640, 360, 40)))
def test_wave_kernel_streaming_jobs_add_up(name, W, H, spp, scenes):
    # launches cut into sample chunks: a wave takes the next job while paths of the last one still run and those add straight to the frame.
    # One launch == two launches of parts of the samples (other chunking, other job streams), and every sample is counted once.
    ds = scenes(name, W / H)
    a = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    b = torch.zeros_like(a)
    sa = torch.zeros(4, dtype=torch.int64, device="cuda:0")
    sb = torch.zeros_like(sa)
    ds.render_into(a, 0, spp, precision="f64", stats=sa)
    k = spp // 3
    ds.render_into(b, 0, k, precision="f64", stats=sb)
    ds.render_into(b, k, spp - k, precision="f64", stats=sb)
    assert sa.tolist() == sb.tolist() and sa[0].item() + sa[3].item() == W * H * spp
    assert torch.allclose(a, b, rtol=4e-6, atol=2e-6)
