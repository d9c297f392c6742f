"""rt_render_rgb8_noalloc (include/rt_hip.h): the allocation-free variant of rt_render_rgb8 — float sums and job counter
in the library's static device array — against rt_render_rgb8, through the C-ABI, two calls in a row, the second with
a smaller frame (the array keeps the first call's contents)."""
import ctypes as C

import pytest
import torch

import ray_tracing_amd as rt
from ray_tracing_amd import _ffi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ("cornell_box", "random_scene"))
def test_noalloc_equals_rgb8_two_calls_second_smaller(name, gpu):
    hs = rt.HostScene(name, 1.5)
    ds = rt.DeviceScene(hs, 0)
    try:
        # spp in [32, 64): both render kernels run one sample chunk per tile, so the sums are order-independent and
        # the two entry points must agree BIT FOR BIT
        for (W, H, spp, s0) in ((384, 256, 33, 0), (120, 80, 40, 7)):
            st_a = torch.zeros(4, dtype=torch.int64, device="cuda:0")
            st_b = torch.zeros_like(st_a)
            want = ds.render_rgb8(W, H, s0, spp, precision="f64", stats=st_a)
            out = torch.full((H, W, 3), 77, dtype=torch.uint8, device="cuda:0")
            got = ds.render_rgb8(W, H, s0, spp, precision="f64", stats=st_b, noalloc=True, out=out)
            torch.cuda.synchronize()
            assert torch.equal(got, want), (name, W, H)
            assert st_a.tolist() == st_b.tolist() and st_a[0].item() == W * H * spp
            assert got.float().std().item() > 1.0  # an image, not a constant
    finally:
        ds.close()
        hs.close()


def test_noalloc_rejects_frames_beyond_the_scratch(gpu):
    hs = rt.HostScene("cornell_box", 1.0)
    ds = rt.DeviceScene(hs, 0)
    try:
        out = torch.zeros((8, 8, 3), dtype=torch.uint8, device="cuda:0")
        p = _ffi.RtRenderParams(1025, 1024, 0, 1, 50, 64, 1, _ffi.RT_F32, 0)  # 1 row of pixels more than RT_NOALLOC_MAX_PIXELS
        rc = _ffi.hip().rt_render_rgb8_noalloc(ds._h, C.byref(p), C.c_void_p(out.data_ptr()), None, None)
        assert rc == 1 and b"RT_NOALLOC_MAX_PIXELS" in _ffi.hip().rt_last_error()
        # the largest admissible frame works (1024 x 1024, 1 sample)
        big = ds.render_rgb8(1024, 1024, 0, 1, precision="f32", noalloc=True)
        want = ds.render_rgb8(1024, 1024, 0, 1, precision="f32")
        assert torch.equal(big, want)
    finally:
        ds.close()
        hs.close()
