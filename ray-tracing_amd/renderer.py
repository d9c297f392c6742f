"""Host-side driver of the hot path: build a reference scene, flatten it, upload it, launch the
render kernels, (all-reduce,) resolve.  Replaces `run_ray_tracing` of the reference
(src/ray_tracing.rs:65-135: thread pool + mpsc + PNG) — the per-pixel work itself
(src/ray_tracing.rs:95-118) runs in librt_hip.so.  PyTorch is used for device memory, streams and
torch.distributed only."""
import ctypes as C
import json
import os
import struct
import zlib

import numpy as np
import torch

from . import _ffi
from ._ffi import RT_F32, RT_F64, RtRenderParams, check

ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")
EARTHMAP = os.path.join(ASSETS, "earthmap.ppm")
SCENES = ("random_scene", "random_scene_light", "two_checker_spheres", "two_perlin_spheres", "earth", "simple_light",
          "cornell_box", "cornell_smoke", "final_scene")
DEFAULT_SEED = 0x5EED0001
_PRECISION = {"f32": RT_F32, "f64": RT_F64, RT_F32: RT_F32, RT_F64: RT_F64}

_inited = set()


def _flags(no_sort=False, deterministic=False):
    return (_ffi.RT_RENDER_NO_SORT if no_sort else 0) | (_ffi.RT_RENDER_DETERMINISTIC if deterministic else 0)


def init(device=0):
    """rt_init: select the GPU for this process (one process per GPU)."""
    check(_ffi.hip().rt_init(int(device)))
    _inited.add(int(device))
    return _ffi.hip().rt_describe().decode()


class HostScene:
    """A scene graph built by the host library from one of the reference's constructors
    (src/scenes.rs) plus its flattened RtSceneDesc.  Needs no GPU."""

    def __init__(self, name, aspect_ratio=1.0, seed=DEFAULT_SEED, image_path=None):
        lib = _ffi.host()
        self.name, self.aspect_ratio, self.seed = name, float(aspect_ratio), int(seed)
        img = (image_path or EARTHMAP).encode()
        self._h = lib.rth_scene_build(name.encode(), float(aspect_ratio), int(seed), img)
        if not self._h:
            raise _ffi.RtError(f"rth_scene_build({name}): {lib.rth_last_error().decode()}")

    @property
    def desc_ptr(self):
        return _ffi.host().rth_scene_desc(self._h)

    @property
    def desc(self):
        return C.cast(self.desc_ptr, C.POINTER(_ffi.RtSceneDesc)).contents

    @property
    def num_world_prims(self):
        return _ffi.host().rth_scene_num_world_prims(self._h)

    def roundtrip_check(self):
        n = _ffi.host().rth_scene_roundtrip_check(self._h)
        return n, _ffi.host().rth_last_error().decode() if n else ""

    def close(self):
        if self._h:
            _ffi.host().rth_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FlatHostScene:
    """A flat scene built in Python (flat.FlatScene / flat.scene_from_json) with HostScene's interface."""

    def __init__(self, flat_scene, name="flat"):
        self.name, self.flat = name, flat_scene
        self._desc = flat_scene.desc()

    @property
    def desc(self):
        return self._desc

    @property
    def desc_ptr(self):
        return C.cast(C.pointer(self._desc), C.c_void_p)

    def close(self):
        pass


class DeviceScene:
    """rt_scene_create on the current device."""

    def __init__(self, host_scene, device=None):
        self.device = torch.cuda.current_device() if device is None else int(device)
        if self.device not in _inited:
            init(self.device)
        self.host = host_scene
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(_ffi.hip().rt_scene_create(host_scene.desc_ptr, C.byref(h)))
        self._h = h

    @property
    def torch_device(self):
        return torch.device(f"cuda:{self.device}")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def trace_primary(self, width, height, time=0.0, medium_xi=0.5, precision="f64"):
        """Deterministic parity probe -> (ids int32 [H,W], t float64 [H,W]); row index = y counted upward."""
        with torch.cuda.device(self.device):
            ids = torch.empty((height, width), dtype=torch.int32, device="cuda")
            ts = torch.empty((height, width), dtype=torch.float64, device="cuda")
            check(_ffi.hip().rt_trace_primary(self._h, width, height, float(time), float(medium_xi), _PRECISION[precision],
                                              C.c_void_p(ids.data_ptr()), C.c_void_p(ts.data_ptr()), self._stream()))
        return ids, ts

    def render_into(self, accum, spp_begin, spp_count, seed=DEFAULT_SEED, max_depth=50, precision="f32", stats=None,
                    max_nan_retries=64, no_sort=False, deterministic=False):
        """accum: float32 CUDA tensor [H, W, 3] (row = y upward), ADDED to. stats: uint64-as-int64 tensor [4] or None.
        deterministic: RT_RENDER_DETERMINISTIC (one lane per pixel, samples summed in index order: reproducible bits)."""
        assert accum.is_cuda and accum.dtype == torch.float32 and accum.is_contiguous() and accum.dim() == 3 and accum.shape[2] == 3
        h, w = accum.shape[0], accum.shape[1]
        p = RtRenderParams(w, h, int(spp_begin), int(spp_count), int(max_depth), int(max_nan_retries), int(seed),
                           _PRECISION[precision], _flags(no_sort, deterministic))
        sp = C.c_void_p(stats.data_ptr()) if stats is not None else C.c_void_p()
        with torch.cuda.device(self.device):
            check(_ffi.hip().rt_render(self._h, C.byref(p), C.c_void_p(accum.data_ptr()), sp, self._stream()))

    def render_rgb8(self, width, height, spp_begin, spp_count, seed=DEFAULT_SEED, max_depth=50, precision="f32", stats=None,
                    max_nan_retries=64, no_sort=False, noalloc=False, out=None, deterministic=False):
        """ray_tracing.rs:95-118 in one call -> uint8 CUDA tensor [H, W, 3], top row first (rt_render_rgb8, or the
        allocation-free rt_render_rgb8_noalloc with noalloc=True; `out` = a preallocated result tensor)."""
        p = RtRenderParams(int(width), int(height), int(spp_begin), int(spp_count), int(max_depth), int(max_nan_retries), int(seed),
                           _PRECISION[precision], _flags(no_sort, deterministic))
        sp = C.c_void_p(stats.data_ptr()) if stats is not None else C.c_void_p()
        with torch.cuda.device(self.device):
            rgb = out if out is not None else torch.empty((height, width, 3), dtype=torch.uint8, device="cuda")
            assert rgb.is_cuda and rgb.dtype == torch.uint8 and rgb.is_contiguous() and rgb.numel() == width * height * 3
            fn = _ffi.hip().rt_render_rgb8_noalloc if noalloc else _ffi.hip().rt_render_rgb8
            check(fn(self._h, C.byref(p), C.c_void_p(rgb.data_ptr()), sp, self._stream()))
        return rgb

    def sample_pixel(self, width, height, x, y, spp_begin, spp_count, seed=DEFAULT_SEED, max_depth=50, precision="f64"):
        p = RtRenderParams(width, height, int(spp_begin), int(spp_count), int(max_depth), 64, int(seed), _PRECISION[precision], 0)
        with torch.cuda.device(self.device):
            out = torch.zeros((spp_count, 3), dtype=torch.float64, device="cuda")
            check(_ffi.hip().rt_debug_sample_pixel(self._h, C.byref(p), int(x), int(y), C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def close(self):
        if getattr(self, "_h", None):
            _ffi.hip().rt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def resolve(accum, spp):
    """ray_tracing.rs:112-117,126-128 on the device: -> uint8 CUDA tensor [H, W, 3], top row first."""
    h, w = accum.shape[0], accum.shape[1]
    rgb = torch.empty((h, w, 3), dtype=torch.uint8, device=accum.device)
    stream = C.c_void_p(torch.cuda.current_stream(accum.device).cuda_stream)
    with torch.cuda.device(accum.device):
        check(_ffi.hip().rt_resolve(C.c_void_p(accum.data_ptr()), w, h, int(spp), C.c_void_p(rgb.data_ptr()), stream))
    return rgb


def comm_probe():
    """Raises unless this process can bind RCCL through the library (rt_exchange.hip loads it at first use).  The id the
    call makes is dropped: an id only becomes a communicator in rt_comm_create."""
    buf = (C.c_uint8 * _ffi.RT_COMM_ID_BYTES)()
    check(_ffi.hip().rt_comm_unique_id(buf))


class Comm:
    """The exchange step through the C-ABI (rt_comm_* / rt_allreduce_*: one RCCL all-reduce of the accumulation
    buffers).  `broadcast_bytes(b: bytes or None) -> bytes` ships rank 0's 128-byte id to every rank (any
    transport: torch.distributed, a file, MPI); it is not needed for a single rank."""

    def __init__(self, rank=0, world_size=1, broadcast_bytes=None, device=None):
        self.device = torch.cuda.current_device() if device is None else int(device)
        if self.device not in _inited:
            init(self.device)
        self.rank, self.world_size = int(rank), int(world_size)
        buf = (C.c_uint8 * _ffi.RT_COMM_ID_BYTES)()
        if self.rank == 0:
            check(_ffi.hip().rt_comm_unique_id(buf))
        if self.world_size > 1:
            if broadcast_bytes is None:
                raise ValueError("world_size > 1 needs broadcast_bytes to ship the communicator id")
            raw = broadcast_bytes(bytes(buf) if self.rank == 0 else None)
            buf = (C.c_uint8 * _ffi.RT_COMM_ID_BYTES).from_buffer_copy(raw)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(_ffi.hip().rt_comm_create(buf, self.world_size, self.rank, C.byref(h)))
        self._h = h

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def all_reduce(self, accum):
        """In-place sum over ranks of a float32 CUDA tensor (the accumulation buffer), on the current stream."""
        assert accum.is_cuda and accum.dtype == torch.float32 and accum.is_contiguous()
        with torch.cuda.device(self.device):
            check(_ffi.hip().rt_allreduce_accum(self._h, C.c_void_p(accum.data_ptr()), accum.numel(), self._stream()))
        return accum

    def all_reduce_stats(self, stats):
        assert stats.is_cuda and stats.dtype == torch.int64 and stats.numel() == _ffi.RT_STATS_WORDS
        with torch.cuda.device(self.device):
            check(_ffi.hip().rt_allreduce_stats(self._h, C.c_void_p(stats.data_ptr()), self._stream()))
        return stats

    def close(self):
        if getattr(self, "_h", None):
            torch.cuda.synchronize(self.device)
            _ffi.hip().rt_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Exchange:
    """The exchange step of a multi-process render, as every launcher of this package uses it (`python -m ray_tracing_amd`,
    bench.py): the C-ABI route — rt_comm_* + rt_allreduce_accum, ONE RCCL all-reduce of the float sums — when EVERY rank can
    take it, else torch.distributed for all of them.  Ranks must agree on the route or the collectives would hang, so the
    choice itself is two small torch.distributed reductions: (1) can this rank bind RCCL through the library, (2) did the
    communicator come up and does a known all-reduce through it sum right.  torch.distributed otherwise only ships the
    128-byte communicator id.  `dist`: the initialised torch.distributed module; `device`: where the control tensors live
    (the rank's GPU; "cpu" under gloo, where step 1 or 2 fails and the fallback is what runs)."""

    def __init__(self, rank, world_size, dist, device, prefer="rt"):
        self.rank, self.world_size, self.dist = int(rank), int(world_size), dist
        self.device = torch.device(device)
        self.comm, self.route, self.note = None, "torch", None
        if prefer != "rt":
            return
        why = ""
        try:
            if self.device.type != "cuda":
                raise _ffi.RtError("no GPU on this rank")
            comm_probe()
            ok = True
        except Exception as e:  # noqa: BLE001
            ok, why = False, f"{type(e).__name__}: {e}"
        if self._all_agree(ok):
            try:
                self.comm = Comm(self.rank, self.world_size, self._bcast, self.device.index)
                probe = torch.ones(1024, dtype=torch.float32, device=self.device)
                self.comm.all_reduce(probe)
                torch.cuda.synchronize(self.device)
                ok = bool((probe == float(self.world_size)).all().item())
                why = "" if ok else "probe all-reduce gave a wrong sum"
            except Exception as e:  # noqa: BLE001
                ok, why = False, f"{type(e).__name__}: {e}"
            ok = self._all_agree(ok)
        if ok:
            self.route = "rt"
        else:
            if self.comm is not None:
                self.comm.close()
                self.comm = None
            self.note = "rt exchange unavailable on some rank, torch.distributed used" + (f" (this rank: {why})" if why else "")

    def _bcast(self, b):
        t = torch.zeros(_ffi.RT_COMM_ID_BYTES, dtype=torch.uint8, device=self.device)
        if b is not None:
            t.copy_(torch.frombuffer(bytearray(b), dtype=torch.uint8))
        self.dist.broadcast(t, 0)
        return bytes(t.cpu().numpy().tobytes())

    def _all_agree(self, ok):
        f = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device)
        self.dist.all_reduce(f, op=self.dist.ReduceOp.MIN)
        return bool(f.item())

    def all_reduce(self, buf):
        """In-place sum over the ranks of a float32 buffer (the accumulation buffer)."""
        if self.comm is not None:
            self.comm.all_reduce(buf)
        else:
            self.dist.all_reduce(buf)
        return buf

    def all_reduce_stats(self, stats):
        if self.comm is not None:
            self.comm.all_reduce_stats(stats)
        else:
            self.dist.all_reduce(stats)
        return stats

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None


def spp_shard(spp, rank, world_size):
    """GPU g of N renders samples [g*spp/N, (g+1)*spp/N) of every pixel (disjoint RNG keys)."""
    return spp * rank // world_size, spp * (rank + 1) // world_size - spp * rank // world_size


def render(scene, width, height, spp, seed=DEFAULT_SEED, max_depth=50, precision="f32", spp_per_launch=None,
           rank=0, world_size=1, all_reduce=None, stats=None):
    """Render this rank's spp shard into a fresh accumulation buffer; if all_reduce is given (Exchange.all_reduce: the
    C-ABI's rt_allreduce_accum; any callable that sums a float32 tensor in place over the ranks) sum the buffers over
    ranks.  Returns accum [H,W,3] f32."""
    accum = torch.zeros((height, width, 3), dtype=torch.float32, device=scene.torch_device)
    begin, count = spp_shard(spp, rank, world_size)
    step = spp_per_launch or count or 1
    s = begin
    while s < begin + count:
        n = min(step, begin + count - s)
        scene.render_into(accum, s, n, seed=seed, max_depth=max_depth, precision=precision, stats=stats)
        s += n
    if all_reduce is not None and world_size > 1:
        all_reduce(accum)
    return accum


# ---- progressive rendering + checkpoints of the accumulation buffer -------------------------
# The reference renders one image in one go and keeps nothing (src/ray_tracing.rs:95-133).  Here samples are
# keyed by (seed, pixel, sample index), so a render is a SUM over sample ranges: it can be done in passes,
# the linear float sums written out between passes, and continued later — by any number of GPUs — to the
# same image an uninterrupted render gives.
CKPT_MAGIC = b"RTACCUM1"


def save_checkpoint(path, accum, meta):
    """accum: [H, W, 3] f32 linear radiance SUMS of samples [0, meta['spp_done']).  Written atomically."""
    arr = accum.detach().cpu().numpy() if hasattr(accum, "detach") else accum
    arr = np.ascontiguousarray(arr, dtype=np.float32)
    if arr.ndim != 3 or arr.shape[2] != 3:
        raise ValueError("accum must be [H, W, 3]")
    head = dict(meta)
    head.update(height=int(arr.shape[0]), width=int(arr.shape[1]))
    blob = json.dumps(head, sort_keys=True).encode()
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(CKPT_MAGIC + struct.pack("<II", len(blob), zlib.crc32(arr.tobytes()) & 0xFFFFFFFF) + blob + arr.tobytes())
    os.replace(tmp, path)


def load_checkpoint(path):
    """-> (numpy [H, W, 3] f32, meta dict).  Raises ValueError on a damaged or foreign file."""
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:8] != CKPT_MAGIC or len(raw) < 16:
        raise ValueError(f"{path}: not an accumulation-buffer checkpoint")
    n, crc = struct.unpack("<II", raw[8:16])
    meta = json.loads(raw[16:16 + n].decode())
    body = raw[16 + n:]
    h, w = int(meta["height"]), int(meta["width"])
    if len(body) != h * w * 12 or (zlib.crc32(body) & 0xFFFFFFFF) != crc:
        raise ValueError(f"{path}: truncated or corrupted checkpoint")
    return np.frombuffer(body, dtype=np.float32).reshape(h, w, 3).copy(), meta


def render_progressive(scene, width, height, spp, pass_spp, seed=DEFAULT_SEED, max_depth=50, precision="f32", spp_per_launch=None,
                       rank=0, world_size=1, all_reduce=None, stats=None, checkpoint=None, resume=False, scene_name="", on_pass=None):
    """Render samples [0, spp) in passes of pass_spp samples; each pass is sharded over the ranks like render().
    After every pass the summed buffer is written to `checkpoint` (rank 0); with resume=True an existing
    checkpoint of the same frame is continued from its spp_done.  on_pass(spp_done, summed accum or None) is
    called on every rank after each pass.  Returns (accum summed over ranks, spp rendered in total)."""
    accum = torch.zeros((height, width, 3), dtype=torch.float32, device=scene.torch_device)
    ident = {"scene": scene_name, "seed": int(seed), "max_depth": int(max_depth), "width": int(width), "height": int(height)}
    done = 0
    if resume and checkpoint and os.path.exists(checkpoint):
        arr, meta = load_checkpoint(checkpoint)
        for k, v in ident.items():
            if meta.get(k) != v:
                raise ValueError(f"checkpoint {checkpoint} is of another render: {k} = {meta.get(k)!r}, wanted {v!r}")
        if meta.get("precision", precision) != precision:  # f32 and f64 samples are different estimates: do not mix them silently
            raise ValueError(f"checkpoint {checkpoint} was rendered in {meta.get('precision')}, this run asks for {precision}")
        done = int(meta["spp_done"])
        if done < 0 or done > spp:
            raise ValueError(f"checkpoint {checkpoint} holds {done} samples per pixel, more than the {spp} asked for")
        if rank == 0:  # the sums so far enter the all-reduce once
            accum += torch.from_numpy(arr).to(accum.device)
    if all_reduce is not None and world_size > 1:
        # every rank must start from the same sample count (a checkpoint file that only some ranks can see would make
        # the passes, and with them the all-reduce calls, differ).  The callback sums FLOAT32 buffers (rt_allreduce_accum has no
        # other type), so the count travels as its four bytes b and their squares, all exact in f32 up to 64 ranks:
        # the ranks agree on a byte <=> N * sum(b^2) == sum(b)^2
        by = [float((done >> (8 * k)) & 0xFF) for k in range(4)]
        chk = torch.tensor(by + [b * b for b in by], dtype=torch.float32, device=accum.device)
        all_reduce(chk)
        c = chk.tolist()
        if any(abs(world_size * c[4 + k] - c[k] ** 2) > 0.5 for k in range(4)):
            raise ValueError(f"ranks disagree on the checkpoint's spp_done (this rank: {done}): put {checkpoint} where every rank reads the same file")
    while done < spp:
        n = min(int(pass_spp), spp - done)
        begin, count = spp_shard(n, rank, world_size)
        step = spp_per_launch or count or 1
        s = done + begin
        while s < done + begin + count:
            k = min(step, done + begin + count - s)
            scene.render_into(accum, s, k, seed=seed, max_depth=max_depth, precision=precision, stats=stats)
            s += k
        done += n
        total = None
        if (checkpoint or on_pass) and done < spp:
            total = accum.clone()
            if all_reduce is not None and world_size > 1:
                all_reduce(total)
            if checkpoint and rank == 0:
                save_checkpoint(checkpoint, total, dict(ident, spp_done=done, precision=precision))
        if on_pass:
            on_pass(done, total)
    if all_reduce is not None and world_size > 1:
        all_reduce(accum)
    if checkpoint and rank == 0:
        save_checkpoint(checkpoint, accum, dict(ident, spp_done=done, precision=precision))
    return accum, done


def write_png(path, rgb):
    """Minimal PNG writer (8-bit RGB, zlib) for the resolved image: [H, W, 3] uint8, top row first."""
    rgb = rgb.cpu().contiguous() if hasattr(rgb, "cpu") else rgb
    h, w = int(rgb.shape[0]), int(rgb.shape[1])
    raw = bytes(rgb.numpy().tobytes()) if hasattr(rgb, "numpy") else bytes(rgb.tobytes())
    rows = b"".join(b"\x00" + raw[y * w * 3:(y + 1) * w * 3] for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(rows, 6)) + chunk(b"IEND", b""))
