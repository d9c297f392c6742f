"""ray-tracing_amd — MI355X-native per-pixel path-tracing hot path (reference: stneng/Ray-Tracing).

Layout:
  csrc/   hand-written HIP kernels for gfx950 + the C-ABI of include/rt_hip.h  -> librt_hip.so
  host/   C++ mirror of the reference's scene API, scenes.rs constructors, flattener -> librt_host.so
  renderer.py  launcher: flatten -> upload -> render (spp-sharded, optionally in checkpointed passes) -> all-reduce -> resolve
  flat.py      flat-scene builder + JSON scene loader (codegen/data/*.json of the reference)

The directory name carries a hyphen, so import it with
    importlib.import_module("ray-tracing_amd")      (or `import ray_tracing_amd`, the alias at the repo root)
"""
from ._ffi import RT_F32, RT_F64, RtError  # noqa: F401
from ._build import build_all  # noqa: F401


def __getattr__(name):  # renderer needs torch; keep `import` of the package itself light
    if name in ("HostScene", "FlatHostScene", "DeviceScene", "render", "render_progressive", "save_checkpoint", "load_checkpoint", "resolve", "write_png", "init", "spp_shard", "Comm", "Exchange", "comm_probe", "SCENES", "DEFAULT_SEED", "EARTHMAP"):
        from . import renderer
        return getattr(renderer, name)
    raise AttributeError(name)
