"""Launcher: what `run_ray_tracing` is in the reference (src/ray_tracing.rs:65-135) — pick a scene, render it,
write the 8-bit image — with the hard-coded tuple / `CI` switch of ray_tracing.rs:68-78 turned into flags.

  python -m ray_tracing_amd --scene final_scene --width 800 --height 800 --spp 1024 --out output/ray_tracing.png
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m ray_tracing_amd --scene cornell_box ...

One process per GPU; under torchrun the spp range is sharded over the ranks and the float accumulation buffers are
summed with one RCCL all-reduce through the C-ABI (renderer.Exchange: rt_allreduce_accum; torch.distributed ships the id).  Prints one JSON record (scene, seed, size, seconds, Msamples/s, casts/sample, NaN retries)."""
import argparse
import json
import os
import sys
import time


def main(argv=None):
    ap = argparse.ArgumentParser(prog="ray_tracing_amd")
    ap.add_argument("--scene", default="final_scene")          # ray_tracing.rs:78
    ap.add_argument("--width", type=int, default=400)          # ray_tracing.rs:75 (local default 400x400 @ 64)
    ap.add_argument("--height", type=int, default=400)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--depth", type=int, default=50)           # ray_tracing.rs:105
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED0001)
    ap.add_argument("--precision", default="f64", choices=("f64", "f32"))
    ap.add_argument("--spp-per-launch", type=int, default=256)
    ap.add_argument("--out", default="output/ray_tracing.png")  # ray_tracing.rs:133
    ap.add_argument("--save-accum", default="", help="also save the linear float accumulation buffer (.npy)")
    ap.add_argument("--checkpoint", default="", help="progressive rendering: write the summed accumulation buffer here after every pass")
    ap.add_argument("--pass-spp", type=int, default=0, help="samples per pass (all GPUs together) between checkpoints; default: spp/8")
    ap.add_argument("--resume", action="store_true", help="continue the render recorded in --checkpoint (same scene, seed, size, depth)")
    ap.add_argument("--exchange", default="rt", choices=("rt", "torch"),
                    help="sum of the ranks' buffers: rt = rt_allreduce_accum (C-ABI, RCCL; falls back when a rank cannot), torch = torch.distributed")
    a = ap.parse_args(argv)

    import torch
    import torch.distributed as dist

    from . import renderer as rt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("no GPU: this renderer has no CPU path")
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    rt.init(local)
    # the exchange step: the C-ABI's one RCCL all-reduce (rt_allreduce_accum) when every rank can take that route
    ex = rt.Exchange(rank, world, dist, f"cuda:{local}", prefer=a.exchange) if world > 1 else None
    t0 = time.perf_counter()
    hs = rt.HostScene(a.scene, a.width / a.height, seed=a.seed)
    ds = rt.DeviceScene(hs, local)
    t_build = time.perf_counter() - t0
    stats = torch.zeros(4, dtype=torch.int64, device=f"cuda:{local}")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    spp_done = a.spp
    if a.checkpoint:
        acc, spp_done = rt.render_progressive(ds, a.width, a.height, a.spp, a.pass_spp or max(1, a.spp // 8), seed=a.seed, max_depth=a.depth,
                                       precision=a.precision, spp_per_launch=a.spp_per_launch, rank=rank, world_size=world,
                                       all_reduce=ex.all_reduce if ex else None, stats=stats, checkpoint=a.checkpoint,
                                       resume=a.resume, scene_name=a.scene)
    else:
        acc = rt.render(ds, a.width, a.height, a.spp, seed=a.seed, max_depth=a.depth, precision=a.precision,
                        spp_per_launch=a.spp_per_launch, rank=rank, world_size=world,
                        all_reduce=ex.all_reduce if ex else None, stats=stats)
    torch.cuda.synchronize()
    t_render = time.perf_counter() - t0
    if ex:
        ex.all_reduce_stats(stats)
    if rank == 0:
        rgb = rt.resolve(acc, spp_done)  # the sums hold spp_done samples per pixel
        if a.out:
            os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
            rt.write_png(a.out, rgb)
        if a.save_accum:
            import numpy as np
            np.save(a.save_accum, acc.cpu().numpy())
        st = stats.cpu().tolist()
        print(json.dumps({"scene": a.scene, "seed": a.seed, "width": a.width, "height": a.height, "spp": a.spp, "max_depth": a.depth,
                          "precision": a.precision, "n_gpus": world, "seconds_scene_build": round(t_build, 4),
                          "seconds_render": round(t_render, 4), "msamples_per_s": round(a.width * a.height * a.spp / t_render / 1e6, 3),
                          "casts_per_sample": round(st[1] / max(st[0], 1), 4), "nan_retries": st[2], "dropped_samples": st[3], "out": a.out,
                          "exchange": (ex.route if ex else None), **({"exchange_note": ex.note} if ex and ex.note else {})}))
    if ex:
        ex.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
