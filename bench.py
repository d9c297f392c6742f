#!/usr/bin/env python3
"""bench.py — Msamples/s of the per-pixel path-tracing hot path on N MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

Workload (BASELINE configs[1]): cornell_box, 1024x1024, mixture-PDF light importance sampling, depth 50.
A STEP is one rt_render launch over the whole frame adding `--spp-per-step` samples to every pixel; the default
is the config's 1024 spp in ONE launch, so a step is one complete render of the config (K = 4 renders by default;
shorter launches pay the drain of each tile's job list more often: 64 spp/launch 2690, 256: 2990, 1024: 3075 Msamples/s).  Scene and accumulation buffer are
resident in HBM before the timed region.  With N > 1 every rank renders its own disjoint sample indices
of the full frame (weak scaling: per-GPU work fixed) and the float accumulation buffers are summed by
ONE RCCL all-reduce inside the timed region (SURVEY §8 e).

value = all samples of all ranks / max-over-ranks wall time of the K steps (+ the all-reduce).
dtype "f64": the device path restates the reference's f64 arithmetic (bit-exact ids / t, sample-for-sample
equal to the CPU oracle); the f32 throughput mode is reported beside it as `f32_value`, never as `value`.

roofline: the path is VALU-bound (no MFMA, ~0 HBM traffic: SURVEY §8 d).  achieved = algorithmic
flops/sample (oracle operation counters with SURVEY §8(d)'s conventions, DESIGN.md "Roofline") x samples per
launch / mean launch duration (HIP events on the launch stream); peak = FP64 vector FMA rate of MI355X
(256 CU x 4 SIMD x 16 f64-lanes/clk x 2 flop x 2.4 GHz = 78.6 TFLOP/s).  The HBM view (accumulation buffer
read+write per launch against 8 TB/s) is reported under roofline.hbm for completeness.

cpu_baseline: the CPU oracle (oracle/, a C++ f64 restatement of the reference renderer; the Rust reference
cannot be built in this image) timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic flops per sample, measured by the oracle's operation counters
# (oracle/_build/oracle_render_counters <scene> 256 256 32 8; conventions of SURVEY §8 d).
# tests/test_bench_constants.py re-derives them from the oracle.
FLOPS_PER_SAMPLE = {
    "cornell_box": 1359.1,
    "random_scene": 17705.7,
    "cornell_smoke": 1746.8,
    "final_scene": 59739.0,
    "earth": 110.5,
    "two_perlin_spheres": 339.7,
}
PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}  # MI355X vector FMA peaks (MI355X_MICROARCH.md; f64 = half the f32 rate)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="cornell_box")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--spp-per-step", type=int, default=1024)
    ap.add_argument("--precision", default="f64", choices=("f64", "f32"))
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED0001)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f32", action="store_true", help="skip the extra f32 throughput measurement")
    ap.add_argument("--cpu-spp", type=int, default=0, help="spp of the CPU baseline sample (0 = sized to ~15 s)")
    ap.add_argument("--no-sort", action="store_true", help="run the wave-level kernel where the block-sorted one would be used")
    ap.add_argument("--out-png", default="")
    return ap.parse_args()


def cpu_baseline(scene, width, height, seed, cpu_spp):
    """Time oracle/_build/oracle_render (counters compiled out) on a bounded sample of the workload."""
    exe = os.path.join(ROOT, "oracle", "_build", "oracle_render")
    img = os.path.join(ROOT, "ray-tracing_amd", "assets", "earthmap.ppm")
    if not os.path.exists(exe):
        return None
    try:
        threads = len(os.sched_getaffinity(0))
    except Exception:
        threads = os.cpu_count() or 1
    threads = max(1, min(threads, int(os.environ.get("OMP_NUM_THREADS", "16")), 16))

    def run(spp):
        out = subprocess.run([exe, scene, str(width), str(height), str(spp), str(threads), hex(seed), img],
                             capture_output=True, text=True, timeout=600)
        if out.returncode != 0:
            raise RuntimeError(out.stderr)
        return json.loads(out.stdout.strip().splitlines()[-1])

    try:
        if cpu_spp <= 0:
            probe = run(2)  # ~1 s: sizes the real sample to about 15 s of CPU work
            cpu_spp = max(4, min(256, int(15.0 / max(probe["seconds"] / 2.0, 1e-3))))
        r = run(cpu_spp)
    except Exception as e:  # the baseline is reported, never required
        return {"value": None, "unit": "Msamples/s", "cores": threads, "kind": "port", "sample": f"failed: {e}"}
    return {"value": round(r["msamples_per_s"], 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"{scene} {width}x{height} @ {cpu_spp} spp of the same frame, depth 50, seed {seed:#x}, "
                      f"{r['seconds']:.2f} s, oracle/ C++ f64 restatement of the reference renderer (threads*16 x-strip jobs)"}


def main():
    a = parse()
    import torch
    import torch.distributed as dist

    import ray_tracing_amd as rt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if world == 1 and a.gpus != 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                         "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ  # under torchrun the RCCL path runs even at N = 1
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm

    desc = rt.init(local)
    W, H, S = a.width, a.height, a.spp_per_step
    hs = rt.HostScene(a.scene, W / H, seed=a.seed)
    ds = rt.DeviceScene(hs, local)
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    stats = torch.zeros(4, dtype=torch.int64, device=dev)

    def barrier():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def step_range(i):
        """Step i of this rank covers sample indices [(i*world + rank)*S, +S): disjoint over ranks and steps."""
        return (i * world + rank) * S

    def run_steps(first, n, precision, events=None):
        for i in range(first, first + n):
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            ds.render_into(accum, step_range(i), S, seed=a.seed, precision=precision, stats=stats, no_sort=a.no_sort)
            if events is not None:
                e1.record()
                events.append((e0, e1))

    def timed(precision, K, Wm, first):
        run_steps(first, Wm, precision)          # untimed warm-up (distinct sample indices)
        if use_dist:
            dist.all_reduce(accum.clone())       # warm RCCL up as well
        accum.zero_()
        stats.zero_()
        events = []
        barrier()
        t0 = time.perf_counter()
        run_steps(first + Wm, K, precision, events)
        if use_dist:
            dist.all_reduce(accum)               # the path's one exchange step (sum of accumulation buffers)
        barrier()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        launch_ms = [e0.elapsed_time(e1) for e0, e1 in events]
        return tmax.item(), launch_ms

    K, Wm = a.steps, a.warmup
    dt, launch_ms = timed(a.precision, K, Wm, 0)
    st = stats.cpu().numpy().astype("float64")
    total_samples = float(W) * H * S * K * world
    value = total_samples / dt / 1e6
    mean_launch_s = sum(launch_ms) / len(launch_ms) / 1e3
    image_mean = (accum.mean().item() / (S * K * world))
    if rank == 0 and a.out_png:
        rt.write_png(a.out_png, rt.resolve(accum, S * K * world))

    f32 = None
    if not a.no_f32 and a.precision == "f64" and world == 1:
        k32 = max(2, min(K, 4))
        dt32, lm32 = timed("f32", k32, 1, K + Wm)
        f32 = {"value": float(W) * H * S * k32 / dt32 / 1e6, "launch_ms": sum(lm32) / len(lm32)}

    if rank == 0:
        # which render kernel the launcher picked (rt_kernels.inc: launch_render)
        d = hs.desc
        has_bvh = d.bvh.n_nodes > 0
        has_medium = any(d.world[i].is_medium for i in range(d.n_world))
        real = "double" if a.precision == "f64" else "float"
        if not a.no_sort and not has_bvh and (d.n_lights > 0 or has_medium):
            kernel_name = f"render_kernel_sorted<{real}> (block-sorted)"
        else:
            kernel_name = f"render_kernel<{real}> (persistent wave jobs)"
        fps = FLOPS_PER_SAMPLE.get(a.scene)
        samples_per_launch = float(W) * H * S
        roofline = None
        if fps:
            achieved = fps * samples_per_launch / mean_launch_s / 1e12
            hbm_bytes = float(W) * H * 3 * 4 * 2  # accumulation buffer read + written once per launch
            roofline = {
                "bound": "valu", "achieved": round(achieved, 4), "peak": PEAK_TFLOPS[a.precision], "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_TFLOPS[a.precision], 5), "traffic": None,
                "kernel": kernel_name,
                "flops_per_sample": fps, "samples_per_launch": samples_per_launch, "launch_ms": round(mean_launch_s * 1e3, 4),
                "hbm": {"bound": "hbm", "achieved": round(hbm_bytes / mean_launch_s / 1e9, 3), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(hbm_bytes / mean_launch_s / 1e9 / PEAK_HBM_GBS, 7), "algorithmic_bytes_per_launch": hbm_bytes},
            }
            pmc = os.path.join(ROOT, "profiles", "hbm_traffic.json")  # written by tools/pmc_summary.py from rocprofv3 --pmc passes
            if os.path.exists(pmc):
                try:
                    rec = json.load(open(pmc))
                    key = f"{a.scene}_{W}x{H}_{S}spp_{a.precision}"
                    if key in rec:
                        roofline["traffic"] = rec[key]["hbm_bytes_per_launch"]
                except Exception:
                    pass
        cpu = None
        if world == 1 and not a.no_cpu_baseline:
            cpu = cpu_baseline(a.scene, W, H, a.seed, a.cpu_spp)
        line = {
            "metric": "Msamples/sec (paths/s) Cornell box 1024² at 1/2/4/8 MI355X vs host CPU" if a.scene == "cornell_box" and W == H == 1024
            else f"Msamples/sec (paths/s) {a.scene} {W}x{H}",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.precision, "data": "synthetic",
            "config": {"workload": f"{a.scene} {W}x{H}, {S} spp per step x {K} steps x {world} GPU = {S * K * world} spp, depth 50, "
                                   f"lights={'mixture-PDF importance sampling' if hs.desc.n_lights else 'none'}",
                       "scene": a.scene, "width": W, "height": H, "spp_per_step": S, "spp_total": S * K * world, "max_depth": 50,
                       "seed": a.seed, "parallelism": f"spp-shard x{world} + 1 all-reduce" if world > 1 else "single GPU", "device": desc},
            "per_gpu_value": round(value / world, 3),
            "casts_per_sample": round(st[1] / max(st[0], 1.0), 4), "nan_retries": int(st[2]), "dropped_samples": int(st[3]),
            "image_mean": round(image_mean, 6),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if f32:
            line["f32_value"] = round(f32["value"], 3)
            line["f32_launch_ms"] = round(f32["launch_ms"], 4)
        if cpu and cpu.get("value"):
            line["gpu_over_cpu"] = round(value / cpu["value"], 2)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
