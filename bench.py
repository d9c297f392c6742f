#!/usr/bin/env python3
"""bench.py — Msamples/s of the per-pixel path-tracing hot path on N MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]

Launch.  `python bench.py --gpus N` works from a plain shell: for N > 1 this process starts — BEFORE it imports torch or
touches a GPU — `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
bench.py ...` as a child, relays rank 0's JSON line and exits with the child's code.  When it is already running under
torchrun (WORLD_SIZE / RANK in the environment: the driver's form) it is a rank and just runs.

Workload (BASELINE configs[1]): cornell_box, 1024x1024, 1024 spp, mixture-PDF light importance sampling, depth 50.
A STEP is ONE COMPLETE RENDER of the config: the frame's `--spp` samples per pixel are sharded over the ranks
(rank r renders sample indices spp_shard(spp, r, N) of every pixel — src/ray_tracing.rs:83-94 is the reference's
only split, over x-strips; here the split is over samples so that every GPU sees the same mix of path lengths), each
rank runs ONE rt_render launch over the whole frame into its zeroed float buffer, and with N > 1 the buffers are summed by
ONE RCCL all-reduce through the C-ABI (rt_allreduce_accum: the gather of src/ray_tracing.rs:119-131) — zeroing,
launch and all-reduce are all inside the timed region, once per step.  Total work is fixed as N grows: "scaling":
"strong".  `--weak` keeps the per-GPU work fixed instead (every rank renders `--spp` samples per step).
Scene and buffers are resident in HBM before the timed region.

value = samples of all ranks / max-over-ranks wall time of the K steps.
dtype "f64": the device path restates the reference's f64 arithmetic (bit-exact ids / t, sample-for-sample equal to the
CPU oracle); the f32 throughput mode is reported beside it as `f32_value`, never as `value`.

roofline: the path is VALU-bound (no MFMA, ~0 HBM traffic: SURVEY §8 d).  achieved = algorithmic flops/sample (oracle
operation counters with SURVEY §8(d)'s conventions, DESIGN.md "Roofline") x samples per launch / mean launch duration
(HIP events on the launch stream); peak = FP64 vector FMA rate of MI355X (256 CU x 4 SIMD x 16 f64-lanes/clk x 2 flop
x 2.4 GHz = 78.6 TFLOP/s).  The flop convention counts the REFERENCE's work (its loose AABB test visits ~27 % of a
tree, the GPU culls that), so `frac` is not comparable across scenes; `executed` beside it is the hardware view from
the committed rocprofv3 PMC passes of the same workload: share of all VALU lane-slots (1024 SIMDs x 16 lanes x cycles)
that executed a lane-instruction.  The HBM view is reported under roofline.hbm.

cpu_baseline: the CPU oracle (oracle/, a C++ f64 restatement of the reference renderer; the Rust reference cannot be
built in this image) timed on ALL of this box's host threads on a bounded sample of the same workload, with the figure
at the reference's default of 16 threads (src/ray_tracing.rs:75) beside it.
"""
import argparse
import json
import os
import socket
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


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="cornell_box")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--spp", "--spp-per-step", dest="spp", type=int, default=1024,
                    help="samples per pixel of one step = one complete render (sharded over the ranks unless --weak)")
    ap.add_argument("--weak", action="store_true", help="weak scaling: every rank renders --spp samples per step")
    ap.add_argument("--precision", default="f64", choices=("f64", "f32"))
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED0001)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f32", action="store_true", help="skip the extra f32 throughput measurement")
    ap.add_argument("--cpu-spp", type=int, default=0, help="spp of the CPU baseline sample (0 = sized to ~12 s)")
    ap.add_argument("--no-sort", action="store_true", help="run the wave-level kernel where the block-sorted one would be used")
    ap.add_argument("--deterministic", action="store_true", help="RT_RENDER_DETERMINISTIC: the reproducible accumulation mode (one lane per pixel)")
    ap.add_argument("--exchange", default="rt", choices=("rt", "torch"),
                    help="all-reduce of the accumulation buffers: rt = rt_allreduce_accum (C-ABI, RCCL); torch = torch.distributed")
    ap.add_argument("--out-png", default="")
    ap.add_argument("--dry-run", action="store_true",
                    help="plumbing check without a GPU (tests/test_bench_launch.py): launcher, rank environment, shard arithmetic and one "
                         "gloo all-reduce of per-pixel SAMPLE COUNTS; renders nothing and measures nothing")
    return ap.parse_args(argv)


def shard(spp, rank, world):
    """Samples [begin, begin+count) of every pixel for rank `rank` of `world` (same rule as renderer.spp_shard)."""
    return spp * rank // world, spp * (rank + 1) // world - spp * rank // world


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(n, argv, port=None):
    """The child the parent starts for --gpus N > 1 (one fresh process per GPU; the parent never touches a GPU)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port or free_port()), os.path.abspath(__file__)] + list(argv)


def self_launch(a, argv):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(launcher_command(a.gpus, argv), env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line:
        print(line, flush=True)
    if proc.returncode != 0:
        raise SystemExit(proc.returncode)
    if not line:
        raise SystemExit("bench.py: the ranks printed no result line")


def host_threads():
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown CPU"


def cpu_baseline(scene, width, height, seed, cpu_spp):
    """Time oracle/_build/oracle_render (counters compiled out) on a bounded sample of the workload: all host threads
    this process may use, and the reference's default of 16 threads (src/ray_tracing.rs:75) beside it."""
    exe = os.path.join(ROOT, "oracle", "_build", "oracle_render")
    img = os.path.join(ROOT, "ray-tracing_amd", "assets", "earthmap.ppm")
    if not os.path.exists(exe):
        return None
    threads = max(1, host_threads())
    nproc = os.cpu_count() or threads

    def run(spp, th):
        out = subprocess.run([exe, scene, str(width), str(height), str(spp), str(th), hex(seed), img],
                             capture_output=True, text=True, timeout=600)
        if out.returncode != 0:
            raise RuntimeError(out.stderr)
        return json.loads(out.stdout.strip().splitlines()[-1])

    try:
        if cpu_spp <= 0:
            probe = run(2, threads)  # sizes the real sample to about 12 s of wall time on all threads
            cpu_spp = max(4, min(1024, int(12.0 / max(probe["seconds"] / 2.0, 1e-3))))
        r = run(cpu_spp, threads)
        r16 = None
        if threads != 16:
            spp16 = max(2, min(cpu_spp, int(cpu_spp * 16 / threads))) if threads > 16 else cpu_spp
            r16 = run(spp16, 16)
    except Exception as e:  # the baseline is reported, never required
        return {"value": None, "unit": "Msamples/s", "cores": threads, "kind": "port", "sample": f"failed: {e}"}
    res = {"value": round(r["msamples_per_s"], 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
           "sample": f"{scene} {width}x{height} @ {cpu_spp} spp of the same frame, depth 50, seed {seed:#x}, {r['seconds']:.2f} s on {threads} threads "
                     f"(all this process may use; nproc {nproc}, {cpu_model()}); oracle/ C++ f64 restatement of the reference renderer "
                     f"(threads*16 x-strip jobs, src/ray_tracing.rs:83-94)"}
    if r16:
        res["value_16_threads"] = round(r16["msamples_per_s"], 4)
        res["note_16_threads"] = "the reference's default thread count (src/ray_tracing.rs:75)"
    return res


def executed_view(scene, W, H, S, precision):
    """Hardware view of the same workload from the committed rocprofv3 PMC passes (profiles/pmc_views.json, written by
    tools/pmc_summary.py): VALU lane-slot share, lanes active per VALU instruction, HBM bytes per launch."""
    p = os.path.join(ROOT, "profiles", "pmc_views.json")
    try:
        rec = json.load(open(p))
    except Exception:
        return None
    return rec.get(f"{scene}_{W}x{H}_{S}spp_{precision}")


def dry_run(a, in_rank):
    """No GPU, no rendering: every rank writes the number of samples it WOULD render into a small per-pixel buffer, the
    buffers are summed with gloo, and rank 0 checks that every pixel got exactly the step's spp and prints the ranges."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if in_rank:
        if a.gpus != world:
            raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
        dist.init_process_group("gloo")
    begin, count = (rank * a.spp, a.spp) if a.weak else shard(a.spp, rank, world)
    total = a.spp * world if a.weak else a.spp
    counts = torch.full((4, 4, 3), float(count))
    ranges = torch.zeros(world, 2, dtype=torch.int64)
    ranges[rank, 0], ranges[rank, 1] = begin, count
    if in_rank:
        dist.all_reduce(counts)
        dist.all_reduce(ranges)
    ok = bool((counts == float(total)).all())
    if rank == 0:
        print(json.dumps({"metric": "dry-run (no GPU work)", "value": 0.0, "unit": "Msamples/s", "n_gpus": world, "dry_run": True,
                          "scaling": "weak" if a.weak else "strong", "spp_per_step": total, "ranges": ranges.tolist(), "counts_ok": ok}), flush=True)
    if in_rank:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)


def main():
    a = parse()
    in_rank = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if a.gpus > 1 and not in_rank:
        return self_launch(a, sys.argv[1:])  # nothing below runs in the parent: it never imports torch

    if a.dry_run:
        return dry_run(a, in_rank)

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC; must be in the environment before the first HIP call
    import torch
    import torch.distributed as dist

    import ray_tracing_amd as rt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if in_rank and a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    use_dist = in_rank  # under torchrun the exchange step runs even at N = 1 (a 1-rank RCCL all-reduce)
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm; plumbing: barrier, max of the clocks, id broadcast

    desc = rt.init(local)
    W, H = a.width, a.height
    hs = rt.HostScene(a.scene, W / H, seed=a.seed)
    ds = rt.DeviceScene(hs, local)
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    stats = torch.zeros(4, dtype=torch.int64, device=dev)

    # the exchange step: rt_allreduce_accum through the C-ABI when every rank can take that route (renderer.Exchange checks it
    # on all ranks and falls back to torch.distributed for ALL of them otherwise; the line then says so)
    ex = rt.Exchange(rank, world, dist, dev, prefer=a.exchange) if use_dist else None
    comm = ex.comm if ex else None
    exchange_note = ex.note if ex else None

    def all_reduce(buf):
        ex.all_reduce(buf)

    def barrier():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # samples of one step on this rank: strong = this rank's shard of the frame's spp; weak = spp for every rank
    if a.weak:
        my_begin, my_count, step_spp_total = rank * a.spp, a.spp, a.spp * world
    else:
        (my_begin, my_count), step_spp_total = shard(a.spp, rank, world), a.spp

    def run_steps(first, n, precision, events=None, ar_events=None):
        for i in range(first, first + n):
            accum.zero_()                                    # a step is a complete render: fresh sums
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if my_count:
                ds.render_into(accum, i * step_spp_total + my_begin, my_count, seed=a.seed, precision=precision, stats=stats, no_sort=a.no_sort, deterministic=a.deterministic)
            if events is not None:
                e1.record()
                events.append((e0, e1))
            if use_dist:                                     # the path's one exchange step, once per render
                if ar_events is not None:
                    e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e2.record()
                all_reduce(accum)
                if ar_events is not None:
                    e3.record()
                    ar_events.append((e2, e3))

    def timed(precision, K, Wm, first):
        run_steps(first, Wm, precision)          # untimed warm-up (distinct sample indices; warms RCCL up as well)
        stats.zero_()
        events, ar_events = [], []
        barrier()
        t0 = time.perf_counter()
        run_steps(first + Wm, K, precision, events, ar_events)
        barrier()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        launch_ms = [e0.elapsed_time(e1) for e0, e1 in events]
        ar_ms = [e0.elapsed_time(e1) for e0, e1 in ar_events]
        return tmax.item(), launch_ms, ar_ms

    K, Wm = a.steps, a.warmup
    dt, launch_ms, ar_ms = timed(a.precision, K, Wm, 0)
    if ex:
        ex.all_reduce_stats(stats)
    st = stats.cpu().numpy().astype("float64")
    total_samples = float(W) * H * step_spp_total * K
    value = total_samples / dt / 1e6
    mean_launch_s = sum(launch_ms) / len(launch_ms) / 1e3
    image_mean = accum.mean().item() / step_spp_total   # the last step's complete render
    if rank == 0 and a.out_png:
        rt.write_png(a.out_png, rt.resolve(accum, step_spp_total))

    # single-GPU reference on the same box (rank 0; the other ranks wait at the barrier): the whole frame's spp in one launch
    single = None
    if world > 1 and not a.weak:
        if rank == 0:
            ev = []
            for i in range(3):
                accum.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ds.render_into(accum, (K + Wm + i) * a.spp, a.spp, seed=a.seed, precision=a.precision, no_sort=a.no_sort, deterministic=a.deterministic)
                e1.record()
                ev.append((e0, e1))
            torch.cuda.synchronize(dev)
            ms = sorted(e0.elapsed_time(e1) for e0, e1 in ev)[1]
            single = {"value": round(float(W) * H * a.spp / (ms * 1e-3) / 1e6, 3), "launch_ms": round(ms, 4),
                      "what": f"rank 0 alone, {a.spp} spp in one launch, median of 3 (device time of the launch, no exchange)"}
        barrier()

    f32 = None
    if not a.no_f32 and a.precision == "f64" and world == 1:
        k32 = max(2, min(K, 4))
        dt32, lm32, _ = timed("f32", k32, 1, K + Wm)
        f32 = {"value": float(W) * H * step_spp_total * k32 / dt32 / 1e6, "launch_ms": sum(lm32) / len(lm32)}

    if rank == 0:
        # which render kernel the launcher picked (rt_kernels.inc: launch_render)
        d = hs.desc
        has_bvh = d.bvh.n_nodes > 0
        has_medium = any(d.world[i].is_medium for i in range(d.n_world))
        real = "double" if a.precision == "f64" else "float"
        if a.deterministic:
            kernel_name = f"render_kernel<{real}, DET> (one lane per pixel, RT_RENDER_DETERMINISTIC)"
        elif not a.no_sort and not has_bvh and (d.n_lights > 0 or has_medium):
            kernel_name = f"render_kernel_sorted<{real}> (block-sorted)"
        else:
            kernel_name = f"render_kernel<{real}> (persistent wave jobs)"
        fps = FLOPS_PER_SAMPLE.get(a.scene)
        samples_per_launch = float(W) * H * my_count   # rank 0's launch
        roofline = None
        if fps:
            achieved = fps * samples_per_launch / mean_launch_s / 1e12
            hbm_bytes = float(W) * H * 3 * 4 * 2  # accumulation buffer read + written once per launch
            view = None if (a.deterministic or a.no_sort) else executed_view(a.scene, W, H, my_count, a.precision)
            live_build = desc.split("build=")[1].split()[0] if "build=" in desc else None  # the library that just ran (rt_describe)
            roofline = {
                "bound": "valu", "achieved": round(achieved, 4), "peak": PEAK_TFLOPS[a.precision], "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_TFLOPS[a.precision], 5),
                "traffic": view.get("hbm_bytes_per_launch") if view else None,
                "kernel": kernel_name,
                "flops_per_sample": fps, "samples_per_launch": samples_per_launch, "launch_ms": round(mean_launch_s * 1e3, 4),
                # `traffic` and `executed` are LOOKED UP from the committed rocprofv3 PMC passes of this workload (profiles/), not
                # measured by this run: `executed.build` / `.head` say which build those passes profiled, `build` which one ran now
                "build": live_build,
                "executed": (dict({k: view[k] for k in ("valu_lane_slot_frac", "valu_lanes_active_frac", "valu_issue_busy_frac", "from", "build", "head")
                                   if k in view}, of_this_build=(view.get("build") == live_build and live_build is not None))
                             if view else None),
                "hbm": {"bound": "hbm", "achieved": round(hbm_bytes / mean_launch_s / 1e9, 3), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(hbm_bytes / mean_launch_s / 1e9 / PEAK_HBM_GBS, 7), "algorithmic_bytes_per_launch": hbm_bytes},
            }
        cpu = None
        if world == 1 and not a.no_cpu_baseline:
            cpu = cpu_baseline(a.scene, W, H, a.seed, a.cpu_spp)
        per_rank = f"{my_count} spp per GPU per step" if not a.weak else f"{a.spp} spp per GPU per step"
        line = {
            "metric": "Msamples/sec (paths/s) Cornell box 1024² at 1/2/4/8 MI355X vs host CPU" if a.scene == "cornell_box" and W == H == 1024
            else f"Msamples/sec (paths/s) {a.scene} {W}x{H}",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True, "scaling": "weak" if a.weak else "strong", "vs_baseline": None,
            "dtype": a.precision, "data": "synthetic",
            "config": {"workload": f"{a.scene} {W}x{H} @ {step_spp_total} spp per step (one complete render; {per_rank}), {K} steps, depth 50, "
                                   f"lights={'mixture-PDF importance sampling' if hs.desc.n_lights else 'none'}",
                       "scene": a.scene, "width": W, "height": H, "spp_per_step": step_spp_total, "spp_per_gpu_per_step": my_count, "max_depth": 50,
                       "seed": a.seed,
                       "parallelism": (f"spp-shard x{world} + 1 all-reduce per render ({'rt_allreduce_accum, RCCL via the C-ABI' if comm is not None else 'torch.distributed nccl'})"
                                       if use_dist else "single GPU"),
                       "device": desc},
            "per_gpu_value": round(value / world, 3),
            "casts_per_sample": round(st[1] / max(st[0], 1.0), 4), "nan_retries": int(st[2]), "dropped_samples": int(st[3]),
            "image_mean": round(image_mean, 6),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if exchange_note:
            line["exchange_note"] = exchange_note
        if ar_ms:
            line["allreduce_ms"] = round(sum(ar_ms) / len(ar_ms), 4)
            line["allreduce_bytes"] = W * H * 12
        if single:
            line["single_gpu_reference"] = single
            line["per_gpu_efficiency_vs_single"] = round(value / world / single["value"], 4)
        if f32:
            line["f32_value"] = round(f32["value"], 3)
            line["f32_launch_ms"] = round(f32["launch_ms"], 4)
        if cpu and cpu.get("value"):
            line["gpu_over_cpu"] = round(value / cpu["value"], 2)
        print(json.dumps(line), flush=True)
    if ex:
        ex.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
