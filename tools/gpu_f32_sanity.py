#!/usr/bin/env python3
"""RT_F32 on the GPU over the scene generators of tests/flat_fuzz.py: every render ends, is finite where the f64 render is, and its
mean agrees with the f64 render's within Monte-Carlo distance (f32 has no bit-exactness bar: DESIGN.md §3).
  python tools/gpu_f32_sanity.py FIRST LAST [generator ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import flat_fuzz  # noqa: E402
import ray_tracing_amd as rt  # noqa: E402

first, last = int(sys.argv[1]), int(sys.argv[2])
gens = sys.argv[3:] or ["random_scene", "medium_scene", "texture_scene", "extreme_scene", "far_scene", "odd_lights_scene", "pile_scene",
                        "instancing_scene", "tie_scene"]
rt.init(0)
bad = 0
for g in gens:
    worst = 0.0
    for seed in range(first, last):
        ds = rt.DeviceScene(rt.FlatHostScene(getattr(flat_fuzz, g)(seed)), 0)
        W, H = 48, 36

        def compare(spp):
            frames = {}
            for precision in ("f64", "f32"):
                acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
                st = torch.zeros(4, dtype=torch.int64, device="cuda:0")
                ds.render_into(acc, 0, spp, precision=precision, stats=st)
                frames[precision] = (acc.cpu().double(), st.cpu().tolist())
            a, b = frames["f64"][0], frames["f32"][0]
            fin = torch.isfinite(a)
            ok = bool(torch.isfinite(b[fin]).all())
            # robust comparison: means of the per-pixel values clipped at the 99th percentile (fireflies differ between the modes)
            cap = torch.quantile(a[fin], 0.99).item() if fin.any() else 0.0
            ma, mb = a[fin].clamp(max=cap).mean().item(), b[fin].clamp(max=cap).mean().item()
            return frames, ok, ma, mb, abs(ma - mb) / max(abs(ma), 1e-9)

        spp = 16
        frames, ok, ma, mb, rel = compare(spp)
        if g == "far_scene":
            rel = 0.0  # coordinates ~1e6 with features of size ~1: f32 resolves 0.06-0.25 there; only "ends, finite, every sample counted" is asked
        if ok and rel > 0.2:
            # heavy-tailed frames (extreme_scene: emitters of 1e3 behind small openings) differ by that much between two f64 renders
            # with different generator seeds at 16 spp (measured: clipped means 2.7 / 3.7 / 3.9): decide at 1024 spp, where the noise
            # is an eighth (f64 25.3 / 26.8 / 25.0 against f32 25.1 / 27.1 / 24.7 on the two seeds that first tripped this)
            spp = 1024
            frames, ok, ma, mb, rel = compare(spp)
            rel = rel + 0.05 if rel > 0.15 else rel  # (keeps the 0.2 threshold below; 0.15 at the higher sample count)
        ds.close()
        worst = max(worst, rel)
        if not ok or rel > 0.2 or frames["f32"][1][0] + frames["f32"][1][3] != W * H * spp:
            bad += 1
            print(f"{g} seed {seed}: finite={ok} clipped means {ma:.5g} / {mb:.5g} stats {frames['f64'][1]} / {frames['f32'][1]}", flush=True)
    print(f"{g}: {last - first} scenes, largest relative difference of the clipped means {worst:.3f}", flush=True)
print(f"done: {bad} suspicious", flush=True)
sys.exit(1 if bad else 0)
