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
        W, H, spp = 48, 36, 16
        frames = {}
        for precision in ("f64", "f32"):
            acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
            st = torch.zeros(4, dtype=torch.int64, device="cuda:0")
            ds.render_into(acc, 0, spp, precision=precision, stats=st)
            frames[precision] = (acc.cpu().double(), st.cpu().tolist())
        ds.close()
        a, b = frames["f64"][0], frames["f32"][0]
        fin = torch.isfinite(a)
        ok = bool(torch.isfinite(b[fin]).all())
        # robust comparison: medians of the per-pixel values clipped at the 99th percentile (fireflies differ between the modes)
        cap = torch.quantile(a[fin], 0.99).item() if fin.any() else 0.0
        ma, mb = a[fin].clamp(max=cap).mean().item(), b[fin].clamp(max=cap).mean().item()
        rel = abs(ma - mb) / max(abs(ma), 1e-9)
        worst = max(worst, rel)
        if not ok or rel > 0.2 or frames["f32"][1][0] + frames["f32"][1][3] != W * H * spp:
            bad += 1
            print(f"{g} seed {seed}: finite={ok} clipped means {ma:.5g} / {mb:.5g} stats {frames['f64'][1]} / {frames['f32'][1]}", flush=True)
    print(f"{g}: {last - first} scenes, largest relative difference of the clipped means {worst:.3f}", flush=True)
print(f"done: {bad} suspicious", flush=True)
sys.exit(1 if bad else 0)
