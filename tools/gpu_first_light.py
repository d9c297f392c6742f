"""First GPU contact: primary-ray parity on every scene, sample-for-sample f64 comparison, and a
rough f32/f64 render timing.  Run on the GPU box: python tools/gpu_first_light.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import ray_tracing_amd as rt
from oracle_lib import OracleScene

print(rt.init(0), flush=True)
W = H = 96
for name in rt.SCENES:
    hs = rt.HostScene(name, 1.0); ds = rt.DeviceScene(hs); oc = OracleScene(name, 1.0)
    for tm in (0.0, 0.5):
        oi, ot = oc.trace_primary(W, H, tm, 0.5)
        for prec in ("f64", "f32"):
            gi, gt = ds.trace_primary(W, H, tm, 0.5, prec)
            gi = gi.cpu().numpy(); gt = gt.cpu().numpy()
            same = gi == oi
            hit = same & (oi >= 0)
            rel = np.abs(gt[hit] - ot[hit]) / np.abs(ot[hit]) if hit.any() else np.zeros(1)
            print(f"{name:22s} t={tm} {prec}: id mismatches {int((~same).sum()):5d}/{W*H}  max rel t err {rel.max():.3e}  bit-exact t {bool(np.array_equal(gt[hit], ot[hit]))}", flush=True)

# sample-for-sample (f64)
for name in ("cornell_box", "random_scene", "cornell_smoke", "final_scene", "two_perlin_spheres", "earth"):
    hs = rt.HostScene(name, 1.0); ds = rt.DeviceScene(hs); oc = OracleScene(name, 1.0)
    x, y, n = 40, 30, 256
    g = ds.sample_pixel(W, H, x, y, 0, n, precision="f64").cpu().numpy()
    o = oc.sample_pixel(W, H, x, y, 0, n)
    err = np.abs(g - o) / np.maximum(np.abs(o), 1e-30)
    close = (np.abs(g - o) <= 1e-9 * np.maximum(np.abs(o), 1.0)).all(axis=1)
    print(f"{name:22s} sample-for-sample f64: {int(close.sum())}/{n} samples equal to 1e-9; mean oracle {o.mean(0)} mean gpu {g.mean(0)}", flush=True)

# timing
hs = rt.HostScene("cornell_box", 1.0); ds = rt.DeviceScene(hs)
for prec, spp in (("f32", 16), ("f64", 16), ("f32", 64), ("f64", 64)):
    acc = torch.zeros((1024, 1024, 3), dtype=torch.float32, device="cuda")
    stats = torch.zeros(4, dtype=torch.int64, device="cuda")
    ds.render_into(acc, 0, 1, precision=prec); torch.cuda.synchronize()
    acc.zero_()
    t0 = time.time(); ds.render_into(acc, 0, spp, precision=prec, stats=stats); torch.cuda.synchronize(); dt = time.time() - t0
    st = stats.cpu().numpy()
    print(f"cornell 1024^2 x {spp} spp {prec}: {dt:.3f} s  {1024*1024*spp/dt/1e6:.1f} Msamples/s  casts/sample {st[1]/max(st[0],1):.2f} nan {st[2]} mean {acc.mean().item()/spp:.5f}", flush=True)
rgb = rt.resolve(acc, 64)
os.makedirs("gpurun_out", exist_ok=True)
rt.write_png("gpurun_out/cornell_first.png", rgb)
