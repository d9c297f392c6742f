#!/usr/bin/env python3
"""Generate tests/golden/* from the CPU oracle (oracle/_build/liboracle.so).

The reference holds no golden vectors for this path (its only tests are the Vec3 unit tests), so
these fixtures pin the ORACLE's outputs: regressions in the oracle or the host scene constructors
show up on CPU, and the GPU tests compare the HIP path against the same files.

  primary_<scene>.npz   jitter-free primary-ray probe, 64x64, aspect 1, seed 0x5EED0001,
                        time 0.0 and 0.5, medium_xi 0.5: ids int32, t float64
  cornell_128_4096.npz  converged linear image (mean radiance, float32) of cornell_box 128x128 at
                        4096 spp, seeds A=0x5EED0001 and B=0xB0B0CAFE (B gives the Monte-Carlo
                        noise floor of the 1 % RMSE test)
  samples_<scene>.npz   per-sample radiance of pixel (20,13) of a 48x48 frame, samples 0..63
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle_lib import OracleScene, DEFAULT_SEED

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
SCENES = ("random_scene", "random_scene_light", "two_checker_spheres", "two_perlin_spheres", "earth", "simple_light",
          "cornell_box", "cornell_smoke", "final_scene")
what = sys.argv[1:] or ["primary", "samples", "converged"]
if "primary" in what:
    for name in SCENES:
        oc = OracleScene(name, 1.0)
        i0, t0 = oc.trace_primary(64, 64, 0.0, 0.5)
        i1, t1 = oc.trace_primary(64, 64, 0.5, 0.5)
        np.savez_compressed(os.path.join(OUT, f"primary_{name}.npz"), ids_t0=i0, t_t0=t0, ids_t05=i1, t_t05=t1,
                            seed=np.uint64(DEFAULT_SEED), medium_xi=0.5)
        print(name, "hits", int((i0 >= 0).sum()), "distinct ids", len(np.unique(i0)))
if "samples" in what:
    for name in SCENES:
        oc = OracleScene(name, 1.0)
        s = oc.sample_pixel(48, 48, 20, 13, 0, 64)
        np.savez_compressed(os.path.join(OUT, f"samples_{name}.npz"), rgb=s, seed=np.uint64(DEFAULT_SEED))
        print(name, "mean sample", s.mean(0))
if "converged" in what:
    oc = OracleScene("cornell_box", 1.0)
    res = {}
    for tag, seed in (("a", DEFAULT_SEED), ("b", 0xB0B0CAFE)):
        t = time.time()
        acc, cnt = oc.render(128, 128, 0, 4096, seed=seed, threads=8)
        res[tag] = (acc / 4096.0).astype(np.float32)
        res["counters_" + tag] = np.array([cnt[k] for k in sorted(cnt)])
        print("converged", tag, f"{time.time()-t:.1f}s", cnt)
    np.savez_compressed(os.path.join(OUT, "cornell_128_4096.npz"), mean_a=res["a"], mean_b=res["b"],
                        seed_a=np.uint64(DEFAULT_SEED), seed_b=np.uint64(0xB0B0CAFE), spp=4096)
