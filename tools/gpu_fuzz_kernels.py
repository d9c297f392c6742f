#!/usr/bin/env python3
"""Soak test on the GPU of the job machinery of the two render kernels (persistent waves / blocks whose jobs stream into each other):
generated flat scenes (tests/flat_fuzz.py) at a frame of several jobs per resident wave / block and few samples, where the oracle
would take minutes.  Checked without it: (1) the block-sorted kernel and the wave kernel produce the same sums and counters,
(2) one launch == two launches of parts of its samples (other chunks, other job streams), (3) every sample is counted once.
Run by hand:  python tools/gpu_fuzz_kernels.py FIRST LAST [generator]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import ray_tracing_amd as rt  # noqa: E402
import flat_fuzz  # noqa: E402

first, last = int(sys.argv[1]), int(sys.argv[2])
gen = getattr(flat_fuzz, sys.argv[3] if len(sys.argv) > 3 else "random_scene")
rt.init(0)
bad = 0
for seed in range(first, last):
    host = rt.FlatHostScene(gen(seed) if gen.__code__.co_argcount else gen())  # (nan_scene takes no seed)
    ds = rt.DeviceScene(host, 0)
    W, H = (1024, 640) if seed % 2 else (701, 1100)
    spp = 2 + seed % 5
    msg = []
    frames, stats = [], []
    for no_sort in (False, True):
        acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
        st = torch.zeros(4, dtype=torch.int64, device="cuda:0")
        ds.render_into(acc, 1, spp, precision="f64", stats=st, no_sort=no_sort)
        frames.append(acc); stats.append(st.tolist())
    if stats[0] != stats[1]:
        msg.append(f"counters differ between the kernels: {stats}")
    if stats[0][0] + stats[0][3] != W * H * spp:
        msg.append(f"samples counted {stats[0][0]} + dropped {stats[0][3]} != {W * H * spp}")
    finite = torch.isfinite(frames[0]) & torch.isfinite(frames[1])
    if not torch.equal(torch.isfinite(frames[0]), torch.isfinite(frames[1])) or not torch.allclose(frames[0][finite], frames[1][finite], rtol=4e-6, atol=2e-6):
        msg.append(f"frames of the two kernels differ at {(~torch.isclose(frames[0], frames[1], rtol=4e-6, atol=2e-6, equal_nan=True)).any(dim=2).sum().item()} pixels")
    parts = torch.zeros_like(frames[0])
    k = 1 + seed % 2
    ds.render_into(parts, 1, k, precision="f64")
    ds.render_into(parts, 1 + k, spp - k, precision="f64")
    finite = torch.isfinite(frames[0]) & torch.isfinite(parts)
    if not torch.allclose(frames[0][finite], parts[finite], rtol=4e-6, atol=2e-6):
        msg.append("one launch != two launches of parts of the samples")
    ds.close(); host.close()
    if msg:
        bad += 1
        print(f"seed {seed}: " + "; ".join(msg), flush=True)
    elif seed % 20 == 0:
        print(f"seed {seed} ok", flush=True)
print(f"{last - first} scenes ({gen.__name__}), {bad} with differences")
sys.exit(1 if bad else 0)
