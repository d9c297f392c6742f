import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import ray_tracing_amd as rt
hs = rt.HostScene("cornell_box", 1.0); ds = rt.DeviceScene(hs, 0)
for (W, H, spp) in [(1024, 1024, 7), (1024, 1024, 3), (1024, 1024, 16), (512, 512, 7), (256, 256, 7), (1024, 1024, 33)]:
    a = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0"); b = torch.zeros_like(a)
    sa = torch.zeros(4, dtype=torch.int64, device="cuda:0"); sb = torch.zeros_like(sa)
    ds.render_into(a, 0, spp, precision="f64", stats=sa)
    ds.render_into(b, 0, spp, precision="f64", stats=sb, no_sort=True)
    torch.cuda.synchronize()
    bad = (~torch.isclose(a, b, rtol=1e-6, atol=1e-6)).any(dim=2)
    ys, xs = torch.nonzero(bad, as_tuple=True)
    print(W, H, spp, "bad pixels", int(bad.sum()), "stats", sa.tolist(), sb.tolist(), "sum", float(a.sum()), float(b.sum()))
    if len(ys):
        print("  first", [(int(y), int(x)) for y, x in zip(ys[:8], xs[:8])], "tiles", sorted(set((int(y)//8, int(x)//8) for y, x in zip(ys, xs)))[:10])
        y, x = int(ys[0]), int(xs[0]); print("  values", a[y, x].tolist(), b[y, x].tolist())
