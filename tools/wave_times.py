"""Measurement (RT_WAVE_TIMES build): mean and max run time of the persistent kernel's waves against the launch time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ray_tracing_amd as rt
name = sys.argv[1]; W, H, spp = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
hs = rt.HostScene(name, W / H); ds = rt.DeviceScene(hs, 0)
for k in range(3):
    acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    st = torch.zeros(4, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ds.render_into(acc, 0, spp, precision="f64", stats=st, no_sort=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = st.tolist()
    print(f"{name}: launch {dt*1e3:.2f} ms; {s[1]} waves, mean wave time {s[2]/max(s[1],1)/1e5:.2f} ms, max {s[3]/1e5:.2f} ms", flush=True)
