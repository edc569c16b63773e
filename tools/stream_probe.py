import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
import raycore_jl_amd as rc
sc = rc.scenes
cfg = sc.config_c2()
t = rc.TLAS(0)
t.add_geometry(*cfg["blas"][0])
t.push_instances(1, cfg["instances"][0][1], cfg["instances"][0][2])
t.sync()
rays = rc.generate_ray_grid(t, cfg["viewdir"], cfg["grid"])
dr = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
dh = torch.empty(len(rays) * 32, dtype=torch.uint8, device="cuda")
side = torch.cuda.Stream()
torch.cuda.synchronize()
for name, st in (("null", None), ("side", side.cuda_stream), ("null", None), ("side", side.cuda_stream)):
    best = 1e9
    for _ in range(30):
        t.trace_device(dr.data_ptr(), dh.data_ptr(), len(rays), stream=st)
        best = min(best, t.last_kernel_ms())
    print(name, best, len(rays) / best / 1e3)
for k in (3, 5):
    t.set_option("kernel", k)
    best = 1e9
    for _ in range(30):
        t.trace_device(dr.data_ptr(), dh.data_ptr(), len(rays), stream=side.cuda_stream)
        best = min(best, t.last_kernel_ms())
    print("kernel", k, "side", best, len(rays) / best / 1e3)
import time
t.set_option("kernel", -1)
t0 = time.time()
while time.time() - t0 < 4.0:
    best = 1e9
    c0 = time.time()
    while time.time() - c0 < 0.5:
        t.trace_device(dr.data_ptr(), dh.data_ptr(), len(rays), stream=side.cuda_stream)
        best = min(best, t.last_kernel_ms())
    print(f"t={time.time()-t0:.1f}s best {best:.3f} ms {len(rays)/best/1e3:.0f} Mrays/s", flush=True)
