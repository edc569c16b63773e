"""Dev tool: dynamic-scene costs (update_transforms + sync = refit; push/delete + sync = rebuild)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raycore_jl_amd as rc
sc = rc.scenes
for n in (256, 5000, 100_000):
    g = sc.rng(1)
    xf = np.tile(sc.IDENTITY3x4, (n, 1)).astype(np.float32)
    xf[:, [3, 7, 11]] = g.uniform(-50, 50, size=(n, 3))
    t = rc.TLAS()
    t0 = time.perf_counter(); h = t.push(sc.fan_sphere(16, 9, radius=0.4), xf); t.sync(); build = time.perf_counter() - t0
    times = []
    for it in range(5):
        xf[:, 3] += 0.1
        t0 = time.perf_counter()
        t.update_transforms(h, xf)
        t.sync()
        times.append((time.perf_counter() - t0) * 1e3)
    assert t.last_sync_action == "refit"
    print(f"n_inst={n}: first build+sync {build*1e3:.2f} ms; update_transforms+sync(refit) ms {[round(x,3) for x in times]}")
