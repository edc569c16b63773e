"""What do the timing events around every launch cost?  Wall-clock time of 60 back-to-back launches of one batch on one stream, per launch,
under RC_EVENT_MODE (0 default events, 1 hipEventDisableSystemFence, 2 no start event, 3 no events), cost_order on / off."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raycore_jl_amd as rc
from tools.perf_probe import build, to_dev
sc = rc.scenes
cfg2 = sc.config_c2(); t2 = build(cfg2)
rays2 = rc.generate_ray_grid(t2, cfg2["viewdir"], cfg2["grid"])
cfg3 = sc.config_c3(); t3 = build(cfg3)
rays3 = sc.c3_primary_rays(cfg3, 2048, 2048)
for name, t, rays in (("C2 1M", t2, rays2), ("C3 4Mi", t3, rays3)):
    d_r, d_h = to_dev(rays), torch.empty(len(rays) * 32, dtype=torch.uint8, device="cuda")
    for co in (0, 1):
        t.set_option("cost_order", co)
        for _ in range(10):
            t.trace_device(d_r.data_ptr(), d_h.data_ptr(), len(rays))
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(60):
                t.trace_device(d_r.data_ptr(), d_h.data_ptr(), len(rays))
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 60)
        ev = t.last_kernel_ms() if int(os.environ.get("RC_EVENT_MODE", "0")) < 2 else float("nan")
        print(f"RC_EVENT_MODE={os.environ.get('RC_EVENT_MODE', '0')} {name} cost_order={co}: {best * 1e3:.4f} ms per launch by the wall clock over 60 launches, {len(rays) / best / 1e6:.1f} Mrays/s; last launch by its events {ev:.4f} ms", flush=True)
