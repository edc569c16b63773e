"""Dev tool: the host-buffer entry point (rc_trace_closest: pageable host arrays in and out) end to end."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raycore_jl_amd as rc
from perf_probe import build

sc = rc.scenes
cfg3 = sc.config_c3()
t3 = build(cfg3)
for res in (512, 1024, 1449, 2048, 4096):
    rays = sc.c3_primary_rays(cfg3, res, res)
    t3.trace(rays)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        h = t3.trace(rays)
        best = min(best, time.perf_counter() - t0)
    from raycore_jl_amd._capi import check, lib, ptr
    hits = np.zeros(len(rays), dtype=rc.HIT_DT)
    best2 = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        check(lib().rc_trace_closest(t3._h, ptr(rays), ptr(hits), len(rays)))
        best2 = min(best2, time.perf_counter() - t0)
    print(f"{len(rays):9d} rays: rc_trace_closest into a reused, touched output buffer {best2 * 1e3:8.3f} ms = {len(rays) / best2 / 1e6:7.1f} Mrays/s")
    t3.host_register(rays); t3.host_register(hits)
    best3 = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        check(lib().rc_trace_closest(t3._h, ptr(rays), ptr(hits), len(rays)))
        best3 = min(best3, time.perf_counter() - t0)
    t3.host_unregister(hits); t3.host_unregister(rays)
    print(f"{len(rays):9d} rays: the same with both arrays page-locked (rc_host_register)      {best3 * 1e3:8.3f} ms = {len(rays) / best3 / 1e6:7.1f} Mrays/s "
          f"({64 * len(rays) / best3 / 1e9:.1f} GB/s over the bus)")
    print(f"{len(rays):9d} rays: host-buffer trace {best * 1e3:8.3f} ms wall = {len(rays) / best / 1e6:7.1f} Mrays/s (kernel {t3.last_kernel_ms():.3f} ms; "
          f"{64 * len(rays) / best / 1e9:.1f} GB/s over the bus)", flush=True)
