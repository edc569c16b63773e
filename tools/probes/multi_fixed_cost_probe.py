"""Dev (GPU): the FIXED cost of a multi-rank totals call -- host threads, memsets, launches, the grouped reduce, the read-back -- measured with 8 scenes on
device 0 as 8 ranks against the stub communicator (tests/fake_rccl), at ray counts so small that tracing is negligible.  What it cannot show: RCCL's own
latency over xGMI.    python tools/probes/multi_fixed_cost_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from fake_rccl.build import build
so = build()
os.environ["RC_RCCL_LIBRARY"] = so; os.environ["RC_ENABLE_DEBUG_HOOKS"] = "1"; os.environ["RC_DEBUG_RANKS_SHARE_DEVICE"] = "1"
import numpy as np
import raycore_jl_amd as rc
from helpers import build_product
cfg = rc.scenes.config_c5()
for g in (1, 2, 8):
    scenes = [build_product(rc, cfg) for _ in range(g)]
    prep = rc.multi_prepare(scenes)
    for rpt in (1, 8, 512, 4096):
        best = 1e9
        for _ in range(6):
            t0 = time.perf_counter()
            if g == 1:
                rc.view_factor_totals(scenes[0], rpt, seed=7)
            else:
                rc.view_factor_totals_multi(scenes, rpt, seed=7)
            best = min(best, time.perf_counter() - t0)
        print(f"ranks {g}  rays_per_triangle {rpt:5d}: {best * 1e3:8.3f} ms per call  (device clock of scenes[0]: {scenes[0].last_kernel_ms():.3f} ms)", flush=True)
    for s in scenes:
        s.free()
