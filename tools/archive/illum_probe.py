"""get_illumination (src/kernels.jl:112-124) at its default 1000 x 1000 grid, repeated: device time per call with and without cost-ordered claiming."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raycore_jl_amd as rc
from tools.perf_probe import build
sc = rc.scenes
for name, cfg in (("C2 (100 k random triangles)", sc.config_c2()), ("C3 lattice", sc.config_c3())):
    t = build(cfg)
    vd = cfg.get("viewdir", (0.3, 0.2, 1.0))
    ref = None
    for co in (0, 1, 0, 1):
        t.set_option("cost_order", co)
        ms = []
        for rep in range(8):
            out = rc.get_illumination(t, vd, 1000)
            ms.append(t.last_kernel_ms())
            if ref is None: ref = out
            assert np.array_equal(out, ref)
        print(f"{name}: cost_order={co}: device ms per call {' '.join(f'{m:.3f}' for m in ms)}  -> best {min(ms):.3f} ms = {1e3 / min(ms):.0f} Mrays/s", flush=True)
