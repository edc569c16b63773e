"""Child of tests/test_oracle_sanitized.py (ThreadSanitizer leg): drives every multi-threaded entry point of the oracle's thread pool with 8
threads -- rco_trace_batch (closest / any, with and without per-ray counters), the deferred-order trace, get_illumination, hits_from_grid,
view_factors (matrix and row blocks), trace4 -- twice each, and checks the results against the single-threaded run.  The interpreter was
started with libtsan in LD_PRELOAD and RC_ORACLE_VARIANT=tsan; any data race the pool has is printed by the runtime (the parent fails on it)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from oracle import pyoracle as po
    import raycore_jl_amd as rc           # scenes only (pure numpy): the product library is never loaded here
    from helpers import build_oracle, random_rays
    cfg = rc.scenes.config_c3(lattice=(2, 2, 1))
    o = build_oracle(po, cfg)
    wb = o.world_bound
    lo, hi = wb[:3], wb[3:]
    rays = random_rays(rc, 20000, 5, lo, hi)
    for mode in ("closest", "any"):
        many = o.trace(rays, mode=mode, nthreads=8)          # the pool first: shared state is at its initial value when the threads meet it
        one = o.trace(rays, mode=mode, nthreads=1)
        for _ in range(2):
            assert o.trace(rays, mode=mode, nthreads=8).tobytes() == one.tobytes() == many.tobytes(), mode
    ill1 = o.get_illumination((0.3, 0.2, 1.0), 96, nthreads=1)
    assert np.array_equal(o.get_illumination((0.3, 0.2, 1.0), 96, nthreads=8), ill1)
    small = rc.scenes.fan_sphere(10, 6, centre=(0, 0, 0), radius=0.5)
    room = np.concatenate([small, rc.scenes.box_room((-1.5, -1.5, -1.5), (1.5, 1.5, 1.5), 2)])
    cfg2 = {"blas": [(room, np.arange(1, len(room) + 1, dtype=np.uint32))], "instances": [(1, rc.scenes.IDENTITY3x4[None], np.zeros(1, np.uint32))]}
    o2 = build_oracle(po, cfg2)
    vf1 = o2.view_factors(64, seed=3, nthreads=1)
    for _ in range(2):
        assert np.array_equal(o2.view_factors(64, seed=3, nthreads=8), vf1)
    s = po.Scene()
    b = s.add_blas(room)
    s.add_instance(b)
    s.build()
    r4 = random_rays(rc, 5000, 9, (-1.5, -1.5, -1.5), (1.5, 1.5, 1.5))
    assert s.trace4(b, r4, nthreads=8).tobytes() == s.trace4(b, r4, nthreads=1).tobytes()
    print("threads-ok")


if __name__ == "__main__":
    main()
