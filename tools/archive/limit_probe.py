"""Dev tool: what bounds the C3 trace?  Same kernel, same per-ray work, different memory behaviour:
(a) the real primary rays; (b) every wave traces 64 copies of ONE ray (perfect coalescing: 1 line per load instruction);
(c) rays permuted randomly (worst coherence).  Compares kernels 3 and 4."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raycore_jl_amd as rc
from perf_probe import build, time_trace


def main():
    sc = rc.scenes
    cfg = sc.config_c3()
    t = build(cfg)
    rays = sc.c3_primary_rays(cfg, 2048, 2048)
    n = len(rays)
    rep = np.repeat(rays[::64], 64)[:n]            # each aligned 64-ray packet = one ray, 64 times
    g = np.random.default_rng(0)
    perm = rays[g.permutation(n)]
    for kern in (3, 4, 1, 0):
        t.set_option("kernel", kern)
        for name, r in (("real", rays), ("packet-of-identical", rep), ("permuted", perm)):
            ms, hits = time_trace(t, r, "closest", 4)
            print(f"kernel {kern} {name:22s} {ms:7.3f} ms {n / ms / 1e3:8.1f} Mrays/s hit={hits['hit'].mean():.3f}", flush=True)


if __name__ == "__main__":
    main()
